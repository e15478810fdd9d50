"""loss heads: sigmoid cross-entropy sums, means, distances, the gradient penalty."""
import ctypes as C
import os
import torch
from .. import _lib
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from .._lib import check  # noqa: F401
from ._core import _L, _p, _stream, _c, HEAD_LOGITS, _PENDING_COSTS, is_unit_seed, LATE_EXT, _LATE_TERMS, mark_ready, wait_ready  # noqa: F401
from .linear import cached_const  # noqa: F401


# ---------------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------------


class BceSum(Function):
    """sum_i weight_i * mean(sigmoid_cross_entropy_with_logits(x_i, label_i)) -> 0-dim tensor; one launch forward and one
    backward for all terms."""

    @staticmethod
    def _tables(logits, labels, weights):
        n = len(logits)
        assert n <= _lib.BCE_MAX, 'too many BCE terms for one launch'
        return ((C.c_void_p * n)(*[x.data_ptr() for x in logits]), (C.c_float * n)(*[float(z) for z in labels]),
                (C.c_float * n)(*[float(w) for w in weights]), (C.c_int * n)(*[x.numel() for x in logits]), n)

    @staticmethod
    def _heads_of(logits):
        """[(CriticHead record, number of terms)] when the terms partition the logits of one or two critic heads into consecutive
        row ranges (in order), else None"""
        if not HEAD_LOGITS:
            return None
        heads, i = [], 0
        while i < len(logits):
            rec = HEAD_LOGITS.get(logits[i].data_ptr())
            # (weak references: a record that outlives its step -- logits that never met a BCE cost -- must not keep tape tensors
            #  alive, and an address can come back for another tensor)
            if (rec is None or rec['g_ptr'] is not None or rec['h']() is None or rec['w_out']() is None or len(heads) == _lib.BCE_HEADS
                    or any(rec is r for r, _ in heads)):
                return None
            rows, j = 0, i
            while j < len(logits) and rows < rec['M'] and logits[j].data_ptr() == rec['ptr'] + 4 * rows:
                rows += logits[j].numel()
                j += 1
            if rows != rec['M']:
                return None
            heads.append((rec, j - i))
            i = j
        for rec, _ in heads:
            HEAD_LOGITS.pop(rec['ptr'], None)
        return heads

    @staticmethod
    def _grad_buffers(logits, device):
        # one gradient buffer; terms that are adjacent rows of one tensor (the critic evaluated on [fake; real]) get adjacent
        # slices, so SplitRows.backward can hand the buffer on without a concatenation
        sizes = [x.numel() for x in logits]
        buf = torch.empty((sum(sizes),), dtype=torch.float32, device=device)
        outs, o = [], 0
        for nn in sizes:
            outs.append(buf[o:o + nn])
            o += nn
        return outs

    @staticmethod
    def forward(ctx, labels, weights, *logits):
        logits = [_c(x).reshape(-1) for x in logits]
        loss = torch.empty((1,), dtype=torch.float32, device=logits[0].device)
        xs, zs, ws, ns, n = BceSum._tables(logits, labels, weights)
        ctx.labels, ctx.weights = labels, weights
        ctx.unit_grads = None
        if any(ctx.needs_input_grad[2:]) and not os.environ.get('GGAN_NO_BCE_FWD_GRAD'):
            # a train op differentiates its cost with a unit seed (UNIT_SEEDS): the gradients for that case leave with the forward
            # launch; any other upstream gradient takes the backward kernel
            outs = BceSum._grad_buffers(logits, loss.device)
            gxs = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
            heads = BceSum._heads_of(logits)
            all_terms = tuple((x.numel(), float(z), float(wt)) for x, z, wt in zip(logits, labels, weights))

            def hinted_ok():
                # every head ran with ITS terms of this cost as its hint (head_bce_hint); four terms at most in the carrying launch
                if heads is None or len(all_terms) > 4:
                    return False
                k0 = 0
                for rec, nt in heads:
                    hh = rec.get('hinted')
                    if hh is None or hh['kind'] != 'bce' or hh['terms'] != all_terms[k0:k0 + nt]:
                        return False
                    k0 += nt
                return True
            if hinted_ok():
                # g and gh of every head exist already; d_wout / d_bout come with each head's backward products, and the LAST head's products
                # carry the cost's value -- all terms in order, the other heads' logits read in place (ext terms) -- nothing to launch here
                new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=loss.device)
                outs, k0 = [], 0
                for hi, (rec, nt) in enumerate(heads):
                    hh = rec['hinted']
                    mine, o = [], 0
                    for x in logits[k0:k0 + nt]:
                        mine.append(hh['g'][o:o + x.numel()])
                        o += x.numel()
                    outs += mine
                    rec['gh'] = hh['gh']
                    rec['d_wout'] = new(rec['H']) if rec['want_out'] else None
                    rec['d_bout'] = new(1) if rec['want_bout'] else None
                    rec['g_ptr'], rec['g_version'] = mine[0].data_ptr(), mine[0]._version
                    if hi == len(heads) - 1:
                        rec['tail'] = dict(kind='bce', terms=all_terms, logits=logits[k0], g=hh['g'], loss=loss,
                                           ext=[logits[j] for j in range(k0)] + [None] * nt)
                        _PENDING_COSTS[loss.data_ptr()] = rec['tail']
                    else:
                        rec['tail'] = dict(kind='bce', terms=hh['terms'], logits=logits[k0], g=hh['g'], loss=None, ext=None)
                    k0 += nt
            elif heads is not None:
                # every term is a row range of a critic head's logits (one head, or the two heads of the mixture scripts): the head
                # kernels of those ops' backward ride along
                new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=loss.device)
                k = 0
                for rec, nt in heads:
                    rec['gh'] = new(rec['M'], rec['H'])
                    rec['d_wout'] = new(rec['H']) if rec['want_out'] else None
                    rec['d_bout'] = new(1) if rec['want_bout'] else None
                    rec['g_ptr'], rec['g_version'] = outs[k].data_ptr(), outs[k]._version
                    k += nt
                m = len(heads)
                arr = lambda ct, vals: (ct * m)(*vals)
                ptrs = lambda key, call=False: arr(C.c_void_p, [(_t.data_ptr() if _t is not None else 0) for _t in
                                                                [(r[key]() if call else r[key]) for r, _ in heads]])
                check(_L().ggan_bce_heads_bwd(xs, zs, ws, ns, n, _p(loss), gxs, m, arr(C.c_int, [nt for _, nt in heads]),
                                              arr(C.c_int, [r['M'] for r, _ in heads]), arr(C.c_int, [r['H'] for r, _ in heads]),
                                              ptrs('h', True), ptrs('w_out', True), arr(C.c_float, [r['alpha'] for r, _ in heads]),
                                              ptrs('gh'), ptrs('d_wout'), ptrs('d_bout'), _stream()), 'ggan_bce_heads_bwd')
            else:
                check(_L().ggan_bce_logits_multi_fwd_grad(xs, zs, ws, ns, n, _p(loss), gxs, _stream()), 'ggan_bce_logits_multi_fwd_grad')
            ctx.unit_grads = outs
        else:
            check(_L().ggan_bce_logits_multi_fwd(xs, zs, ws, ns, n, _p(loss), _stream()), 'ggan_bce_logits_multi_fwd')
        ctx.save_for_backward(*logits)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if ctx.unit_grads is not None and is_unit_seed(g):
            return (None, None) + tuple(ctx.unit_grads)
        g = _c(g.reshape(1))
        logits = ctx.saved_tensors
        outs = BceSum._grad_buffers(logits, g.device)
        xs, zs, ws, ns, n = BceSum._tables(logits, ctx.labels, ctx.weights)
        gxs = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
        check(_L().ggan_bce_logits_multi_bwd(xs, zs, ws, ns, n, _p(g), gxs, _stream()), 'ggan_bce_logits_multi_bwd')
        return (None, None) + tuple(outs)


class Distance(Function):
    """weight * mean(|x - y|^p), p = 1 | 2 -> 0-dim tensor (tflib/utils/distance.py)."""

    @staticmethod
    def forward(ctx, x, y, p, weight):
        ctx.shapes = (tuple(x.shape), tuple(y.shape))
        x, y = _c(x).reshape(-1), _c(y).reshape(-1)
        assert x.numel() == y.numel()
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
        check(_L().ggan_dist_fwd(_p(x), _p(y), _p(out), x.numel(), int(p), float(weight), 0, _stream()), 'ggan_dist_fwd')
        ctx.p, ctx.weight = int(p), float(weight)
        ctx.save_for_backward(x, y)
        return out.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = _c(g.reshape(1))
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if gx is not None or gy is not None:
            check(_L().ggan_dist_bwd(_p(x), _p(y), _p(g), _p(gx), _p(gy), x.numel(), ctx.p, ctx.weight, _stream()),
                  'ggan_dist_bwd')
        return (gx.reshape(ctx.shapes[0]) if gx is not None else None, gy.reshape(ctx.shapes[1]) if gy is not None else None,
                None, None)


class MeanSum(Function):
    """sum_i weight_i * mean(x_i) -> 0-dim tensor (Wasserstein costs)."""

    @staticmethod
    def _hinted_head(xs, weights):
        """the CriticHead record whose 'mean' hint these terms fulfil: the leading terms are the hinted row ranges of its logits (sizes and
        weights as hinted), every further term has one element and weight 1 (the gradient penalty) -- else None"""
        rec = HEAD_LOGITS.get(xs[0].data_ptr()) if HEAD_LOGITS else None
        if rec is None or rec['g_ptr'] is not None or rec['h']() is None or rec['w_out']() is None:
            return None
        hh = rec.get('hinted')
        if hh is None or hh['kind'] != 'mean' or len(xs) < len(hh['terms']):
            return None
        rows = 0
        for x, wt, (n, _, hw) in zip(xs, weights, hh['terms']):
            if x.numel() != n or float(wt) != hw or x.data_ptr() != rec['ptr'] + 4 * rows:
                return None
            rows += n
        nh = len(hh['terms'])
        if rows != rec['M'] or any(x.numel() != 1 or float(wt) != 1.0 for x, wt in zip(xs[nh:], weights[nh:])):
            return None
        HEAD_LOGITS.pop(rec['ptr'], None)
        return rec

    @staticmethod
    def forward(ctx, weights, *xs):
        ctx.shapes = [x.shape for x in xs]
        xs = [_c(x).reshape(-1) for x in xs]
        loss = torch.empty((1,), dtype=torch.float32, device=xs[0].device)
        ctx.weights = weights
        ctx.sizes = [x.numel() for x in xs]
        ctx.dev = xs[0].device
        ctx.unit_grads = None
        n = len(xs)
        hrec = MeanSum._hinted_head(xs, weights) if (any(ctx.needs_input_grad[1:]) and n <= 4) else None
        if hrec is not None:
            # the critic head ran with this cost's row terms as its hint (head_bce_hint(kind='mean')): g and gh exist, the cost's value
            # (with the one-element terms that follow the head's rows: the gradient penalty), d_wout and d_bout come with the head's
            # backward products -- nothing to launch here
            hh = hrec['hinted']
            new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=loss.device)
            nh = len(hh['terms'])
            outs, o = [], 0
            for x in xs[:nh]:
                outs.append(hh['g'][o:o + x.numel()])
                o += x.numel()
            outs += [cached_const(1.0, (1,), loss.device) for _ in xs[nh:]]      # (d cost / d term = its weight, 1)
            hrec['gh'] = hh['gh']
            hrec['d_wout'] = new(hrec['H']) if hrec['want_out'] else None
            hrec['d_bout'] = new(1) if hrec['want_bout'] else None
            hrec['g_ptr'], hrec['g_version'] = outs[0].data_ptr(), outs[0]._version
            if LATE_EXT[0] and len(xs) > nh:
                # (the one-element terms -- the penalty -- stay out of the head's launch: the backward pass of these logits then does not wait
                #  for whatever computes them; the caller adds them to the value later: functional.add_late_terms)
                hrec['tail'] = dict(kind='mean', terms=hh['terms'], logits=xs[0], g=hh['g'], loss=loss, ext=None)
                for x in xs[nh:]:
                    _LATE_TERMS.append((loss, x))
            else:
                for x in xs[nh:]:
                    wait_ready(x)
                hrec['tail'] = dict(kind='mean', terms=hh['terms'] + tuple((1, 0.0, 1.0) for _ in xs[nh:]), logits=xs[0], g=hh['g'], loss=loss,
                                    ext=[None] * nh + list(xs[nh:]))
            _PENDING_COSTS[loss.data_ptr()] = hrec['tail']
            ctx.unit_grads = outs
            return loss.reshape(())
        for x in xs:
            if x.numel() == 1:
                wait_ready(x)          # (a one-element term whose producer ran on another stream and was not joined: LATE_EXT)
        if n <= _lib.BCE_MAX and not os.environ.get('GGAN_NO_BCE_FWD_GRAD'):
            # one launch for all terms; with it (as BceSum) the gradients for the unit seed of a train op, in ONE buffer so that
            # the halves of a batched critic's logits get adjacent slices (SplitRows.backward: no concatenation)
            outs = BceSum._grad_buffers(xs, ctx.dev) if any(ctx.needs_input_grad[1:]) else None
            px = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
            pw = (C.c_float * n)(*[float(w) for w in weights])
            pn = (C.c_int * n)(*ctx.sizes)
            pg = (C.c_void_p * n)(*[t.data_ptr() for t in outs]) if outs is not None else None
            check(_L().ggan_mean_multi_fwd_grad(px, pw, pn, n, _p(loss), pg, _stream()), 'ggan_mean_multi_fwd_grad')
            ctx.unit_grads = outs
            return loss.reshape(())
        for i, (x, w) in enumerate(zip(xs, weights)):
            check(_L().ggan_mean_fwd(_p(x), float(w), _p(loss), x.numel(), int(i > 0), _stream()), 'ggan_mean_fwd')
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if ctx.unit_grads is not None and is_unit_seed(g):
            # (a one-element term of weight 1 -- the gradient penalty riding in the cost launch -- receives the seed itself, so that its
            #  producer recognises it in turn)
            return (None,) + tuple((g if (n == 1 and float(w) == 1.0 and shp == g.shape) else u.reshape(shp))
                                   for u, n, w, shp in zip(ctx.unit_grads, ctx.sizes, ctx.weights, ctx.shapes))
        g = _c(g.reshape(1))
        outs = []
        for n, w, shp in zip(ctx.sizes, ctx.weights, ctx.shapes):
            gx = torch.empty((n,), dtype=torch.float32, device=ctx.dev)
            check(_L().ggan_mean_bwd(_p(g), float(w), _p(gx), n, _stream()), 'ggan_mean_bwd')
            outs.append(gx.reshape(shp))
        return (None,) + tuple(outs)


class GradPenalty(Function):
    """lam * mean_b((||g[b,:]||_2 - 1)^2)  (gan_inference_cifar10.py:363-364).  The forward launch also leaves d(pen)/dg for a unit
    upstream gradient (the penalty enters the critic cost with weight 1): backward hands that out when the train op's unit seed
    comes back (as BceSum / MeanSum do), and runs the backward kernel otherwise."""

    _ARRIVE = {}

    @staticmethod
    def forward(ctx, g, lam):
        g = _c(g)
        B, D = g.shape
        slopes = torch.empty((B,), dtype=torch.float32, device=g.device)
        pen = torch.empty((1,), dtype=torch.float32, device=g.device)
        ctx.unit_grad = None
        if ctx.needs_input_grad[0] and not os.environ.get('GGAN_NO_BCE_FWD_GRAD'):
            key = (g.device.type, g.device.index)
            arrive = GradPenalty._ARRIVE.get(key)
            if arrive is None:
                arrive = GradPenalty._ARRIVE[key] = torch.zeros((1,), dtype=torch.int32, device=g.device)
            gg = torch.empty_like(g)
            check(_L().ggan_gp_penalty_fwd_grad(_p(g), _p(slopes), _p(pen), _p(gg), _p(arrive), B, D, lam, _stream()),
                  'ggan_gp_penalty_fwd_grad')
            ctx.unit_grad = gg
        else:
            check(_L().ggan_gp_penalty_fwd(_p(g), _p(slopes), _p(pen), B, D, lam, _stream()), 'ggan_gp_penalty_fwd')
        ctx.lam = lam
        ctx.save_for_backward(g, slopes)
        if LATE_EXT[0]:
            mark_ready(pen)
        return pen.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, gpen):
        if ctx.unit_grad is not None and is_unit_seed(gpen):
            return ctx.unit_grad, None
        g, slopes = ctx.saved_tensors
        B, D = g.shape
        gg = torch.empty_like(g)
        check(_L().ggan_gp_penalty_bwd(_p(g), _p(slopes), _p(_c(gpen.reshape(1))), _p(gg), B, D, ctx.lam, _stream()),
              'ggan_gp_penalty_bwd')
        return gg, None
