"""Linear family: Gemm / Gemm2 (+ fused backward), the critic head, the dynamic scan of the state-space models."""
import ctypes as C
import os
import threading as _threading
import torch
import weakref
from .. import _lib
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from .._lib import ACT_NONE, check  # noqa: F401
from ._core import (  # noqa: F401
    _L, _p, _stream, _c, FUSED_CONV_BWD, _DATA_ONLY, _is_param, _skip_undefined, workspace, _new_out, HEAD_LOGITS, _PENDING_COSTS,
    _tail_value)
from .pointwise import ActBwd  # noqa: F401
from .conv import TALL_ROWS, ColSum  # noqa: F401


# ---------------------------------------------------------------------------------------------------
# dense
# ---------------------------------------------------------------------------------------------------
@_skip_undefined
class Gemm(Function):
    """C[M,N] = op(A) op(B) + bias[N]; ta/tb read the stored operand transposed."""

    @staticmethod
    def forward(ctx, a, b, bias, ta, tb, act, alpha, slot=None):
        a, b = _c(a), _c(b)
        M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
        K2, N = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
        assert K == K2, (a.shape, b.shape, ta, tb)
        out = _new_out(slot, (M, N), a.device)
        ws = workspace(a.device)
        check(_L().ggan_gemm(int(ta), int(tb), M, N, K, _p(a), _p(b), _p(_c(bias)) if bias is not None else _p(None),
                             _p(out), act, alpha, _p(ws), ws.numel(), _stream()), 'ggan_gemm')
        ctx.ta, ctx.tb, ctx.act, ctx.alpha, ctx.has_bias = ta, tb, act, alpha, bias is not None
        ctx.b_param, ctx.bias_param = _is_param(b), _is_param(bias)
        ctx.save_for_backward(a, b, out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, out = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        if (ctx.act != ACT_NONE and not ta and not tb and not torch.is_grad_enabled() and FUSED_CONV_BWD
                and a.shape[1] <= 1024 and a.shape[0] < TALL_ROWS):
            # (measured, tools/bench_gemm.py: for the long-K layers the masked operand loads cost more than the separate
            # act_bwd pass they replace, 27-48 us vs 26-39 us; for K <= 1024 the fused pair wins)
            return _fused_linear_backward(ctx, g, a, b, out)
        if ctx.act != ACT_NONE:
            g = ActBwd.apply(g, out, ctx.act, ctx.alpha)
        da = db = dbias = None
        skip_b = _DATA_ONLY[0] and ctx.b_param                                 # (parameter gradients not wanted by this grad call)
        skip_bias = _DATA_ONLY[0] and ctx.bias_param
        if skip_b and (skip_bias or not (ctx.has_bias and ctx.needs_input_grad[2])):
            pass
        elif skip_b or skip_bias:
            if ctx.needs_input_grad[1] and not skip_b:
                db = Gemm.apply(a, g, None, not ta, False, ACT_NONE, 0.0) if not tb else Gemm.apply(g, a, None, True, ta, ACT_NONE, 0.0)
            if ctx.has_bias and ctx.needs_input_grad[2] and not skip_bias:
                dbias = ColSum.apply(g)
        elif (ctx.needs_input_grad[1] and not tb and ctx.has_bias and ctx.needs_input_grad[2]
                and not torch.is_grad_enabled() and g.shape[0] < TALL_ROWS):
            db, dbias = gemm_colsum_(a, g, not ta)                             # dW and db in ONE launch
        else:
            if ctx.needs_input_grad[1]:
                if not tb:
                    db = Gemm.apply(a, g, None, not ta, False, ACT_NONE, 0.0)      # op(A)^T g
                else:
                    db = Gemm.apply(g, a, None, True, ta, ACT_NONE, 0.0)           # g^T op(A)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                dbias = ColSum.apply(g)
        if ctx.needs_input_grad[0]:
            if not ta:
                da = Gemm.apply(g, b, None, False, not tb, ACT_NONE, 0.0)      # g op(B)^T
            else:
                da = Gemm.apply(b, g, None, tb, True, ACT_NONE, 0.0)           # op(B) g^T
        return (da, db, dbias) + (None,) * (len(ctx.needs_input_grad) - 3)


def _fused_linear_backward(ctx, g, x, w, y):
    """Linear(+activation) backward in two launches: the activation derivative is applied while g is staged."""
    g = _c(g)
    M, K = x.shape
    N = w.shape[1]
    L = _L()
    ws = workspace(g.device)
    dx = dw = db = None
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        dw = torch.empty((K, N), dtype=torch.float32, device=g.device)
        db = torch.empty((N,), dtype=torch.float32, device=g.device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        check(L.ggan_linear_bwd_weight_act(M, N, K, _p(x), _p(g), _p(y), ctx.act, ctx.alpha, _p(dw), _p(db), _p(ws), ws.numel(),
                                           _stream()), 'ggan_linear_bwd_weight_act')
        if not ctx.needs_input_grad[1]:
            dw = None
    if ctx.needs_input_grad[0]:
        dx = torch.empty((M, K), dtype=torch.float32, device=g.device)
        check(L.ggan_linear_bwd_data_act(M, N, K, _p(g), _p(y), ctx.act, ctx.alpha, _p(w), _p(dx), _p(ws), ws.numel(), _stream()),
              'ggan_linear_bwd_data_act')
    return (dx, dw, db) + (None,) * (len(ctx.needs_input_grad) - 3)


@_skip_undefined
class Gemm2(Function):
    """[a1 | a2] @ w + bias without materialising the concatenation (tf.concat + Linear of the joint critic,
    gan_inference_cifar10.py:246-248): the GEMM reads its A operand from two buffers, its weight gradient likewise, and its
    data gradient leaves in two buffers (ggan_gemm_split).  Falls back to concatenate + Gemm while a double backward is
    recorded (wali-gp) or when the split is not tile-aligned."""

    @staticmethod
    def usable(a1, a2):
        return a1.shape[1] % 64 == 0 and a2.shape[1] >= 1 and a1.shape[0] == a2.shape[0]

    @staticmethod
    def forward(ctx, a1, a2, w, bias, act, alpha):
        a1, a2, w = _c(a1), _c(a2), _c(w)
        M, K1, K2 = a1.shape[0], a1.shape[1], a2.shape[1]
        K, N = w.shape
        assert K == K1 + K2, (a1.shape, a2.shape, w.shape)
        out = torch.empty((M, N), dtype=torch.float32, device=a1.device)
        ws = workspace(a1.device)
        check(_L().ggan_gemm_split(0, 0, M, N, K, _p(a1), _p(a2), K1, _p(w), _p(_c(bias)) if bias is not None else _p(None), _p(out),
                                   _p(None), 0, _p(None), act, alpha, _p(ws), ws.numel(), _stream()), 'ggan_gemm_split')
        ctx.act, ctx.alpha, ctx.has_bias = act, alpha, bias is not None
        ctx.w_param, ctx.bias_param = _is_param(w), _is_param(bias)
        ctx.save_for_backward(a1, a2, w, out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, g):
        a1, a2, w, out = ctx.saved_tensors
        K1 = a1.shape[1]
        if torch.is_grad_enabled():          # differentiable composition (second derivatives: wali-gp)
            if ctx.act != ACT_NONE:
                g = ActBwd.apply(g, out, ctx.act, ctx.alpha)
            # (the two halves leave in two buffers here too: column slices of one product were views that the conv stack then copied,
            #  and their backward a zero-filled [M, K] buffer plus a copy per critic step)
            da1, da2 = Gemm2Dgrad.apply(g, w, K1) if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else (None, None)
            dw = (Gemm.apply(torch.cat([a1, a2], 1), g, None, True, False, ACT_NONE, 0.0)
                  if (ctx.needs_input_grad[2] and not (_DATA_ONLY[0] and ctx.w_param)) else None)
            db = ColSum.apply(g) if (ctx.has_bias and ctx.needs_input_grad[3] and not (_DATA_ONLY[0] and ctx.bias_param)) else None
            return (da1, da2, dw, db, None, None)
        g = _c(g)
        if ctx.act != ACT_NONE:
            g = ActBwd.apply(g, out, ctx.act, ctx.alpha)
        M, N = g.shape
        K = w.shape[0]
        L, ws = _L(), workspace(g.device)
        da1 = da2 = dw = db = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            dw = torch.empty((K, N), dtype=torch.float32, device=g.device)
            db = torch.empty((N,), dtype=torch.float32, device=g.device) if ctx.has_bias else None
            # dW[K,N] = [a1 | a2]^T g (A stored [M, K]: transposed read, the sources split the OUTPUT rows) + column sums of g
            check(L.ggan_gemm_split(1, 0, K, N, M, _p(a1), _p(a2), K1, _p(g), _p(None), _p(dw), _p(None), 0, _p(db), ACT_NONE, 0.0,
                                    _p(ws), ws.numel(), _stream()), 'ggan_gemm_split')
            if not ctx.needs_input_grad[2]:
                dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            da1 = torch.empty_like(a1)
            da2 = torch.empty_like(a2)
            # [da1 | da2] = g w^T, columns < K1 to da1, the rest to da2
            check(L.ggan_gemm_split(0, 1, M, K, N, _p(g), _p(None), 0, _p(w), _p(None), _p(da1), _p(da2), K1, _p(None), ACT_NONE, 0.0,
                                    _p(ws), ws.numel(), _stream()), 'ggan_gemm_split')
        return da1, da2, dw, db if (ctx.has_bias and ctx.needs_input_grad[3]) else None, None, None


_CONSTS = {}


def cached_const(value, shape, device):
    """a persistent constant tensor (never written): unit seeds of inner gradient calls, zero operands -- no fill launch per step"""
    device = torch.device(device)
    key = (float(value), tuple(shape), device.type, device.index)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full(tuple(shape), float(value), dtype=torch.float32, device=device)
    return t


class Gemm2Dgrad(Function):
    """(da1[M,K1], da2[M,K-K1]) = the column halves of g[M,N] @ w[K,N]^T -- Gemm2's data gradient as an op of its own, so that the
    gradient-penalty pass can differentiate through it: its backward is Gemm2's forward ([h1 | h2] @ w) and Gemm2's weight-gradient
    form ([h1 | h2]^T g), both reading the two halves in place (ggan_gemm_split).  An undefined half is a zero operand."""

    @staticmethod
    def forward(ctx, g, w, K1):
        g, w = _c(g), _c(w)
        M, N = g.shape
        K = w.shape[0]
        da1 = torch.empty((M, K1), dtype=torch.float32, device=g.device)
        da2 = torch.empty((M, K - K1), dtype=torch.float32, device=g.device)
        ws = workspace(g.device)
        check(_L().ggan_gemm_split(0, 1, M, K, N, _p(g), _p(None), 0, _p(w), _p(None), _p(da1), _p(da2), K1, _p(None), ACT_NONE, 0.0,
                                   _p(ws), ws.numel(), _stream()), 'ggan_gemm_split')
        ctx.K1 = K1
        ctx.w_param = _is_param(w)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(g, w)
        return da1, da2

    @staticmethod
    def backward(ctx, h1, h2):
        if h1 is None and h2 is None:
            return None, None, None
        g, w = ctx.saved_tensors
        K1 = ctx.K1
        M, N = g.shape
        K = w.shape[0]
        h1 = _c(h1) if h1 is not None else cached_const(0.0, (M, K1), g.device)
        h2 = _c(h2) if h2 is not None else cached_const(0.0, (M, K - K1), g.device)
        dg = dw = None
        if ctx.needs_input_grad[0]:
            dg = Gemm2.apply(h1, h2, w, None, ACT_NONE, 0.0)
        if ctx.needs_input_grad[1] and not (_DATA_ONLY[0] and ctx.w_param):
            if torch.is_grad_enabled():            # (a third derivative: plain composition)
                dw = Gemm.apply(torch.cat([h1, h2], 1), g, None, True, False, ACT_NONE, 0.0)
            else:
                dw = torch.empty((K, N), dtype=torch.float32, device=g.device)
                ws = workspace(g.device)
                check(_L().ggan_gemm_split(1, 0, K, N, M, _p(h1), _p(h2), K1, _p(g), _p(None), _p(dw), _p(None), 0, _p(None), ACT_NONE,
                                           0.0, _p(ws), ws.numel(), _stream()), 'ggan_gemm_split')
        return dg, dw, None


_HEAD_HINT = _threading.local()      # .terms: the cost hint in force on this thread (head_bce_hint)


class head_bce_hint(object):
    """`with head_bce_hint([(rows, label, weight), ...]):` -- the caller knows that the logits of the critic head evaluated inside are
    going to be the terms (consecutive row ranges, in order) of ONE BceSum cost that a train op differentiates with a unit seed.  The
    head's forward then leaves the cost's gradient and gh behind in its tail launch (ggan_critic_head_fwd_bce) and its backward's
    product launch carries the cost itself (ggan_critic_head_bwd_tail): one launch less on the critical chain of the step, values
    bit-identical.  The cost's value exists once the head's backward has run -- only for steps that run it (engine.Trainer).  A cost
    built from other terms than hinted ignores the hint's by-products (BceSum launches as always)."""

    def __init__(self, terms, kind='bce'):
        """kind 'bce': terms (rows, label, weight) of a BceSum cost; kind 'mean': terms (rows, 0, weight) of a MeanSum cost (the
        Wasserstein costs; further one-element terms of weight 1 -- the gradient penalty -- may follow the head's rows in that cost)"""
        self.terms = (kind, tuple((int(n), float(z), float(w)) for n, z, w in terms)) if terms else None

    def __enter__(self):
        self.prev = getattr(_HEAD_HINT, 'terms', None)
        _HEAD_HINT.terms = self.terms
        return self

    def __exit__(self, *exc):
        _HEAD_HINT.terms = self.prev
        return False


@_skip_undefined
class CriticHead(Function):
    """logits[M] = Linear(H -> 1)(lrelu(Linear([a1 | a2] -> H))): the tail of a critic as one op (ggan_critic_head_fwd/bwd:
    split-K slabs summed by the tail kernel that also forms the logits; backward = one head kernel + one grouped launch for
    the weight- and the data-gradient product).  a2 may be None.  Not differentiable twice: the gradient-penalty pass of
    wali-gp composes the layers instead (tflib.ops.linear.LinearLReLULinear(differentiable=True))."""

    @staticmethod
    def forward(ctx, a1, a2, w, b, w_out, b_out, alpha):
        a1, w, b, w_out, b_out = _c(a1), _c(w), _c(b), _c(w_out), _c(b_out)
        a2 = _c(a2) if a2 is not None else None
        M, K1 = a1.shape
        K2 = a2.shape[1] if a2 is not None else 0
        H = w.shape[1]
        assert w.shape[0] == K1 + K2 and w_out.numel() == H and b.numel() == H, (a1.shape, w.shape, w_out.shape)
        h = torch.empty((M, H), dtype=torch.float32, device=a1.device)
        logits = torch.empty((M,), dtype=torch.float32, device=a1.device)
        ws = workspace(a1.device)
        registers = M <= _lib.HEAD_BCE_MAX_ROWS and any(ctx.needs_input_grad) and not os.environ.get('GGAN_NO_HEAD_BCE')
        hint = getattr(_HEAD_HINT, 'terms', None)
        hinted = None
        hkind, hterms = hint if hint is not None else (None, ())
        if (registers and hint is not None and 1 <= len(hterms) <= 3 and sum(n for n, _, _ in hterms) == M and H <= 2048
                and not os.environ.get('GGAN_NO_HEAD_HINT')):
            # the caller vouches for the cost these logits feed (head_bce_hint): its gradient and gh leave with the tail launch
            g = torch.empty((M,), dtype=torch.float32, device=a1.device)
            gh = torch.empty((M, H), dtype=torch.float32, device=a1.device)
            nt = len(hterms)
            tabs = ((C.c_int * nt)(*[n for n, _, _ in hterms]), (C.c_float * nt)(*[z for _, z, _ in hterms]),
                    (C.c_float * nt)(*[wt for _, _, wt in hterms]))
            check(_L().ggan_critic_head_fwd_bce(M, K1, K2, H, _p(a1), _p(a2), _p(w), _p(b), _p(w_out), _p(b_out), float(alpha), _p(h),
                                                _p(logits), 1 if hkind == 'mean' else 0, nt, tabs[0], tabs[1], tabs[2], _p(g), _p(gh), _p(ws),
                                                ws.numel(), _stream()), 'ggan_critic_head_fwd_bce')
            hinted = dict(kind=hkind, terms=hterms, g=g, gh=gh)
        else:
            check(_L().ggan_critic_head_fwd(M, K1, K2, H, _p(a1), _p(a2), _p(w), _p(b), _p(w_out), _p(b_out), float(alpha), _p(h),
                                            _p(logits), _p(ws), ws.numel(), _stream()), 'ggan_critic_head_fwd')
        ctx.alpha, ctx.has_a2 = float(alpha), a2 is not None
        ctx.save_for_backward(a1, a2, w, w_out, h)
        # a BCE cost on exactly these logits may take the head kernel of this op's backward into its own launch (BceSum)
        ctx.rec = None
        if registers:
            ctx.rec = dict(ptr=logits.data_ptr(), M=M, H=H, h=weakref.ref(h), w_out=weakref.ref(w_out), alpha=float(alpha),
                           want_out=ctx.needs_input_grad[4],
                           want_bout=ctx.needs_input_grad[5], g_ptr=None, hinted=hinted, tail=None)
            if len(HEAD_LOGITS) >= 8:            # (heads whose logits never met a BCE cost: Wasserstein modes)
                HEAD_LOGITS.clear()
            HEAD_LOGITS[ctx.rec['ptr']] = ctx.rec
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a1, a2, w, w_out, h = ctx.saved_tensors
        g = _c(g)
        M, K1 = a1.shape
        K2 = a2.shape[1] if a2 is not None else 0
        H = w.shape[1]
        need = ctx.needs_input_grad
        dev = g.device
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        rec = ctx.rec
        # (address AND content version of the gradient buffer the cost launch wrote: a gradient autograd accumulated in place into a buffer
        #  at that address carries another version and takes the backward kernel)
        fused = rec is not None and rec['g_ptr'] is not None and rec['g_ptr'] == g.data_ptr() and rec.get('g_version') == g._version
        gh = rec['gh'] if fused else new(M, H)
        want_a = need[0] or (a2 is not None and need[1])
        d_a1 = new(M, K1) if want_a else None
        d_a2 = new(M, K2) if (want_a and a2 is not None) else None
        d_w = new(K1 + K2, H) if (need[2] or need[3]) else None
        d_b = new(H) if need[3] else None
        d_wout = (rec['d_wout'].view(w_out.shape) if fused else new(*w_out.shape)) if need[4] else None
        d_bout = (rec['d_bout'] if fused else new(1)) if need[5] else None
        ws = workspace(dev)
        tail = rec.get('tail') if rec is not None else None
        if tail is not None:
            rec['tail'] = None
            if tail.get('loss') is not None:
                _PENDING_COSTS.pop(tail['loss'].data_ptr(), None)      # (the value is written below, one way or the other)
            if fused:
                # hinted head: gh and g left with the forward's tail launch; the products' launch carries the cost, d_wout, d_bout
                nt = len(tail['terms'])
                tabs = ((C.c_int * nt)(*[n for n, _, _ in tail['terms']]), (C.c_float * nt)(*[z for _, z, _ in tail['terms']]),
                        (C.c_float * nt)(*[wt for _, _, wt in tail['terms']]))
                ext = tail.get('ext')
                pext = (C.c_void_p * nt)(*[(e.data_ptr() if e is not None else 0) for e in ext]) if ext else None
                check(_L().ggan_critic_head_bwd_tail(M, K1, K2, H, _p(a1), _p(a2), _p(w), _p(h), _p(w_out), ctx.alpha, _p(gh), _p(d_a1), _p(d_a2),
                                                     _p(d_w), _p(d_b), _p(d_wout), _p(d_bout), _p(tail['logits']), _p(tail['g']),
                                                     1 if tail.get('kind') == 'mean' else 0, nt, tabs[0], tabs[1], tabs[2], pext,
                                                     _p(tail['loss']), _p(ws), ws.numel(), _stream()), 'ggan_critic_head_bwd_tail')
                return (d_a1 if need[0] else None, d_a2 if (a2 is not None and need[1]) else None, d_w if need[2] else None, d_b, d_wout,
                        d_bout, None)
            # (another upstream gradient than the unit seed reached this head: the cost's value is still owed -- by the head that carries it)
            if tail.get('loss') is not None:
                _tail_value(tail)
        # (fused: gh, d_wout, d_bout left with the cost's launch -- ggan_bce_head_bwd; g = NULL launches the products only)
        check(_L().ggan_critic_head_bwd(M, K1, K2, H, _p(None if fused else g), _p(a1), _p(a2), _p(w), _p(h), _p(w_out), ctx.alpha, _p(gh), _p(d_a1),
                                        _p(d_a2), _p(d_w), _p(d_b), _p(d_wout), _p(d_bout), _p(ws), ws.numel(), _stream()),
              'ggan_critic_head_bwd')
        return (d_a1 if need[0] else None, d_a2 if (a2 is not None and need[1]) else None, d_w if need[2] else None, d_b, d_wout,
                d_bout, None)


class DynScan(Function):
    """zs[B, T+1, dl]: the transition operator of the state-space scripts applied T times (ggan_dyn_scan_fwd / _bwd: one scan
    launch per direction, the weight gradients as products over all T*B rows).  zw / b_zw None: OP_DYN_MODE 'res'."""

    @staticmethod
    def usable(z0, eps, w_1):
        return w_1.shape[0] == 256 and w_1.shape[1] == 256 and z0.shape[1] <= 16 and eps.shape[1] <= 16

    @staticmethod
    def forward(ctx, z0, eps, w_in, b_in, w_1, b_1, w_out, b_out, zw, b_zw, T, alpha):
        z0, eps, w_in, b_in, w_1, b_1, w_out, b_out = (_c(t) for t in (z0, eps, w_in, b_in, w_1, b_1, w_out, b_out))
        zw, b_zw = (_c(zw), _c(b_zw)) if zw is not None else (None, None)
        B, dl = z0.shape
        dt, Hd = eps.shape[1], w_1.shape[0]
        assert w_in.shape == (dl + dt, Hd) and w_out.shape == (Hd, dl), (w_in.shape, w_out.shape)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=z0.device)
        zs, h1, h2 = new(B, T + 1, dl), new(T, B, Hd), new(T, B, Hd)
        check(_L().ggan_dyn_scan_fwd(B, T, dl, dt, Hd, _p(z0), _p(eps), _p(w_in), _p(b_in), _p(w_1), _p(b_1), _p(w_out), _p(b_out),
                                     _p(zw), _p(b_zw), float(alpha), _p(zs), _p(h1), _p(h2), _stream()), 'ggan_dyn_scan_fwd')
        ctx.T, ctx.alpha = T, float(alpha)
        ctx.save_for_backward(eps, w_in, w_1, w_out, zw, zs, h1, h2)
        return zs

    @staticmethod
    @once_differentiable
    def backward(ctx, g_zs):
        eps, w_in, w_1, w_out, zw, zs, h1, h2 = ctx.saved_tensors
        g_zs = _c(g_zs)
        B, T1, dl = zs.shape
        T, dt, Hd = T1 - 1, eps.shape[1], w_1.shape[0]
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=zs.device)
        G1, G2, Go, Xin = new(T * B, Hd), new(T * B, Hd), new(T * B, dl), new(T * B, dl + dt)
        d_z0, d_eps = new(B, dl), new(B, dt)
        check(_L().ggan_dyn_scan_bwd(B, T, dl, dt, Hd, _p(g_zs), _p(zs), _p(eps), _p(h1), _p(h2), _p(w_in), _p(w_1), _p(w_out), _p(zw),
                                     ctx.alpha, _p(G1), _p(G2), _p(Go), _p(Xin), _p(d_z0), _p(d_eps), _stream()), 'ggan_dyn_scan_bwd')
        need = ctx.needs_input_grad
        d_win = d_bin = d_w1 = d_b1 = d_wout = d_bout = d_zw = d_bzw = None
        if need[2] or need[3]:
            d_win, d_bin = gemm_colsum_(Xin, G1, True)                      # [dl+dt, H], [H]
        if need[4] or need[5]:
            d_w1, d_b1 = gemm_colsum_(h1.view(T * B, Hd), G2, True)         # [H, H], [H]
        if need[6] or need[7] or need[9]:
            d_wout, d_bout = gemm_colsum_(h2.view(T * B, Hd), Go, True)     # [H, dl], [dl]
        if zw is not None and need[8]:
            d_zw = Gemm.apply(Xin[:, :dl].contiguous(), Go, None, True, False, ACT_NONE, 0.0)
            d_bzw = d_bout
        return (d_z0 if need[0] else None, d_eps if need[1] else None, d_win, d_bin, d_w1, d_b1, d_wout, d_bout, d_zw, d_bzw,
                None, None)


def gemm_colsum_(a, g, ta):
    """C = op(A) @ g and colsum[n] = sum_k g[k, n] in one kernel (no autograd: used inside plain backward passes)."""
    a, g = _c(a), _c(g)
    M, K = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    K2, N = g.shape
    assert K == K2
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    cs = torch.empty((N,), dtype=torch.float32, device=a.device)
    ws = workspace(a.device)
    check(_L().ggan_gemm_colsum(int(ta), M, N, K, _p(a), _p(g), _p(out), _p(cs), _p(ws), ws.numel(), _stream()),
          'ggan_gemm_colsum')
    return out, cs


def linear(x, w, bias=None, act=ACT_NONE, alpha=0.0):
    return Gemm.apply(x, w, bias, False, False, act, alpha)
