"""state shared by every op family: library handle and pointer helpers, workspaces and side streams, launch plans (target_workgroups / launch_hint), conv geometry, row slots, and the registries the loss heads and optimizers share (HEAD_LOGITS, pending costs, unit seeds)."""
import ctypes as C
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID, ConvGeom, check  # noqa: F401

_WS = {}
_WS_BYTES = 192 << 20


def _L():
    return _lib.load()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t):
    if not t.is_cuda:
        raise _lib.GganError('graphical_gan_amd ops need tensors on a HIP device (got %s); there is no CPU path' % t.device)
    return t


def _c(t):
    """contiguous fp32 device tensor"""
    _dev(t)
    if t.dtype != torch.float32:
        raise _lib.GganError('fp32 expected, got %s' % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


import os as _os
FUSED_CONV_BWD = _os.environ.get('GGAN_NO_FUSED_BWD') is None


# Filter gradients as split-K partial slabs: inside `defer_wgrad_reduce` the filter-gradient kernels leave their slabs in a
# private buffer and return slab 0 (+ a registry entry); `pack_` sums the slabs while it gathers the gradient bucket, so the
# per-layer reduce launches disappear.  Only legal when the returned tensors go straight to pack_ (one gradient
# contribution per parameter, no other consumer) -- pack_ raises if a registered tensor never reached it.
_DEFER = [None]


# ctx.needs_input_grad says whether an input REQUIRES grad, not whether the running torch.autograd.grad call asked for it.  The
# gradient-penalty construction differentiates the critic w.r.t. its INPUT only (create_graph=True); without a hint every layer's
# first-order backward would also form its weight and bias gradients there -- three filter-gradient launches, two weight-gradient
# products and their reduce / column-sum launches per critic step of wali-gp (~150 us), all discarded by the tape.
_DATA_ONLY = [False]


def _is_param(t):
    """a registry parameter or one of its second leaves (tflib.param tags both): the operands data_grad_only may skip.  A weight slot
    fed with a data-dependent tensor (a product of two activations, say) keeps its gradient."""
    return t is not None and getattr(t, 'param_name', None) is not None


class data_grad_only(object):
    """with data_grad_only(): torch.autograd.grad(out, [x], create_graph=True) -- layer backwards skip the gradients of PARAMETER
    operands (weights / biases handed out by tflib.param); any other operand in a weight slot is differentiated as usual"""

    def __enter__(self):
        self.prev, _DATA_ONLY[0] = _DATA_ONLY[0], True
        return self

    def __exit__(self, *a):
        _DATA_ONLY[0] = self.prev


def _skip_undefined(cls):
    """Single-output Functions: an undefined incoming gradient means "no gradient", not a zero tensor to push through the layer.
    The tape reaches a forward node whenever the graph has an edge to it, also when every edge delivers None at run time -- the
    gradient-penalty pass is the case that matters: its first-order backward reads the forward activations only as LeakyReLU
    sign references (derivative zero a.e., returned as None), so in the final backward the critic's forward nodes of that pass
    become ready with an undefined gradient.  torch materialises it as zeros by default and the node then runs its whole backward
    on zeros (measured: three conv layers' filter- and data-gradient kernels per critic step of wali-gp, all on zero input)."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx.set_materialize_grads(False)
        ctx._n_in = len(args)
        return fwd(ctx, *args)

    def backward(ctx, g):
        if g is None:
            return (None,) * ctx._n_in
        return bwd(ctx, g)
    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls


class defer_wgrad_reduce(object):
    def __init__(self, enabled=True):
        self.enabled = bool(enabled) and _os.environ.get('GGAN_NO_DEFER_WGRAD') is None

    def __enter__(self):
        if self.enabled:
            _DEFER[0] = {}
        return self

    def __exit__(self, et, ev, tb):
        reg, _DEFER[0] = _DEFER[0], None
        if self.enabled and et is None and reg:
            raise _lib.GganError('%d deferred filter-gradient slab sets were never packed: %s' % (len(reg), sorted((v[0], v[1]) for v in reg.values())))


def _wgrad_parts(x, gy, y, act, alpha, geom, with_bias):
    """Filter gradient via ggan_conv2d_bwd_filter_parts; None when deferral is off or the geometry is not covered."""
    reg = _DEFER[0]
    if reg is None:
        return None
    N, Ci, H, W, Co, Ho, Wo, k = geom[:8]
    if k != 5:
        return None
    elems = k * k * Ci * Co
    stride = elems + (Co if with_bias else 0)
    tiles = -(-Ci // 16) * -(-Co // 32)                      # (conv_wgrad.hip: 16 ci x 32 co per workgroup, split-K to ~256 workgroups, <= 64 slabs)
    cap = min(64, max(1, -(-256 // tiles))) * stride
    if Ci <= 4 and stride <= 8192 and N * Ho >= 8192:        # thin layers at 512..1024 frames: up to 256 small slabs (conv_thin.hip)
        cap = 256 * stride
    elif Ci <= 4 and stride <= 8192 and Co <= 32:            # ... a 32-channel thin layer: up to 128 (two column groups x 128 = the chip)
        cap = max(cap, 128 * stride)
    part = torch.empty((cap,), dtype=torch.float32, device=x.device)
    n, st = C.c_int(0), C.c_size_t(0)
    g = _geom(geom)
    rc = _L().ggan_conv2d_bwd_filter_parts(C.byref(g), _p(x), _p(gy), _p(y) if act != ACT_NONE else _p(None), act, alpha,
                                           1 if with_bias else 0, _p(part), cap, C.byref(n), C.byref(st), _stream())
    if rc == 1:
        return None
    check(rc, 'ggan_conv2d_bwd_filter_parts')
    gw = part[:elems].view(k, k, Ci, Co)
    gb = part[elems:elems + Co] if with_bias else None
    if n.value > 1:
        reg[gw.data_ptr()] = (n.value, st.value, part)
        if gb is not None:
            reg[gb.data_ptr()] = (n.value, st.value, part)
    return gw, gb


_STREAMS = {}


def shared_stream(device, role):
    """One HIP stream per (device, role) for the whole process ('capture': warm-up + graph capture, 'side': the second branch of a
    step graph).  HIP maps streams onto a handful of hardware queues round-robin: a fresh pair of streams per Trainer would, after a
    few Trainers in one process, put the two branches of a step graph on the SAME hardware queue, where they serialise (measured:
    a workload run after another one in the same process was 2-3 % slower than alone)."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device(), role)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def workspace(device):
    """Persistent split-K / filter-transpose scratch, one per (device, stream): kernels on one stream are serialised."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.GganError('workspace must be created before graph capture (run one eager warm-up step)')
        ws = torch.empty(_WS_BYTES, dtype=torch.uint8, device=device)
        ws[:65536].zero_()       # GGAN_WS_RESERVED: split-K arrival counters start (and are left) at zero
        _WS[key] = ws
    return ws


# ---------------------------------------------------------------------------------------------------
# geometry (TF padding arithmetic)
# ---------------------------------------------------------------------------------------------------
_TARGET = [0]
_SERIAL_BWD = [False]


class serial_backward(object):
    """with serial_backward(): the backward launches of layers recorded under target_workgroups take the DEFAULT plan -- for a backward
    pass whose two chains are NOT going to run side by side (the data-parallel generator step differentiates the Generator's and the
    Extractor's halves one after the other, so that the first gradient bucket can go on the wire early: engine._bwd_phase1 / 2)"""

    def __enter__(self):
        self.prev, _SERIAL_BWD[0] = _SERIAL_BWD[0], True

    def __exit__(self, *a):
        _SERIAL_BWD[0] = self.prev


def _bwd_target(ctx):
    return 0 if _SERIAL_BWD[0] else getattr(ctx, 'target', 0)


class target_workgroups(object):
    """with target_workgroups(n): the conv ops recorded inside plan their launches -- forward AND, later, backward -- for n workgroups
    (ggan_conv_geom.plan_wgs / plan_wgs_filter of their calls) instead of about one per CU: for layers of two chains that run side by side on two streams"""

    def __init__(self, n):
        self.n = int(n or 0)

    def __enter__(self):
        self.prev, _TARGET[0] = _TARGET[0], self.n

    def __exit__(self, *a):
        _TARGET[0] = self.prev


# The launch plan of a conv call (ggan_conv_geom.plan_wgs / plan_wgs_filter / plan_flags) is per CALL: it is filled into the geometry
# struct from this thread's current setting -- autograd runs backward nodes on worker threads, each with its own -- and the library keeps
# no process-wide plan (round 3 review: set / launch / restore sequences on C globals interleaved between threads).
import threading as _threading
_PLAN = _threading.local()
_HINT_FILTER = _os.environ.get('GGAN_HINT_FILTER', '1') != '0'
_PLAIN = [False]       # force_plain(): debug cross-check on the plain kernels (process-wide on purpose: a test switch)


def force_plain(on):
    """every conv call from now on asks for the plain one-thread-per-output kernels (GGAN_PLAN_PLAIN); returns the old setting"""
    old, _PLAIN[0] = _PLAIN[0], bool(on)
    return old


class launch_hint(object):
    """with launch_hint(n): the launches of conv calls made inside (and not under target_workgroups) plan for n workgroups
    (engine.Trainer._launch_hint: the wali-gp critic step, whose penalty pass runs beside the main pass)"""

    def __init__(self, n):
        self.n = int(n or 0)

    def __enter__(self):
        self.prev = getattr(_PLAN, 'hint', 0)
        _PLAN.hint = self.n

    def __exit__(self, *a):
        _PLAN.hint = self.prev


def _carries_hint(cls):
    """conv Functions: the launch hint in force when the layer was recorded also plans its backward launches -- autograd runs backward
    nodes on its own worker threads, where this thread's setting is not visible"""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._hint = getattr(_PLAN, 'hint', 0)
        return fwd(ctx, *args)

    def backward(ctx, *gs):
        with launch_hint(ctx._hint):
            return bwd(ctx, *gs)
    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls


class _planned_for(object):
    def __init__(self, n):
        self.n = int(n or 0)

    def __enter__(self):
        self.prev = getattr(_PLAN, 'both', 0)
        if self.n:
            _PLAN.both = self.n

    def __exit__(self, *a):
        _PLAN.both = self.prev


def same_geometry(size, k, stride, padding='SAME'):
    if padding == 'SAME':
        out = -(-size // stride)
        total = max((out - 1) * stride + k - size, 0)
        return out, total // 2
    if padding == 'VALID':
        return (size - k) // stride + 1, 0
    raise Exception('Unsupported configuration')


def conv_geom(N, Ci, H, W, Co, k, stride, padding='SAME'):
    Ho, pt = same_geometry(H, k, stride, padding)
    Wo, pl = same_geometry(W, k, stride, padding)
    return (N, Ci, H, W, Co, Ho, Wo, k, stride, pt, pl)


# ---- launch plans per launch SITE (round 6) ------------------------------------------------------------------------------------------
# target_workgroups / launch_hint plan a whole pass or step: every conv launch of a chain that runs beside another chain asks for ~128
# workgroups.  But a captured step graph is a fixed schedule, and in it some of those launches turn out to run ALONE (the head of the
# penalty chain, the tail of the longer backward pass: a third of the G+D+GP iteration had ONE half-chip kernel in flight, profiles/
# r05d_timeline.md).  A launch SITE is (scope, ordinal): the scope names the step of the iteration being built (engine.Trainer:
# 'gen0', 'disc0' .. 'disc4'), the ordinal counts the conv geometries built inside it in issue order -- forward launches on the
# recording thread, backward launches on autograd's worker thread for the device; the two never run at the same time and autograd's
# order is a function of the recorded graph, so the numbering is the same in the eager rehearsal, in the capture and in the next process.
# A site plan maps site -> (plan_wgs, plan_wgs_filter) and overrides whatever the pass-level settings say (-1 = leave that field);
# entries carry the geometry they were made for and are ignored (counted in site_mismatches) when the site holds another one.
_SITE = dict(scope=None, n=0, table={}, log=None, mismatches=0)


class site_scope(object):
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = (_SITE['scope'], _SITE['n'])
        _SITE['scope'], _SITE['n'] = self.name, 0

    def __exit__(self, *a):
        _SITE['scope'], _SITE['n'] = self.prev


def set_site_plan(table):
    """table: {'disc2:17': dict(geom=[N, Ci, H, W, Co], wgs=0, wgs_filter=-1), ...} or None"""
    _SITE['table'] = dict(table or {})
    _SITE['mismatches'] = 0


def record_sites(on=True):
    """start (or stop) logging every site passed: site_log() -> [(site, (N, Ci, H, W, Co, Ho, Wo), plan_wgs, plan_wgs_filter)]"""
    _SITE['log'] = [] if on else None


def site_log():
    return list(_SITE['log'] or [])


def site_mismatches():
    return _SITE['mismatches']


def _geom(t):
    both, hint = getattr(_PLAN, 'both', 0), getattr(_PLAN, 'hint', 0)
    # (the hint plans the filter gradient too since round 5: with the four-wave kernel 128 workgroups x 4 chunks beat 256 x 2 beside a
    #  second chain -- headline 4.29 -> 4.17 ms; GGAN_HINT_FILTER=0: filter gradients keep their default, as in rounds 3-4)
    wgs, wgs_f = (both or hint), (both or (hint if _HINT_FILTER else 0))
    scope = _SITE['scope']
    if scope is not None:
        site = '%s:%d' % (scope, _SITE['n'])
        _SITE['n'] += 1
        ov = _SITE['table'].get(site)
        if ov is not None:
            if list(ov.get('geom', t[:5])) != [int(v) for v in t[:5]]:
                _SITE['mismatches'] += 1
            else:
                wgs = ov['wgs'] if ov.get('wgs', -1) >= 0 else wgs
                wgs_f = ov['wgs_filter'] if ov.get('wgs_filter', -1) >= 0 else wgs_f
        if _SITE['log'] is not None:
            _SITE['log'].append((site, tuple(int(v) for v in t[:7]), int(wgs), int(wgs_f)))
    return ConvGeom(*(tuple(t[:11]) + (wgs, wgs_f, _lib.PLAN_PLAIN if _PLAIN[0] else 0)))


class RowSlot(object):
    """Rows [lo, hi) of a preallocated [rows, cols] buffer: where a producer is asked to leave its result so that a later
    row concatenation costs nothing (JoinRows).  Deliberately not a tensor: autograd sees the view a producer returns as a
    freshly created output."""

    def __init__(self, buf, lo, hi):
        assert buf.is_contiguous() and buf.dim() == 2 and 0 <= lo < hi <= buf.shape[0]
        self.buf, self.lo, self.hi = buf, lo, hi

    def take(self, shape):
        v = self.buf[self.lo:self.hi]
        n = 1
        for d in shape:
            n *= int(d)
        assert v.numel() == n and v.dtype == torch.float32, (tuple(v.shape), tuple(shape))
        return v.view(tuple(shape))


def _new_out(slot, shape, device):
    return slot.take(shape) if slot is not None else torch.empty(tuple(shape), dtype=torch.float32, device=device)


def _adjacent(a, b):
    return (a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and tuple(a.shape[1:]) == tuple(b.shape[1:])
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel())




HEAD_LOGITS = {}        # logits data pointer -> CriticHead record (CriticHead.forward registers, BceSum.forward consumes)

# A hinted critic head (head_bce_hint) owes its cost's VALUE until its backward launch: the cost tensor BceSum / MeanSum return is
# unwritten memory in between.  Whoever reads the value first settles the debt: the objectives call settle_cost() before any arithmetic
# on a cost (cost + s_f, / n, + rec_penalty), the head's backward clears it, and a Trainer step checks that none is left over.
_PENDING_COSTS = {}     # cost data pointer -> tail record (the head that carries the value)


def _tail_value(tail):
    """the cost launch a hinted head's tail record stands for (ggan_bce_logits_multi_fwd / ggan_mean_multi_fwd_grad), now"""
    nt = len(tail['terms'])
    lg = tail['logits']
    ext = tail.get('ext') or [None] * nt
    xs, o = [], 0
    for (n, _, _), e in zip(tail['terms'], ext):
        if e is not None:
            xs.append(e.data_ptr())
        else:
            xs.append(lg.data_ptr() + 4 * o)
            o += n
    pw, pn = (C.c_float * nt)(*[wt for _, _, wt in tail['terms']]), (C.c_int * nt)(*[n for n, _, _ in tail['terms']])
    if tail.get('kind') == 'mean':
        check(_L().ggan_mean_multi_fwd_grad((C.c_void_p * nt)(*xs), pw, pn, nt, _p(tail['loss']), None, _stream()), 'ggan_mean_multi_fwd_grad')
    else:
        check(_L().ggan_bce_logits_multi_fwd((C.c_void_p * nt)(*xs), (C.c_float * nt)(*[z for _, z, _ in tail['terms']]), pw, pn, nt,
                                             _p(tail['loss']), _stream()), 'ggan_bce_logits_multi_fwd')


def settle_cost(cost):
    """`cost` is about to be read by something else than the train op's backward: if a hinted critic head still owes its value, compute
    it now (one small launch; the head's backward then no longer writes it).  Returns cost."""
    if cost is None or not _PENDING_COSTS or not torch.is_tensor(cost):
        return cost
    tail = _PENDING_COSTS.pop(cost.data_ptr(), None)
    if tail is not None and tail.get('loss') is not None:
        _tail_value(tail)
        tail['loss'] = None
    return cost


def pending_costs():
    return len(_PENDING_COSTS)


# ---- one-element cost terms added LATE (round 6) ---------------------------------------------------------------------------------------
# The gradient penalty of a wali-gp critic step is a one-element term of its cost.  Riding in the hinted critic head's backward launch
# (which writes the cost's value) it made the [fake; real] pass's whole backward wait for the penalty chain's first-order phase -- the cost
# VALUE needs the penalty, the backward does not.  With LATE_EXT set (models.GraphicalGAN.forward, for a penalty that ran on a stream of its
# own) MeanSum leaves such terms out of the head's launch and registers them here; engine.Trainer adds them to the cost behind the backward
# pass (add_late_terms: one one-element launch per term, after the streams have been joined).  Timing probe: headline 3.81 -> 3.75 ms.
LATE_EXT = [False]
_LATE_TERMS = []        # (cost tensor [1], term tensor [1])
_READY = {}             # data pointer of a one-element term -> event recorded behind the launch that wrote it


def mark_ready(t):
    st = torch.cuda.current_stream(t.device)
    ev = torch.cuda.Event()
    ev.record(st)
    _READY[t.data_ptr()] = (ev, st)


def wait_ready(t):
    rec = _READY.get(t.data_ptr())
    if rec is not None:
        torch.cuda.current_stream(t.device).wait_event(rec[0])


def add_late_terms():
    """cost += every term registered for it.  The one-element launch goes to the stream that PRODUCED the term (the penalty's stream, idle by
    now) behind the current stream's position -- so it runs beside whatever the current stream does next (the update launch) instead of in
    front of it; returns the events the current stream has to wait for before anything reads the cost (engine.Trainer: behind the pack)."""
    evs = []
    while _LATE_TERMS:
        loss, term = _LATE_TERMS.pop(0)
        rec = _READY.pop(term.data_ptr(), None)
        cur = torch.cuda.current_stream(loss.device)
        st = rec[1] if rec is not None else cur
        if st.cuda_stream != cur.cuda_stream:
            here = torch.cuda.Event()
            here.record(cur)
            st.wait_event(here)                      # (the cost's own part was written by a launch of the current stream)
        with torch.cuda.stream(st):
            check(_L().ggan_axpby(_p(loss), _p(term), _p(loss), 1, 1.0, 1.0, 0.0, _stream()), 'ggan_axpby')
            if st.cuda_stream != cur.cuda_stream:
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
    return evs


def drop_late_terms():
    _LATE_TERMS.clear()
    _READY.clear()


def drop_pending_costs():
    _PENDING_COSTS.clear()
UNIT_SEEDS = {}         # data pointer -> the all-ones tensor an optimizer seeds d(cost)/d(cost) with (kept alive here: an address
                        # in this table can never belong to another tensor); emptied by optim.reset_optimizers


def unit_seed(like):
    """the persistent ones tensor an optimizer differentiates its cost with; registered so that ops whose forward already
    produced the gradients for a unit upstream gradient (BceSum) can recognise it"""
    one = torch.ones_like(like)
    UNIT_SEEDS[one.data_ptr()] = one
    return one


def is_unit_seed(g):
    """g IS a registered unit seed: same storage address AND still the tensor registered there, unmodified (a gradient that autograd
    accumulated in place into a buffer at that address has another shape / version and takes the backward kernel)"""
    one = UNIT_SEEDS.get(g.data_ptr())
    return one is not None and g.shape == one.shape and g._version == one._version and (g is one or g._base is one or g.untyped_storage().data_ptr() == one.untyped_storage().data_ptr())
