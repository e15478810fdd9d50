"""torch.autograd bindings of the libggan C ABI.

torch supplies device memory, the current HIP stream and the autograd tape (the role tf.gradients plays in
the reference); every piece of arithmetic is a HIP kernel reached through include/ggan.h.  Each backward
is itself built from these Functions, so the gradient-penalty double backward works:
  conv_fwd'   = (conv_dgrad, conv_wgrad)        conv_dgrad' = (conv_fwd, conv_wgrad)
  conv_wgrad' = (conv_dgrad, conv_fwd)          gemm'       = (gemm, gemm)
There is no CPU path: tensors must live on a HIP device and libggan.so must load.
"""
import ctypes as C
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID, ConvGeom, check  # noqa: F401

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


def _geom(t):
    both, hint = getattr(_PLAN, 'both', 0), getattr(_PLAN, 'hint', 0)
    # (the hint plans the filter gradient too since round 5: with the four-wave kernel 128 workgroups x 4 chunks beat 256 x 2 beside a
    #  second chain -- headline 4.29 -> 4.17 ms; GGAN_HINT_FILTER=0: filter gradients keep their default, as in rounds 3-4)
    return ConvGeom(*(tuple(t[:11]) + ((both or hint), (both or (hint if _HINT_FILTER else 0)), _lib.PLAN_PLAIN if _PLAIN[0] else 0)))



# ---------------------------------------------------------------------------------------------------
# convolution family
# ---------------------------------------------------------------------------------------------------
# tests: rows of a data-gradient that a grad_rows backward leaves unwritten are filled with NaN, so any consumer shows up
DEBUG_POISON_CHECK = bool(os.environ.get('GGAN_POISON_UNWRITTEN'))


class PendingCast(object):
    """The scaled minibatch real_x = mul*(float(x_int)/div - .5) (+ noise) of a device ring, NOT YET COMPUTED: `out` is the float tensor it
    will live in (tflib.ops.act.cast_scale(..., defer=True)).  The first consumer decides who writes it: ConvFwd on a thin-channel first
    layer forms it while it stages its input (ggan_conv2d_fwd_cast_ring: one launch instead of two); anything else calls materialize()
    (the plain ggan_cast_scale_ring_i32 launch).  Data, never differentiated."""

    def __init__(self, x_int, noise, div, mul, slot, ring):
        self.x_int, self.noise, self.div, self.mul, self.ring = x_int, noise, float(div), float(mul), ring
        self.out = _new_out(slot, x_int.shape, x_int.device)
        self.done = False
        self.shape = tuple(x_int.shape)

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        n = self.out.numel()
        known = 1
        for d in shape:
            known *= d if d != -1 else 1
        self.shape = tuple((n // known) if d == -1 else d for d in shape)
        return self

    def materialize(self):
        if not self.done:
            rt, ca, cb, off = self.ring
            nz = _p(_c(self.noise)) if self.noise is not None else _p(None)
            check(_L().ggan_cast_scale_ring_i32(_p(rt), rt.shape[0], _p(ca), _p(cb), int(off), nz, _p(self.out), self.out.numel(), self.div,
                                                self.mul, _stream()), 'ggan_cast_scale_ring_i32')
            self.done = True
        return self.out


@_carries_hint
@_skip_undefined
class ConvFwd(Function):
    """y = conv(x[N,Ci,H,W], w[k,k,Ci,Co]) + bias  (tf.nn.conv2d + bias_add; also the Deconv2D data-gradient)."""

    @staticmethod
    def forward(ctx, x, w, bias, geom, act, alpha, grad_rows=None):
        """grad_rows (optional): only images [0, grad_rows) of x need a gradient (the rest of the batch is data); a plain
        backward pass with frozen weights then runs the data-gradient on that sub-batch and leaves the other rows of
        the returned gradient unwritten -- the caller promises nothing reads them."""
        N, Ci, H, W, Co, Ho, Wo = geom[:7]
        pend = x if isinstance(x, PendingCast) else None
        if pend is not None:
            # the input is a minibatch still waiting in the device ring as int32: this layer scales it on the way in (one launch less at
            # the head of the Extractor chain) unless the geometry needs the float tensor first
            x = pend.out.view(N, Ci, H, W)
            if pend.done:
                pend = None
        x, w = _c(x), _c(w)
        assert tuple(x.shape) == (N, Ci, H, W) and tuple(w.shape) == (geom[7], geom[7], Ci, Co), (x.shape, w.shape, geom)
        ctx.grad_rows = int(grad_rows) if grad_rows else None
        ctx.target = _TARGET[0]
        y = torch.empty((N, Co, Ho, Wo), dtype=torch.float32, device=x.device)
        ws = workspace(x.device)
        with _planned_for(ctx.target):
            g = _geom(geom)
            bp = _p(_c(bias)) if bias is not None else _p(None)
            if pend is not None:
                rt, ca, cb, off = pend.ring
                nz = _p(_c(pend.noise)) if pend.noise is not None else _p(None)
                rc = _L().ggan_conv2d_fwd_cast_ring(C.byref(g), _p(rt), rt.shape[0], _p(ca), _p(cb), int(off), nz, pend.div, pend.mul,
                                                    _p(x), _p(w), bp, _p(y), act, alpha, _stream())
                if rc == 1:
                    pend.materialize()          # (geometry outside the thin-channel kernel: the two launches)
                    pend = None
                else:
                    check(rc, 'ggan_conv2d_fwd_cast_ring')
                    pend.done = True
            if pend is None:
                check(_L().ggan_conv2d_fwd(C.byref(g), _p(x), _p(w), bp, _p(y), act, alpha, _p(ws), ws.numel(), _stream()), 'ggan_conv2d_fwd')
        ctx.geom, ctx.act, ctx.alpha, ctx.has_bias = geom, act, alpha, bias is not None
        ctx.w_param, ctx.b_param = _is_param(w), _is_param(bias)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        if not torch.is_grad_enabled() and FUSED_CONV_BWD:
            # plain backward: two launches -- the activation derivative is applied while gy is staged and the bias
            # gradient comes out of the filter-gradient kernel (no act_bwd / chansum passes, no intermediate tensor)
            with _planned_for(_bwd_target(ctx)):
                r = _fused_conv_backward(ctx, gy, x, w, y)
            if r is not None:
                return r + (None,)
        want_w = ctx.needs_input_grad[1] and not (_DATA_ONLY[0] and ctx.w_param)
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not (_DATA_ONLY[0] and ctx.b_param)
        if (ctx.act in (ACT_LRELU, ACT_RELU) and torch.is_grad_enabled() and ctx.needs_input_grad[0] and not want_w and not want_b
                and not ctx.grad_rows and not _os.environ.get('GGAN_NO_DGRAD_MASKED')):
            # a double backward is being recorded and only the data gradient is asked for (the gradient-penalty pass): the activation
            # derivative rides in the data-gradient launch, and in its backward's launches (ConvDgradMasked)
            with _planned_for(_bwd_target(ctx)):        # (the plan remembered for this layer's backward launches)
                return ConvDgradMasked.apply(gy, y, w, ctx.geom, ctx.act, ctx.alpha), None, None, None, None, None, None
        if ctx.act != ACT_NONE:
            gy = ActBwd.apply(gy, y, ctx.act, ctx.alpha)       # lrelu/relu: sign(y) == sign(pre-activation)
        gx = gw = gb = None
        if ctx.needs_input_grad[1] and not (_DATA_ONLY[0] and ctx.w_param):
            gw = ConvWgrad.apply(x, gy, ctx.geom)
        if ctx.has_bias and ctx.needs_input_grad[2] and not (_DATA_ONLY[0] and ctx.b_param):
            gb = ChanSum.apply(gy)
        if ctx.needs_input_grad[0]:
            gx = ConvDgrad.apply(gy, w, None, ctx.geom, ACT_NONE, 0.0)
        return gx, gw, gb, None, None, None, None


def _fused_conv_backward(ctx, gy, x, w, y):
    gy = _c(gy)
    geom = ctx.geom
    N, Ci, H, W, Co, Ho, Wo, k = geom[:8]
    g = _geom(geom)
    L = _L()
    ws = workspace(gy.device)
    act = ctx.act
    yref = _p(y) if act != ACT_NONE else _p(None)
    gx = gw = gb = None
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        parts = _wgrad_parts(x, gy, y, act, ctx.alpha, geom, want_b)
        if parts is not None:
            gw, gb = parts
        else:
            gw = torch.empty((k, k, Ci, Co), dtype=torch.float32, device=gy.device)
            gb = torch.empty((Co,), dtype=torch.float32, device=gy.device) if want_b else None
            rc = L.ggan_conv2d_bwd_filter_act(C.byref(g), _p(x), _p(gy), yref, act, ctx.alpha, _p(gw), _p(gb), _p(ws), ws.numel(),
                                              _stream())
            if rc == 1:
                return None          # geometry not covered by the fused kernels: caller takes the unfused path
            check(rc, 'ggan_conv2d_bwd_filter_act')
        if not ctx.needs_input_grad[1]:
            gw = None
    if ctx.needs_input_grad[0]:
        gx = torch.empty((N, Ci, H, W), dtype=torch.float32, device=gy.device)
        if ctx.grad_rows and ctx.grad_rows < N and gw is None and gb is None:
            g = _geom((ctx.grad_rows,) + tuple(geom[1:]))      # leading images only (contiguous NCHW prefix of gy, y, gx)
            if DEBUG_POISON_CHECK:                               # tests: the unwritten rows must never be read downstream
                gx[ctx.grad_rows:].fill_(float('nan'))
        check(L.ggan_conv2d_bwd_data_act(C.byref(g), _p(gy), yref, act, ctx.alpha, _p(w), _p(gx), _p(ws), ws.numel(), _stream()),
              'ggan_conv2d_bwd_data_act')
    return gx, gw, gb, None, None, None


@_carries_hint
@_skip_undefined
class ConvDgrad(Function):
    """gx[N,Ci,H,W] = conv^T(gy[N,Co,Ho,Wo], w) + bias[Ci]  (Conv2DBackpropInput; also the Deconv2D forward)."""

    @staticmethod
    def forward(ctx, gy, w, bias, geom, act, alpha, slot=None):
        gy, w = _c(gy), _c(w)
        N, Ci, H, W, Co, Ho, Wo = geom[:7]
        assert tuple(gy.shape) == (N, Co, Ho, Wo) and tuple(w.shape) == (geom[7], geom[7], Ci, Co), (gy.shape, w.shape, geom)
        ctx.target = _TARGET[0]
        gx = _new_out(slot, (N, Ci, H, W), gy.device)
        ws = workspace(gy.device)
        with _planned_for(ctx.target):
            g = _geom(geom)
            check(_L().ggan_conv2d_bwd_data(C.byref(g), _p(gy), _p(w), _p(_c(bias)) if bias is not None else _p(None),
                                            _p(gx), act, alpha, _p(ws), ws.numel(), _stream()), 'ggan_conv2d_bwd_data')
        ctx.geom, ctx.act, ctx.alpha, ctx.has_bias = geom, act, alpha, bias is not None
        ctx.save_for_backward(gy, w, gx if act != ACT_NONE else None)
        return gx

    @staticmethod
    def backward(ctx, h):
        gy, w, out = ctx.saved_tensors
        d_gy = d_w = d_b = None
        reg = _DEFER[0]
        if (ctx.act != ACT_NONE and ctx.has_bias and ctx.needs_input_grad[2] and reg is not None and not torch.is_grad_enabled()
                and not _os.environ.get('GGAN_NO_ACT_CHANSUM')):
            # activation derivative and the bias gradient in ONE pass: the channel sums leave as partial slabs for the pack kernel
            h = _c(h)
            N, Cc = h.shape[0], h.shape[1]
            HW = h.numel() // (N * Cc)
            cap = min(64, max(1, -(-512 // Cc))) * Cc
            part = torch.empty((cap,), dtype=torch.float32, device=h.device)
            hm = torch.empty_like(h)
            n = C.c_int(0)
            check(_L().ggan_act_bwd_chansum(_p(h), _p(out), _p(hm), _p(part), cap, C.byref(n), N, Cc, HW, ctx.act, ctx.alpha, _stream()),
                  'ggan_act_bwd_chansum')
            h = hm
            d_b = part[:Cc]
            if n.value > 1:
                reg[d_b.data_ptr()] = (n.value, Cc, part)
        elif ctx.act != ACT_NONE:
            h = ActBwd.apply(h, out, ctx.act, ctx.alpha)
        tgt = _bwd_target(ctx)
        if ctx.needs_input_grad[1]:
            with _planned_for(tgt):
                parts = _wgrad_parts(_c(h), gy, None, ACT_NONE, 0.0, ctx.geom, False) if not torch.is_grad_enabled() else None
                d_w = parts[0] if parts is not None else ConvWgrad.apply(h, gy, ctx.geom)
        if ctx.has_bias and ctx.needs_input_grad[2] and d_b is None:
            pre = getattr(h, '_ggan_chansum', None) if not torch.is_grad_enabled() else None
            d_b = pre if (pre is not None and pre.numel() == h.shape[1]) else ChanSum.apply(h)
        if ctx.needs_input_grad[0]:
            with target_workgroups(tgt):
                d_gy = ConvFwd.apply(h, w, None, ctx.geom, ACT_NONE, 0.0)
        return (d_gy, d_w, d_b) + (None,) * (len(ctx.needs_input_grad) - 3)


@_carries_hint
@_skip_undefined
class ConvDgradMasked(Function):
    """gx = conv^T(gy * act'(yref), w): ActBwd + ConvDgrad as one differentiable op (ggan_conv2d_bwd_data_act stages gy through the
    mask).  Backward, for the gradient h arriving at gx: d_gy = conv(h, w) * act'(yref) (ggan_conv2d_fwd_masked: the mask in the
    epilogue; conv + act_bwd where that geometry has no such kernel), d_w = the filter gradient of (h, gy * act'(yref)) with the mask
    applied while gy is staged (ggan_conv2d_bwd_filter_act).  yref gets no gradient (lrelu / relu: zero almost everywhere)."""

    @staticmethod
    def forward(ctx, gy, yref, w, geom, act, alpha):
        gy, yref, w = _c(gy), _c(yref), _c(w)
        N, Ci, H, W, Co, Ho, Wo = geom[:7]
        assert tuple(gy.shape) == (N, Co, Ho, Wo) == tuple(yref.shape), (gy.shape, yref.shape, geom)
        gx = torch.empty((N, Ci, H, W), dtype=torch.float32, device=gy.device)
        ws = workspace(gy.device)
        ctx.target = getattr(_PLAN, 'both', 0)            # (set by the layer whose backward this op is; its own backward launches follow it)
        g = _geom(geom)
        check(_L().ggan_conv2d_bwd_data_act(C.byref(g), _p(gy), _p(yref), act, alpha, _p(w), _p(gx), _p(ws), ws.numel(), _stream()),
              'ggan_conv2d_bwd_data_act')
        ctx.geom, ctx.act, ctx.alpha = geom, act, alpha
        ctx.w_param = _is_param(w)
        ctx.save_for_backward(gy, yref, w)
        return gx

    @staticmethod
    def backward(ctx, h):
        gy, yref, w = ctx.saved_tensors
        geom = ctx.geom
        d_gy = d_w = None
        if torch.is_grad_enabled():              # (a third derivative: plain composition)
            gm = ActBwd.apply(gy, yref, ctx.act, ctx.alpha)
            if ctx.needs_input_grad[0]:
                d_gy = ActBwd.apply(ConvFwd.apply(h, w, None, geom, ACT_NONE, 0.0), yref, ctx.act, ctx.alpha)
            if ctx.needs_input_grad[2] and not (_DATA_ONLY[0] and ctx.w_param):
                d_w = ConvWgrad.apply(h, gm, geom)
            return d_gy, None, d_w, None, None, None
        h = _c(h)
        N, Ci, H, W, Co, Ho, Wo, k = geom[:8]
        with _planned_for(ctx.target):
            g = _geom(geom)
        L, ws = _L(), workspace(h.device)
        if ctx.needs_input_grad[0]:
            d_gy = torch.empty((N, Co, Ho, Wo), dtype=torch.float32, device=h.device)
            rc = L.ggan_conv2d_fwd_masked(C.byref(g), _p(h), _p(w), _p(d_gy), _p(yref), ctx.act, ctx.alpha, _p(ws), ws.numel(), _stream())
            if rc == 1:                          # no kernel fuses the mask for this geometry: the two launches
                check(L.ggan_conv2d_fwd(C.byref(g), _p(h), _p(w), _p(None), _p(d_gy), ACT_NONE, 0.0, _p(ws), ws.numel(), _stream()),
                      'ggan_conv2d_fwd')
                d_gy = ActBwd.apply(d_gy, yref, ctx.act, ctx.alpha)
            else:
                check(rc, 'ggan_conv2d_fwd_masked')
        if ctx.needs_input_grad[2] and not (_DATA_ONLY[0] and ctx.w_param):
            parts = _wgrad_parts(h, gy, yref, ctx.act, ctx.alpha, geom, False)
            if parts is not None:
                d_w = parts[0]
            else:
                d_w = torch.empty((k, k, Ci, Co), dtype=torch.float32, device=h.device)
                rc = L.ggan_conv2d_bwd_filter_act(C.byref(g), _p(h), _p(gy), _p(yref), ctx.act, ctx.alpha, _p(d_w), _p(None), _p(ws),
                                                  ws.numel(), _stream())
                if rc == 1:
                    d_w = ConvWgrad.apply(h, ActBwd.apply(gy, yref, ctx.act, ctx.alpha), geom)
                else:
                    check(rc, 'ggan_conv2d_bwd_filter_act')
        return d_gy, None, d_w, None, None, None


@_carries_hint
@_skip_undefined
class ConvWgrad(Function):
    """gw[k,k,Ci,Co] = sum_n,oh,ow x (*) gy  (Conv2DBackpropFilter)."""

    @staticmethod
    def forward(ctx, x, gy, geom):
        x, gy = _c(x), _c(gy)
        N, Ci, H, W, Co, Ho, Wo, k = geom[:8]
        assert tuple(x.shape) == (N, Ci, H, W) and tuple(gy.shape) == (N, Co, Ho, Wo), (x.shape, gy.shape, geom)
        gw = torch.empty((k, k, Ci, Co), dtype=torch.float32, device=x.device)
        ws = workspace(x.device)
        g = _geom(geom)
        check(_L().ggan_conv2d_bwd_filter(C.byref(g), _p(x), _p(gy), _p(gw), _p(None), _p(ws), ws.numel(), _stream()),
              'ggan_conv2d_bwd_filter')
        ctx.geom = geom
        ctx.save_for_backward(x, gy)
        return gw

    @staticmethod
    def backward(ctx, h):
        x, gy = ctx.saved_tensors
        d_x = d_gy = None
        if ctx.needs_input_grad[0]:
            d_x = ConvDgrad.apply(gy, h, None, ctx.geom, ACT_NONE, 0.0)
        if ctx.needs_input_grad[1]:
            d_gy = ConvFwd.apply(x, h, None, ctx.geom, ACT_NONE, 0.0)
        return d_x, d_gy, None


@_skip_undefined
class ChanSum(Function):
    """out[c] = sum_{n,h,w} x[n,c,h,w]  (BiasAddGrad, NCHW)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        out = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        ws = workspace(x.device)
        check(_L().ggan_chansum(_p(x), _p(out), N, Cc, HW, _p(ws), ws.numel(), _stream()), 'ggan_chansum')
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        shp = ctx.shape
        return g.view(1, -1, *([1] * (len(shp) - 2))).expand(shp).contiguous()


# Linear backward on a tall operand (the Conv3D patch matrices, 10^4..10^6 rows): the single-launch fused paths below keep the whole
# row range in one workgroup per output tile, right for minibatch-sized operands and serial for these -- tall operands take the split-K
# GEMM and the slab column sum instead.
TALL_ROWS = 8192


@_skip_undefined
class ColSum(Function):
    """out[c] = sum_r x[r,c]  (BiasAddGrad of Linear)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        rows, cols = x.shape
        out = torch.empty((cols,), dtype=torch.float32, device=x.device)
        if rows >= TALL_ROWS:
            ws = workspace(x.device)
            check(_L().ggan_colsum_tall(_p(x), _p(out), rows, cols, _p(ws), ws.numel(), _stream()), 'ggan_colsum_tall')
        else:
            check(_L().ggan_colsum(_p(x), _p(out), rows, cols, _stream()), 'ggan_colsum')
        ctx.rows = rows
        return out

    @staticmethod
    def backward(ctx, g):
        return g.view(1, -1).expand(ctx.rows, g.numel()).contiguous()


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


# ---------------------------------------------------------------------------------------------------
# pointwise
# ---------------------------------------------------------------------------------------------------
@_skip_undefined
class ActFwd(Function):
    @staticmethod
    def forward(ctx, x, act, alpha):
        x = _c(x)
        y = torch.empty_like(x)
        check(_L().ggan_act_fwd(_p(x), _p(y), x.numel(), act, alpha, _stream()), 'ggan_act_fwd')
        ctx.act, ctx.alpha = act, alpha
        ctx.save_for_backward(y if act in (ACT_TANH, ACT_SIGMOID) else x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (ref,) = ctx.saved_tensors
        return ActBwd.apply(gy, ref, ctx.act, ctx.alpha), None, None


@_skip_undefined
class ActBwd(Function):
    """gx = gy * act'(ref); ref = forward input (lrelu/relu: only its sign is used, so the forward output
    works too) or forward output (tanh/sigmoid)."""

    @staticmethod
    def forward(ctx, gy, ref, act, alpha):
        gy, ref = _c(gy), _c(ref)
        gx = torch.empty_like(gy)
        check(_L().ggan_act_bwd(_p(gy), _p(ref), _p(gx), gy.numel(), act, alpha, _stream()), 'ggan_act_bwd')
        ctx.act, ctx.alpha = act, alpha
        ctx.save_for_backward(gy, ref)
        return gx

    @staticmethod
    def backward(ctx, h):
        gy, ref = ctx.saved_tensors
        d_gy = ActBwd.apply(h, ref, ctx.act, ctx.alpha) if ctx.needs_input_grad[0] else None
        d_ref = None
        if ctx.needs_input_grad[1] and ctx.act in (ACT_TANH, ACT_SIGMOID):
            # off the hot path (second-order term through a smooth activation); piecewise-linear
            # activations have zero second derivative a.e.
            if ctx.act == ACT_TANH:
                d_ref = h * gy * (-2.0 * ref)
            else:
                d_ref = h * gy * (1.0 - 2.0 * ref)
        return d_gy, d_ref, None, None


def leaky_relu(x, alpha=0.2):
    return ActFwd.apply(x, ACT_LRELU, float(alpha))


def relu(x):
    return ActFwd.apply(x, ACT_RELU, 0.0)


def tanh(x):
    return ActFwd.apply(x, ACT_TANH, 0.0)


def sigmoid(x):
    return ActFwd.apply(x, ACT_SIGMOID, 0.0)


@_skip_undefined
class BatchNormTrain(Function):
    """Training-mode BN over all axes but channel axis 1 (NCHW) or over axis 0 of [N,C]."""

    @staticmethod
    def forward(ctx, x, scale, offset, eps, act, alpha):
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        sc, of = _c(scale).reshape(-1), _c(offset).reshape(-1)
        y = torch.empty_like(x)
        mean = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        check(_L().ggan_bn_fwd_train(_p(x), _p(sc), _p(of), _p(y), _p(mean), _p(invstd), N, Cc, HW, eps, act, alpha,
                                     _stream()), 'ggan_bn_fwd_train')
        ctx.dims = (N, Cc, HW)
        ctx.act, ctx.alpha = act, alpha
        ctx.pshape = tuple(scale.shape)
        ctx.save_for_backward(x, sc, mean, invstd, y if act != ACT_NONE else None, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, sc, mean, invstd, y, scale = ctx.saved_tensors
        N, Cc, HW = ctx.dims
        if torch.is_grad_enabled():        # a double backward is being recorded (gradient penalty through this layer)
            gx, gs, go = BatchNormBwd.apply(x, gy, y, scale.reshape(-1), mean, invstd, ctx.act, ctx.alpha, ctx.dims)
            return gx, gs.view(ctx.pshape), go.view(ctx.pshape), None, None, None
        gy = _c(gy)
        gx = torch.empty_like(x)
        gs = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        go = torch.empty_like(gs)
        csum = torch.empty_like(gs) if HW > 1 else None
        check(_L().ggan_bn_bwd_act(_p(x), _p(gy), _p(y) if ctx.act != ACT_NONE else _p(None), ctx.act, ctx.alpha, _p(sc),
                                   _p(mean), _p(invstd), _p(gx), _p(gs), _p(go), _p(csum), N, Cc, HW, _stream()),
              'ggan_bn_bwd_act')
        if csum is not None:
            gx._ggan_chansum = csum      # picked up by the producing layer's backward if gx reaches it unchanged
        return gx, gs.view(ctx.pshape), go.view(ctx.pshape), None, None, None


class BatchNormBwd(Function):
    """The first backward of BatchNormTrain as a differentiable op: only on the tape while a double backward is recorded (MODE
    vegan-wgan-gp: gradient penalty on a critic with BatchNorm).  Its own backward (ggan_bn_bwd_bwd) covers gradients arriving
    at gx; the scale / offset gradients are not differentiated again (no objective of the reference needs that)."""

    @staticmethod
    def forward(ctx, x, gy, y, sc, mean, invstd, act, alpha, dims):
        N, Cc, HW = dims
        gy = _c(gy)
        gx = torch.empty_like(x)
        gs = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        go = torch.empty_like(gs)
        check(_L().ggan_bn_bwd_act(_p(x), _p(gy), _p(y) if act != ACT_NONE else _p(None), act, alpha, _p(sc), _p(mean), _p(invstd),
                                   _p(gx), _p(gs), _p(go), _p(None), N, Cc, HW, _stream()), 'ggan_bn_bwd_act')
        ctx.dims, ctx.act, ctx.alpha = dims, act, alpha
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, gy, y, sc, mean, invstd)
        return gx, gs, go

    @staticmethod
    @once_differentiable
    def backward(ctx, h, hs, ho):
        if hs is not None or ho is not None:
            raise NotImplementedError('second derivative of the BatchNorm scale/offset gradients')
        if h is None:
            return (None,) * 9
        x, gy, y, sc, mean, invstd = ctx.saved_tensors
        N, Cc, HW = ctx.dims
        h = _c(h)
        ggy, gx2 = torch.empty_like(x), torch.empty_like(x)
        gsc = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        check(_L().ggan_bn_bwd_bwd(_p(x), _p(gy), _p(y) if ctx.act != ACT_NONE else _p(None), ctx.act, ctx.alpha, _p(h), _p(sc),
                                   _p(mean), _p(invstd), _p(ggy), _p(gx2), _p(gsc), N, Cc, HW, _stream()), 'ggan_bn_bwd_bwd')
        return gx2, ggy, None, gsc.view(sc.shape), None, None, None, None, None


class LinearBatchNormRows(Function):
    """y = act(BN_rows(x @ W + b)) in ONE launch (ggan_linear_bn_rows_fwd): Linear 'Generator.Input' + Batchnorm 'Generator.BN1' over the
    batch axis + relu (gan_inference_cifar10.py:134-138).  The backward is the composition's: ggan_bn_bwd_act on the kept Linear
    output, then the Linear layer's gradients (dW and db in one launch, the data gradient when the input needs one)."""

    @staticmethod
    def usable(x, w):
        """mirrors the limits of ggan_linear_bn_rows_fwd (linear_bn.hip): whole minibatch in one workgroup's LDS -- x [M, K + 4], the weight
        slice [K, 32] and the row-group partials must fit 160 KB (M = 128 with K >= 252 does not) -- rows in 16 equal groups, 16-byte
        aligned operands; anything else takes Linear + Batchnorm"""
        if x.dim() != 2 or _os.environ.get('GGAN_NO_LINEAR_BN'):
            return False
        M, K = x.shape
        N = w.shape[1]
        lds = (M * (K + 4) + K * 32 + 16 * 32) * 4
        return (M <= 128 and M % 16 == 0 and K <= 256 and K % 4 == 0 and N % 32 == 0 and x.is_contiguous() and w.is_contiguous()
                and lds <= 160 * 1024 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, x, w, b, scale, offset, eps, act, alpha):
        x, w = _c(x), _c(w)
        M, K = x.shape
        N = w.shape[1]
        sc, of = _c(scale).reshape(-1), _c(offset).reshape(-1)
        bp = _p(_c(b)) if b is not None else _p(None)
        h = torch.empty((M, N), dtype=torch.float32, device=x.device)
        y = torch.empty_like(h)
        mean = torch.empty((N,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        check(_L().ggan_linear_bn_rows_fwd(_p(x), _p(w), bp, _p(sc), _p(of), _p(h), _p(y), _p(mean),
                                           _p(invstd), M, K, N, eps, act, alpha, _stream()), 'ggan_linear_bn_rows_fwd')
        ctx.act, ctx.alpha, ctx.has_bias, ctx.pshape = act, alpha, b is not None, tuple(scale.shape)
        ctx.save_for_backward(x, w, h, sc, mean, invstd, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, h, sc, mean, invstd, y = ctx.saved_tensors
        M, N = h.shape
        gy = _c(gy)
        gs = torch.empty((N,), dtype=torch.float32, device=h.device)
        go = torch.empty_like(gs)
        need = ctx.needs_input_grad
        if (not need[0] and need[1] and mean.dim() == 1 and x.shape[1] in (64, 128, 256) and M <= 128 and M % 16 == 0 and N % 32 == 0
                and x.data_ptr() % 16 == 0 and not _os.environ.get('GGAN_NO_LINEAR_BN_BWD')):
            # the input is noise (no data gradient): BatchNorm's backward and the weight-gradient product in one launch
            dw = torch.empty((x.shape[1], N), dtype=torch.float32, device=h.device)
            db = torch.empty((N,), dtype=torch.float32, device=h.device) if (ctx.has_bias and need[2]) else None
            rc = _L().ggan_linear_bn_rows_bwd(_p(x), _p(gy), _p(h), _p(y) if ctx.act != ACT_NONE else _p(None), _p(sc), _p(mean), _p(invstd),
                                              _p(dw), _p(db), _p(gs), _p(go), M, x.shape[1], N, ctx.act, ctx.alpha, _stream())
            if rc != 1:
                check(rc, 'ggan_linear_bn_rows_bwd')
                return (None, dw, db, gs.view(ctx.pshape) if need[3] else None, go.view(ctx.pshape) if need[4] else None, None, None, None)
        gh = torch.empty_like(h)
        check(_L().ggan_bn_bwd_act(_p(h), _p(gy), _p(y) if ctx.act != ACT_NONE else _p(None), ctx.act, ctx.alpha, _p(sc), _p(mean),
                                   _p(invstd), _p(gh), _p(gs), _p(go), _p(None), M, N, 1, _stream()), 'ggan_bn_bwd_act')
        dx = dw = db = None
        if need[1] or (ctx.has_bias and need[2]):
            dw, db = gemm_colsum_(x, gh, True)                         # dW = x^T gh and db = column sums of gh in one launch
            if not need[1]:
                dw = None
            if not (ctx.has_bias and need[2]):
                db = None
        if need[0]:
            dx = Gemm.apply(gh, w, None, False, True, ACT_NONE, 0.0)    # gh W^T
        return (dx, dw, db, gs.view(ctx.pshape) if need[3] else None, go.view(ctx.pshape) if need[4] else None, None, None, None)


def _all_gather_rows(t, group):
    """[world, *t.shape]: every replica's `t`, in rank order"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    from . import rccl
    # (rccl.get_stats, called by Trainer(sync_bn=True): the gradient buckets' communicator in serial mode -- this gather waits for a bucket
    #  that is still on the wire on the communicator's stream, and the next bucket waits for it)
    comm = rccl._STATS[0] if (t.is_cuda and (group is None or group is dist.group.WORLD)) else None
    if comm is not None:             # an enqueue on the current stream (capturable: cross-replica BatchNorm inside a step graph)
        return comm.all_gather(out, t.contiguous())
    try:
        dist.all_gather_into_tensor(out, t, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        out = torch.stack(parts)
    return out


class SyncBatchNormTrain(Function):
    """BatchNormTrain with statistics over the global batch of a process group of equal-sized replicas: local
    statistics kernel -> all-gather of 2*C floats -> normalisation kernel, the same split in the backward."""

    @staticmethod
    def forward(ctx, x, scale, offset, eps, act, alpha, group):
        import torch.distributed as dist
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        y = torch.empty_like(x)
        mean = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        sc, of = _c(scale).reshape(-1), _c(offset).reshape(-1)
        st = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        check(_L().ggan_bn_sync_stats(_p(x), _p(st), N, Cc, HW, _stream()), 'ggan_bn_sync_stats')
        allst = _all_gather_rows(st, group)
        world = allst.shape[0]
        check(_L().ggan_bn_sync_apply(_p(x), _p(allst), world, _p(sc), _p(of), _p(y), _p(mean), _p(invstd), N, Cc, HW, eps, act,
                                      alpha, _stream()), 'ggan_bn_sync_apply')
        ctx.dims = (N, Cc, HW)
        ctx.act, ctx.alpha = act, alpha
        ctx.group, ctx.world, ctx.rank = group, world, dist.get_rank(group)
        ctx.pshape = tuple(scale.shape)
        ctx.save_for_backward(x, sc, mean, invstd, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, sc, mean, invstd, y = ctx.saved_tensors
        N, Cc, HW = ctx.dims
        gy = _c(gy)
        yp = _p(y) if ctx.act != ACT_NONE else _p(None)
        sums = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        check(_L().ggan_bn_sync_bwd_stats(_p(x), _p(gy), yp, ctx.act, ctx.alpha, _p(mean), _p(invstd), _p(sums), N, Cc, HW,
                                          _stream()), 'ggan_bn_sync_bwd_stats')
        allsums = _all_gather_rows(sums, ctx.group)
        gx = torch.empty_like(x)
        gs = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        go = torch.empty_like(gs)
        check(_L().ggan_bn_sync_bwd_apply(_p(x), _p(gy), yp, ctx.act, ctx.alpha, _p(sc), _p(mean), _p(invstd), _p(allsums),
                                          ctx.world, ctx.rank, _p(gx), _p(gs), _p(go), N, Cc, HW, _stream()),
              'ggan_bn_sync_bwd_apply')
        return gx, gs.view(ctx.pshape), go.view(ctx.pshape), None, None, None, None


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


class JoinRows(Function):
    """cat([a, b], 0) for the critic evaluated once on [fake; real].  When the two operands already sit back to back in one
    buffer (their producers were handed RowSlots) the result is an alias of that memory: no copy kernel; the backward hands
    out the two row ranges of the incoming gradient (views)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.n = a.shape[0]
        if _adjacent(a, b):
            out = torch.empty(0, dtype=a.dtype, device=a.device)
            out.set_(a.untyped_storage(), a.storage_offset(), (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())
            return out
        return torch.cat([a, b], 0)

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.n], g[ctx.n:]


class Fanout(Function):
    """n aliases of a tensor that several branches of a step read (the mixture scripts: the code p_z feeds the Generator and the
    critics, q_z the mixture posterior and the critics, [p_z; q_z] both critics, the component means both hyper nets).  Their
    gradients are summed HERE, in alias order, by this library's pointwise launch (ggan_axpby through Axpby, differentiable) --
    not by at::add wherever autograd happens to meet the second contribution."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)      # (an alias nobody differentiates contributes None, not a zero tensor + an addition launch)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            acc = Axpby.apply(acc, g, 1.0, 1.0, 0.0)
        return acc, None


def fanout(x, n=2):
    """n aliases of x whose gradients meet in one Fanout node (x itself n times where no gradient can flow)"""
    if (not torch.is_tensor(x) or not x.is_cuda or not x.requires_grad or not torch.is_grad_enabled() or os.environ.get('GGAN_NO_FANOUT')):
        return (x,) * n
    return Fanout.apply(x, n)


class SplitRows(Function):
    """(x[:n], x[n:]) for the critic evaluated once on [fake; real]; the backward is ONE concatenation instead of two
    zero-padded slice gradients and their sum."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.rows = n, x.shape[0]
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        like = ga if ga is not None else gb
        if ga is None:
            ga = like.new_zeros((ctx.n,) + tuple(like.shape[1:]))
        if gb is None:
            gb = like.new_zeros((ctx.rows - ctx.n,) + tuple(like.shape[1:]))
        if (ga.is_contiguous() and gb.is_contiguous() and ga.dtype == gb.dtype
                and ga.untyped_storage().data_ptr() == gb.untyped_storage().data_ptr()
                and gb.storage_offset() == ga.storage_offset() + ga.numel()):
            # the two halves already sit back to back in one buffer (BceSum.backward): no copy
            return torch.as_strided(ga, (ctx.rows,) + tuple(ga.shape[1:]), ga.stride(), ga.storage_offset()), None
        return torch.cat([ga, gb], 0), None


@_skip_undefined
class CastScaleI32(Function):
    """real_x = mul*(float(x)/div - .5) + noise  (no gradient: the input is data)."""

    @staticmethod
    def forward(ctx, x_int, noise, div, mul, slot=None, ring=None):
        """ring: (int32 [R, ...] tensor of pre-staged minibatches, counter a, counter b, offset) -- the minibatch is slot
        (a + b + offset) mod R of the ring instead of x_int (ggan_cast_scale_ring_i32)"""
        _dev(x_int)
        assert x_int.dtype == torch.int32
        x_int = x_int.contiguous()
        y = _new_out(slot, x_int.shape, x_int.device)
        nz = _p(_c(noise)) if noise is not None else _p(None)
        if ring is not None:
            rt, ca, cb, off = ring
            assert rt.dtype == torch.int32 and rt.is_contiguous() and rt[0].numel() == x_int.numel()
            check(_L().ggan_cast_scale_ring_i32(_p(rt), rt.shape[0], _p(ca), _p(cb), int(off), nz, _p(y), x_int.numel(), div, mul,
                                                _stream()), 'ggan_cast_scale_ring_i32')
            return y
        check(_L().ggan_cast_scale_i32(_p(x_int), nz, _p(y), x_int.numel(), div, mul, _stream()), 'ggan_cast_scale_i32')
        return y

    @staticmethod
    def backward(ctx, g):
        return (None,) * len(ctx.needs_input_grad)


@_skip_undefined
class Axpby(Function):
    """out = a*x + b*y + c"""

    @staticmethod
    def forward(ctx, x, y, a, b, c, slot=None):
        x = _c(x)
        y = _c(y) if y is not None else None
        out = _new_out(slot, x.shape, x.device)
        check(_L().ggan_axpby(_p(x), _p(y), _p(out), x.numel(), a, b, c, _stream()), 'ggan_axpby')
        ctx.a, ctx.b, ctx.has_y = a, b, y is not None
        return out

    @staticmethod
    def backward(ctx, g):
        gx = Axpby.apply(g, None, ctx.a, 0.0, 0.0) if ctx.needs_input_grad[0] else None
        gy = Axpby.apply(g, None, ctx.b, 0.0, 0.0) if (ctx.has_y and ctx.needs_input_grad[1]) else None
        return (gx, gy) + (None,) * (len(ctx.needs_input_grad) - 2)


@_skip_undefined
class MixMean(Function):
    """p_z[B,D] = k[B,K] @ mu[K,D] + noise[B,D]: HyperGenerator of the gmgan scripts (gmgan_inference_cifar10.py:150-153) as ONE pointwise
    launch (ggan_mix_mean) instead of Gemm + Axpby at the head of the Generator chain; backward: d mu = k^T g (one product), d noise = g."""

    @staticmethod
    def usable(k, mu, noise):
        return k.dim() == 2 and mu.dim() == 2 and mu.shape[1] % 4 == 0 and k.is_cuda

    @staticmethod
    def forward(ctx, k, mu, noise, slot=None):
        k, mu, noise = _c(k), _c(mu), _c(noise)
        B, K = k.shape
        D = mu.shape[1]
        assert mu.shape[0] == K and tuple(noise.shape) == (B, D), (k.shape, mu.shape, noise.shape)
        out = _new_out(slot, (B, D), k.device)
        check(_L().ggan_mix_mean(_p(k), _p(mu), _p(noise), _p(out), B, K, D, _stream()), 'ggan_mix_mean')
        ctx.mu_param = _is_param(mu)
        ctx.save_for_backward(k, mu)
        return out

    @staticmethod
    def backward(ctx, g):
        k, mu = ctx.saved_tensors
        dk = dmu = None
        if ctx.needs_input_grad[1] and not (_DATA_ONLY[0] and ctx.mu_param):
            dmu = Gemm.apply(k, g, None, True, False, ACT_NONE, 0.0)          # k^T g
        if ctx.needs_input_grad[0]:
            dk = Gemm.apply(g, mu, None, False, True, ACT_NONE, 0.0)          # g mu^T
        return dk, dmu, (g if ctx.needs_input_grad[2] else None), None


class GmmLatent(Function):
    """HyperExtractor of the gmgan scripts in one launch per direction (ggan_gmm_latent_*): component logits of z under the
    mixture prior and the Gumbel-softmax relaxation of the component assignment.  Returns (logits, k)."""

    @staticmethod
    def forward(ctx, z, mu, gumbel_u, log_pi, temp, slot=None):
        z, mu, gumbel_u = _c(z), _c(mu), _c(gumbel_u)
        B, D = z.shape
        K = mu.shape[0]
        assert tuple(mu.shape) == (K, D) and tuple(gumbel_u.shape) == (B, K)
        logits = torch.empty((B, K), dtype=torch.float32, device=z.device)
        k = _new_out(slot, (B, K), z.device)
        check(_L().ggan_gmm_latent_fwd(_p(z), _p(mu), _p(gumbel_u), _p(logits), _p(k), B, K, D, float(log_pi), float(temp),
                                       _stream()), 'ggan_gmm_latent_fwd')
        ctx.temp = float(temp)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, mu, k)
        return logits, k

    @staticmethod
    @once_differentiable
    def backward(ctx, g_logits, g_k):
        z, mu, k = ctx.saved_tensors
        n_in = len(ctx.needs_input_grad)
        if (g_logits is None and g_k is None) or not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return (None,) * n_in
        B, D = z.shape
        K = mu.shape[0]
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        dmu = torch.empty_like(mu) if ctx.needs_input_grad[1] else None
        gl = _c(g_logits) if g_logits is not None else None
        gk = _c(g_k) if g_k is not None else None
        check(_L().ggan_gmm_latent_bwd(_p(z), _p(mu), _p(k), _p(gl), _p(gk), _p(dz), _p(dmu), B, K, D, ctx.temp, _stream()),
              'ggan_gmm_latent_bwd')
        return (dz, dmu) + (None,) * (n_in - 2)


class MixRbfMmd2(Function):
    """biased MMD^2 between two sets of codes under a mixture of RBF kernels (tflib/objs/mmd.py:65-67) -> 0-dim tensor"""

    @staticmethod
    def forward(ctx, x, y, sigmas, wts):
        x, y = _c(x), _c(y)
        m, d = x.shape
        n = y.shape[0]
        assert y.shape[1] == d
        ns = len(sigmas)
        sg = (C.c_float * ns)(*[float(v) for v in sigmas])
        wt = (C.c_float * ns)(*[float(v) for v in wts]) if wts is not None else None
        out = torch.empty((), dtype=torch.float32, device=x.device)
        scratch = torch.empty((m + n,), dtype=torch.float32, device=x.device)
        check(_L().ggan_mix_rbf_mmd2_fwd(_p(x), _p(y), m, n, d, sg, wt, ns, _p(out), _p(scratch), _stream()), 'ggan_mix_rbf_mmd2_fwd')
        ctx.sg, ctx.wt, ctx.ns = sg, wt, ns
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        m, d = x.shape
        n = y.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if dx is None and dy is None:
            return None, None, None, None
        check(_L().ggan_mix_rbf_mmd2_bwd(_p(x), _p(y), m, n, d, ctx.sg, ctx.wt, ctx.ns, _p(_c(g)), _p(dx), _p(dy), _stream()),
              'ggan_mix_rbf_mmd2_bwd')
        return dx, dy, None, None


class Reparam(Function):
    """(z, std) = (mean + eps * exp(log_std), exp(log_std)): the stochastic encoder head (gan_inference_cifar10.py:173-188)"""

    @staticmethod
    def forward(ctx, mean, log_std, eps):
        mean, log_std, eps = _c(mean), _c(log_std), _c(eps)
        z, sd = torch.empty_like(mean), torch.empty_like(mean)
        check(_L().ggan_reparam_fwd(_p(mean), _p(log_std), _p(eps), _p(z), _p(sd), mean.numel(), _stream()), 'ggan_reparam_fwd')
        ctx.save_for_backward(eps, sd)
        return z, sd

    @staticmethod
    @once_differentiable
    def backward(ctx, gz, gsd):
        eps, sd = ctx.saved_tensors
        gmean, glog = torch.empty_like(sd), torch.empty_like(sd)
        check(_L().ggan_reparam_bwd(_p(_c(gz)) if gz is not None else _p(None), _p(_c(gsd)) if gsd is not None else _p(None), _p(eps), _p(sd),
                                    _p(gmean), _p(glog), sd.numel(), _stream()), 'ggan_reparam_bwd')
        return gmean, glog, None


AGG_KL, AGG_IKL, AGG_JSD = 0, 1, 2


class AggDiv(Function):
    """Monte-Carlo KL / inverse KL / JSD between the aggregated posterior (mixture of the minibatch's diagonal Gaussians mu, sd [nx, d])
    and N(0, I) (tflib/objs/kl_aggregated.py:46-74) -> 0-dim tensor.  k_onehot [nz, nx], eps_q [nz, d]: the component draws and noise
    of the samples from q (kl, jsd); z_p [nz, d]: the samples from the prior (ikl, jsd)."""

    @staticmethod
    def forward(ctx, mu, sd, k_onehot, eps_q, z_p, kind, n_coms):
        mu, sd = _c(mu), _c(sd)
        nx, d = mu.shape
        nz = (z_p if kind != AGG_KL else eps_q).shape[0]
        ns = 2 * nz if kind == AGG_JSD else nz
        k_onehot = _c(k_onehot) if kind != AGG_IKL else None
        eps_q = _c(eps_q) if kind != AGG_IKL else None
        z_p = _c(z_p) if kind != AGG_KL else None
        assert k_onehot is None or (tuple(k_onehot.shape) == (nz, nx) and tuple(eps_q.shape) == (nz, d))
        assert z_p is None or tuple(z_p.shape) == (nz, d)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=mu.device)
        out, Z, A, Bv, T = new(), new(ns, d), new(ns, nx), new(ns), new(ns)
        check(_L().ggan_agg_div_fwd(kind, _p(mu), _p(sd), _p(k_onehot), _p(eps_q), _p(z_p), nx, nz, d, int(n_coms), _p(out), _p(Z), _p(A),
                                    _p(Bv), _p(T), _stream()), 'ggan_agg_div_fwd')
        ctx.dims = (kind, nx, nz, d, int(n_coms), ns)
        ctx.save_for_backward(mu, sd, k_onehot, eps_q, Z, A, Bv)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        mu, sd, k_onehot, eps_q, Z, A, Bv = ctx.saved_tensors
        kind, nx, nz, d, n_coms, ns = ctx.dims
        gmu, gsd = torch.empty_like(mu), torch.empty_like(sd)
        W = torch.empty((ns, nx), dtype=torch.float32, device=mu.device)
        GZ = torch.empty((nz, d), dtype=torch.float32, device=mu.device)
        check(_L().ggan_agg_div_bwd(kind, _p(mu), _p(sd), _p(k_onehot), _p(eps_q), nx, nz, d, n_coms, _p(Z), _p(A), _p(Bv), _p(_c(g)),
                                    _p(W), _p(GZ), _p(gmu), _p(gsd), _stream()), 'ggan_agg_div_bwd')
        return gmu, gsd, None, None, None, None, None


def _dims3(x_shape, fl, fs, Co, stride_len, stride):
    N, L, H, W, Ci = x_shape
    dims = (C.c_int * 10)(N, L, H, W, Ci, Co, fl, fs, int(stride_len), int(stride))
    out3 = (C.c_int * 3)()
    check(_L().ggan_conv3d_out_shape(dims, out3), 'ggan_conv3d_out_shape')
    return dims, tuple(out3)


class Im2Col3d(Function):
    """col[N*Lo*Ho*Wo, fl*fs*fs*Ci] of an NDHWC volume (SAME padding, strides (stride_len, stride, stride)); adjoint: Col2Im3d."""

    @staticmethod
    def forward(ctx, x, fl, fs, stride_len, stride):
        x = _c(x)
        dims, (Lo, Ho, Wo) = _dims3(x.shape, fl, fs, 1, stride_len, stride)
        col = torch.empty((x.shape[0] * Lo * Ho * Wo, fl * fs * fs * x.shape[4]), dtype=torch.float32, device=x.device)
        check(_L().ggan_im2col3d(dims, _p(x), _p(col), _stream()), 'ggan_im2col3d')
        ctx.args = (tuple(x.shape), fl, fs, stride_len, stride)
        return col

    @staticmethod
    def backward(ctx, g):
        return Col2Im3d.apply(g, *ctx.args), None, None, None, None


class Col2Im3d(Function):
    @staticmethod
    def forward(ctx, col, x_shape, fl, fs, stride_len, stride):
        col = _c(col)
        dims, _ = _dims3(x_shape, fl, fs, 1, stride_len, stride)
        gx = torch.empty(x_shape, dtype=torch.float32, device=col.device)
        check(_L().ggan_col2im3d(dims, _p(col), _p(gx), _stream()), 'ggan_col2im3d')
        ctx.args = (fl, fs, stride_len, stride)
        return gx

    @staticmethod
    def backward(ctx, g):
        return Im2Col3d.apply(g, *ctx.args), None, None, None, None, None


def _conv3d_patch(x, w, bias, stride_len, stride, act, alpha):
    """the layer as patch matrix @ filter (every derivative is again im2col / col2im / Gemm: closed under differentiation)"""
    fl, fs, _, Ci, Co = w.shape
    _, (Lo, Ho, Wo) = _dims3(x.shape, fl, fs, Co, stride_len, stride)
    col = Im2Col3d.apply(x, fl, fs, stride_len, stride)
    y = Gemm.apply(col, w.reshape(fl * fs * fs * Ci, Co), bias.reshape(-1) if bias is not None else None, False, False, act, alpha)
    return y.view(x.shape[0], Lo, Ho, Wo, Co)


def _igemm_ok(dims, kind):
    return bool(_L().ggan_conv3d_igemm_ok(dims, kind)) and not os.environ.get('GGAN_CONV3D_PATCH_MATRIX')


@_skip_undefined
class Conv3dImplicit(Function):
    """Conv3D as implicit GEMMs (ggan_conv3d_fwd / _wgrad / _dgrad: the patch matrix is never materialised).  First-order
    gradients run on the implicit kernels where ggan_conv3d_igemm_ok covers the geometry; a backward pass that is itself being
    differentiated (create_graph) is rebuilt from the patch-matrix operators, which are closed under differentiation."""

    @staticmethod
    def forward(ctx, x, w, bias, stride_len, stride, act, alpha, grad_rows=None):
        """grad_rows (optional): only volumes [0, grad_rows) of x need a gradient (the rest of the batch is data -- the critic on
        [fake; real] in a generator step); honoured when neither the filter nor the bias asks for one"""
        x, w = _c(x), _c(w)
        fl, fs, _, Ci, Co = w.shape
        ctx.grad_rows = int(grad_rows) if grad_rows else None
        dims, (Lo, Ho, Wo) = _dims3(x.shape, fl, fs, Co, stride_len, stride)
        y = torch.empty((x.shape[0], Lo, Ho, Wo, Co), dtype=torch.float32, device=x.device)
        ws = workspace(x.device)
        b = _c(bias.reshape(-1)) if bias is not None else None
        check(_L().ggan_conv3d_fwd(dims, _p(x), _p(w), _p(b), _p(y), act, alpha, _p(ws), ws.numel(), _stream()), 'ggan_conv3d_fwd')
        ctx.args = (stride_len, stride, act, alpha, None if bias is None else tuple(bias.shape))
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride_len, stride, act, alpha, bshape = ctx.args
        fl, fs, _, Ci, Co = w.shape
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], bshape is not None and ctx.needs_input_grad[2]
        N = x.shape[0]
        rows = ctx.grad_rows if (ctx.grad_rows and ctx.grad_rows < N and not need_w and not need_b
                                 and not torch.is_grad_enabled()) else N
        if rows < N:            # leading volumes only (contiguous prefixes of gy, y, gx)
            gy, y = _c(gy)[:rows], (y[:rows] if y is not None else None)
        gm = ActBwd.apply(gy, y, act, alpha) if act != ACT_NONE else _c(gy)
        gm2 = gm.reshape(-1, Co)
        if torch.is_grad_enabled():         # double backward: the differentiable composition
            gb = ColSum.apply(gm2).reshape(bshape) if need_b else None
            col = Im2Col3d.apply(x, fl, fs, stride_len, stride) if need_w else None
            gw = Gemm.apply(col, gm2, None, True, False, ACT_NONE, 0.0).view(w.shape) if need_w else None
            gx = (Col2Im3d.apply(Gemm.apply(gm2, w.reshape(-1, Co), None, False, True, ACT_NONE, 0.0), tuple(x.shape), fl, fs,
                                 stride_len, stride) if need_x else None)
            return gx, gw, gb, None, None, None, None, None
        dims, _ = _dims3((rows,) + tuple(x.shape[1:]), fl, fs, Co, stride_len, stride)
        gx = gw = gb = None
        if need_b:        # (row slabs + a fixed-order second stage from a few hundred rows on: the one-workgroup-per-column sum is serial)
            gb = torch.empty((Co,), dtype=torch.float32, device=x.device)
            ws = workspace(x.device)
            check(_L().ggan_colsum_tall(_p(gm2), _p(gb), gm2.shape[0], Co, _p(ws), ws.numel(), _stream()), 'ggan_colsum_tall')
            gb = gb.reshape(bshape)
        if need_w:
            if _igemm_ok(dims, 1):
                gw = torch.empty_like(w)
                ws = workspace(x.device)
                check(_L().ggan_conv3d_wgrad(dims, _p(x), _p(gm), _p(gw), _p(ws), ws.numel(), _stream()), 'ggan_conv3d_wgrad')
            else:
                gw = Gemm.apply(Im2Col3d.apply(x, fl, fs, stride_len, stride), gm2, None, True, False, ACT_NONE, 0.0).view(w.shape)
        if need_x:
            gx = torch.empty_like(x)
            if rows < N and DEBUG_POISON_CHECK:         # tests: the unwritten volumes must never be read downstream
                gx[rows:].fill_(float('nan'))
            if _igemm_ok(dims, 2):
                check(_L().ggan_conv3d_dgrad(dims, _p(gm), _p(w), _p(gx), _stream()), 'ggan_conv3d_dgrad')
            else:
                gcol = Gemm.apply(gm2, w.reshape(-1, Co), None, False, True, ACT_NONE, 0.0)
                check(_L().ggan_col2im3d(dims, _p(gcol), _p(gx), _stream()), 'ggan_col2im3d')
        return gx, gw, gb, None, None, None, None, None


def conv3d(x, w, bias, stride_len, stride, act=ACT_NONE, alpha=0.2, grad_rows=None):
    """act(conv3d(x[N,L,H,W,Ci], w[fl,fs,fs,Ci,Co], strides (sl,s,s), SAME) + bias)  (tflib/ops/conv3d.py:33-48) as an implicit GEMM
    on the MFMA units (Conv3dImplicit); geometries it does not cover take the patch matrix times the filter read in place as a
    [K, Co] matrix (_conv3d_patch)."""
    fl, fs, fs2, Ci, Co = w.shape
    assert x.dim() == 5 and fs == fs2 and x.shape[4] == Ci, (tuple(x.shape), tuple(w.shape))
    dims, (Lo, Ho, Wo) = _dims3(x.shape, fl, fs, Co, stride_len, stride)
    if _igemm_ok(dims, 0):
        return Conv3dImplicit.apply(x, w, bias, int(stride_len), int(stride), act, float(alpha), grad_rows)
    return _conv3d_patch(x, w, bias, stride_len, stride, act, alpha)


@_skip_undefined
class RowLerp(Function):
    """out[r,:] = x[r,:] + alpha[r]*(y[r,:]-x[r,:])  (the wali-gp interpolates)."""

    @staticmethod
    def forward(ctx, x, y, alpha):
        x, y, alpha = _c(x), _c(y), _c(alpha)
        rows, cols = x.shape
        out = torch.empty_like(x)
        check(_L().ggan_row_lerp(_p(x), _p(y), _p(alpha), _p(out), rows, cols, _stream()), 'ggan_row_lerp')
        ctx.save_for_backward(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        z = torch.zeros_like(g)
        gx = RowLerp.apply(g, z, alpha) if ctx.needs_input_grad[0] else None
        gy = RowLerp.apply(z, g, alpha) if ctx.needs_input_grad[1] else None
        return gx, gy, None


# ---------------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------------
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
            hrec['tail'] = dict(kind='mean', terms=hh['terms'] + tuple((1, 0.0, 1.0) for _ in xs[nh:]), logits=xs[0], g=hh['g'], loss=loss,
                                ext=[None] * nh + list(xs[nh:]))
            _PENDING_COSTS[loss.data_ptr()] = hrec['tail']
            ctx.unit_grads = outs
            return loss.reshape(())
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


# ---------------------------------------------------------------------------------------------------
# optimiser primitives (no autograd)
# ---------------------------------------------------------------------------------------------------
def adam_step_(theta, g, m, v, step, lr, beta1, beta2, eps=1e-8, grad_scale=1.0, counted=False):
    """counted=True: `step` was already advanced to this update's ordinal (pack_(..., bump=step))."""
    n = theta.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n and step.dtype == torch.int32
    L = _L()
    if counted:
        check(L.ggan_adam_step_counted(_p(theta), _p(g), _p(m), _p(v), n, _p(step), lr, beta1, beta2, eps, grad_scale,
                                       _stream()), 'ggan_adam_step_counted')
        return
    check(L.ggan_adam_step(_p(theta), _p(g), _p(m), _p(v), n, _p(step), lr, beta1, beta2, eps, grad_scale, _stream()),
          'ggan_adam_step')
    check(L.ggan_adam_advance(_p(step), _stream()), 'ggan_adam_advance')


def rmsprop_step_(theta, g, ms, lr, decay=0.9, eps=1e-10, grad_scale=1.0, clip=None):
    lo, hi = (float('-inf'), float('inf')) if clip is None else (float(clip[0]), float(clip[1]))
    check(_L().ggan_rmsprop_step(_p(theta), _p(g), _p(ms), theta.numel(), lr, decay, eps, grad_scale, lo, hi, _stream()),
          'ggan_rmsprop_step')


NOISE_NORMAL, NOISE_UNIFORM, NOISE_ONEHOT = 0, 1, 2


def noise_state(device, seed=None):
    """{seed, draw number, arrival counter} of ggan_noise_fill as an int64[3] device tensor (seed: torch.initial_seed())"""
    seed = torch.initial_seed() if seed is None else seed
    return torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=device)


def noise_fill_(state, specs):
    """One launch for all the noise of a step.  specs: list of (tensor, kind, a, b) -- NOISE_NORMAL: a + b*N(0,1); NOISE_UNIFORM:
    [a, b); NOISE_ONEHOT: rows of the 2-D tensor become one-hot with a uniformly drawn index.  In place; graph-capturable (the
    draw number advances on the device)."""
    assert state.dtype == torch.int64 and state.numel() == 3 and state.is_cuda
    n = len(specs)
    for t, _, _, _ in specs:
        assert t.is_contiguous() and t.dtype == torch.float32 and t.device == state.device
    dsts = (C.c_void_p * n)(*[t.data_ptr() for t, _, _, _ in specs])
    sizes = (C.c_size_t * n)(*[t.numel() for t, _, _, _ in specs])
    kinds = (C.c_int * n)(*[int(k) for _, k, _, _ in specs])
    a = (C.c_float * n)(*[float(x) for _, _, x, _ in specs])
    b = (C.c_float * n)(*[float(x) for _, _, _, x in specs])
    widths = (C.c_int * n)(*[int(t.shape[-1]) if k == NOISE_ONEHOT else 0 for t, k, _, _ in specs])
    check(_L().ggan_noise_fill(dsts, sizes, kinds, a, b, widths, n, _p(state), _stream()), 'ggan_noise_fill')


def pack_(tensors, offsets, flat, bump=None, adam=None):
    """flat[offsets[i] : offsets[i]+n_i] = tensors[i] (None -> zeros); tensors registered by `defer_wgrad_reduce` are
    summed over their split-K slabs on the way.  An entry may be a pair (t, t2): two gradient contributions of one parameter
    (either may be None), summed here.  bump: int32 device counter incremented once (the Adam step ordinal).
    adam = (theta, m, v, step, arrive, lr, beta1, beta2, eps, grad_scale): the Adam update rides in the same launch
    (ggan_pack_adam; at most PACK_MAX tensors; `step` takes the place of bump)."""
    L = _L()
    reg = _DEFER[0]
    assert adam is None or len(tensors) <= _lib.PACK_MAX
    for i0 in range(0, len(tensors), _lib.PACK_MAX):
        chunk = [(t if isinstance(t, tuple) else (t, None)) for t in tensors[i0:i0 + _lib.PACK_MAX]]
        chunk = [((b, None) if a is None else (a, b)) for a, b in chunk]        # a lone second contribution is the first
        n = len(chunk)
        sizes = (C.c_size_t * n)(*[int(s) for s in [o[1] for o in offsets[i0:i0 + n]]])
        offs = (C.c_size_t * n)(*[int(o[0]) for o in offsets[i0:i0 + n]])

        def table(col):
            ts = [c[col] for c in chunk]
            srcs = (C.c_void_p * n)(*[t.data_ptr() if t is not None else 0 for t in ts])
            info = [reg.pop(t.data_ptr(), None) if (reg and t is not None) else None for t in ts]
            parts = (C.c_int * n)(*[(e[0] if e else 1) for e in info])
            strides = (C.c_size_t * n)(*[(e[1] if e else 0) for e in info])
            return srcs, parts, strides
        s1, p1, st1 = table(0)
        if adam is not None:
            s2, p2, st2 = table(1) if any(c[1] is not None for c in chunk) else (None, None, None)
            theta, m, v, step, arrive, lr, b1, b2, eps, gscale = adam
            check(L.ggan_pack_adam(s1, sizes, offs, p1, st1, s2, p2, st2, n, _p(flat), _p(theta), _p(m), _p(v), _p(step), _p(arrive),
                                   lr, b1, b2, eps, gscale, _stream()), 'ggan_pack_adam')
        elif any(c[1] is not None for c in chunk):
            s2, p2, st2 = table(1)
            check(L.ggan_pack_parts2(s1, sizes, offs, p1, st1, s2, p2, st2, n, _p(flat), _p(bump if i0 == 0 else None), _stream()),
                  'ggan_pack_parts2')
        else:
            check(L.ggan_pack_parts(s1, sizes, offs, p1, st1, n, _p(flat), _p(bump if i0 == 0 else None), _stream()),
                  'ggan_pack_parts')
