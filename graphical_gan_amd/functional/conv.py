"""Conv2D / Deconv2D family: forward, data gradient (plain and masked), filter gradient, channel / column sums."""
import ctypes as C
import os
import os as _os
import torch
from torch.autograd import Function
from .._lib import ACT_NONE, ACT_LRELU, ACT_RELU, check  # noqa: F401
from ._core import (  # noqa: F401
    _L, _p, _stream, _c, FUSED_CONV_BWD, _DEFER, _DATA_ONLY, _is_param, _skip_undefined, _wgrad_parts, workspace, _TARGET,
    _bwd_target, target_workgroups, _PLAN, _carries_hint, _planned_for, _geom, _new_out)
from .pointwise import ActBwd  # noqa: F401


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
