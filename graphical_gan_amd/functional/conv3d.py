"""Conv3D of the state-space video critics (patch GEMM and implicit-GEMM paths)."""
import ctypes as C
import os
import torch
from torch.autograd import Function
from .._lib import ACT_NONE, check  # noqa: F401
from ._core import _L, _p, _stream, _c, _skip_undefined, workspace  # noqa: F401
from .pointwise import ActBwd  # noqa: F401
from .conv import DEBUG_POISON_CHECK, ColSum  # noqa: F401
from .linear import Gemm  # noqa: F401


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
