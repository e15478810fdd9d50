"""activations (lib.ops nonlinearities): ActFwd / ActBwd and the functional wrappers."""
import torch
from torch.autograd import Function
from .._lib import ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID, check  # noqa: F401
from ._core import _L, _p, _stream, _c, _skip_undefined  # noqa: F401


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
