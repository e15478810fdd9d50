"""Conv2D (tflib/ops/conv2d.py:20-123): NCHW, HWIO filter [k,k,in,out], stride s, TF 'SAME' padding, bias_add.
Arithmetic: ggan_conv2d_fwd (fp32 MFMA implicit GEMM) and its gradients."""
import numpy as np

from ... import functional as F
from .. import param as _param
from .. import tap as _tap
from .. import initial_values_needed as _draw

_default_weightnorm = False


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


_weights_stdev = None


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def _uniform(stdev, size):
    return np.random.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')


def _mask(mask_type, filter_size, input_dim, output_dim):
    """PixelCNN masks of tflib/ops/conv2d.py:29-52 (unused by the GAN scripts, kept for API parity)."""
    mask_type, mask_n_channels = mask_type
    mask = np.ones((filter_size, filter_size, input_dim, output_dim), dtype='float32')
    center = filter_size // 2
    mask[center + 1:, :, :, :] = 0.
    mask[center, center + 1:, :, :] = 0.
    for i in range(mask_n_channels):
        for j in range(mask_n_channels):
            if (mask_type == 'a' and i >= j) or (mask_type == 'b' and i > j):
                mask[center, center, i::mask_n_channels, j::mask_n_channels] = 0.
    return mask


def Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, mask_type=None, stride=1, weightnorm=None,
           biases=True, gain=1., padding='SAME', activation=None, alpha=0.2, grad_rows=None):
    """inputs: (batch, channels, height, width) -> (batch, output_dim, ceil(h/stride), ceil(w/stride)).
    `activation`/`alpha` are an optional fused epilogue and `grad_rows` an optional backward pruning hint (see
    functional.ConvFwd) -- extensions; the defaults give the reference behaviour."""
    fan_in = input_dim * filter_size ** 2
    fan_out = output_dim * filter_size ** 2 / (stride ** 2)
    if mask_type is not None:
        fan_in /= 2.
        fan_out /= 2.
    filters_stdev = np.sqrt(4. / (fan_in + fan_out)) if he_init else np.sqrt(2. / (fan_in + fan_out))
    shape = (filter_size, filter_size, input_dim, output_dim)
    if _draw(name + '.Filters'):
        filter_values = _uniform(_weights_stdev if _weights_stdev is not None else filters_stdev, shape)
        filter_values *= gain
    else:
        filter_values = None          # session.run time: the graph exists, nothing is drawn (tflib.end_build_phase)
    filters = _param(name + '.Filters', filter_values)
    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        norm_values = np.sqrt(np.sum(np.square(filter_values), axis=(0, 1, 2))) if filter_values is not None else None
        target_norms = _param(name + '.g', norm_values)
        norms = (filters * filters).sum(dim=(0, 1, 2)).sqrt()
        filters = filters * (target_norms / norms)
    if mask_type is not None:
        import torch
        filters = filters * torch.as_tensor(_mask(mask_type, filter_size, input_dim, output_dim), device=filters.device)
    b = _param(name + '.Biases', np.zeros(output_dim, dtype='float32')) if biases else None
    N, C, H, W = inputs.shape
    assert C == input_dim, (name, inputs.shape, input_dim)
    geom = F.conv_geom(N, input_dim, H, W, output_dim, filter_size, stride, padding)
    act = F.ACT_NONE if activation is None else activation
    out = F.ConvFwd.apply(inputs, filters, b, geom, act, float(alpha), grad_rows)
    return _tap(name, out) if act in (F.ACT_LRELU, F.ACT_RELU) else out
