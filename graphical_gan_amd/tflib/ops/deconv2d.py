"""Deconv2D (tflib/ops/deconv2d.py:20-119): filter [k,k,out,in], stride-2 transposed conv, 'SAME' (out = 2*in).
Computed natively in NCHW by ggan_deconv2d_fwd -- the reference's two layout transposes (:91,:116) are
mathematically no-ops and are simply not performed."""
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


def Deconv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, weightnorm=None, biases=True, gain=1.,
             mask_type=None, stride=2, padding='SAME', activation=None, alpha=0.2, out=None):
    """inputs: (batch, input_dim, h, w) -> (batch, output_dim, stride*h, stride*w).  `activation`/`alpha`: optional fused
    epilogue; `out`: optional functional.RowSlot the result is written into (extensions)."""
    if mask_type is not None:
        raise Exception('Unsupported configuration')
    fan_in = input_dim * filter_size ** 2 / (stride ** 2)
    fan_out = output_dim * filter_size ** 2
    filters_stdev = np.sqrt(4. / (fan_in + fan_out)) if he_init else np.sqrt(2. / (fan_in + fan_out))
    shape = (filter_size, filter_size, output_dim, input_dim)
    if _draw(name + '.Filters'):
        filter_values = _uniform(_weights_stdev if _weights_stdev is not None else filters_stdev, shape)
        filter_values *= gain
    else:
        filter_values = None          # session.run time: the graph exists, nothing is drawn (tflib.end_build_phase)
    filters = _param(name + '.Filters', filter_values)
    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        norm_values = np.sqrt(np.sum(np.square(filter_values), axis=(0, 1, 3))) if filter_values is not None else None
        target_norms = _param(name + '.g', norm_values)
        norms = (filters * filters).sum(dim=(0, 1, 3)).sqrt()
        filters = filters * (target_norms / norms).unsqueeze(1)
    b = _param(name + '.Biases', np.zeros(output_dim, dtype='float32')) if biases else None
    N, C, h, w = inputs.shape
    assert C == input_dim, (name, inputs.shape, input_dim)
    if padding == 'SAME':
        H, W = stride * h, stride * w
    else:  # 'VALID' (deconv2d.py:98-99)
        H, W = stride * (h - 1) + filter_size, stride * (w - 1) + filter_size
    # geometry of the forward conv [N,output_dim,H,W] -> [N,input_dim,h,w] whose input-gradient this is
    geom = F.conv_geom(N, output_dim, H, W, input_dim, filter_size, stride, padding)
    assert geom[5] == h and geom[6] == w, (geom, h, w)
    act = F.ACT_NONE if activation is None else activation
    y = F.ConvDgrad.apply(inputs, filters, b, geom, act, float(alpha), out)
    return _tap(name, y) if act in (F.ACT_LRELU, F.ACT_RELU) else y
