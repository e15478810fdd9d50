"""Conv3D (tflib/ops/conv3d.py:6-51): NDHWC data [N, L, H, W, C], filter [filter_len, filter_size, filter_size, in, out], strides
(stride_len, stride, stride), SAME padding, biases of shape [1,1,1,1,out].  Same signature and initialisation as the reference; only
used by the '3dcnn' sequence critic of the state-space scripts.  Runs as an implicit GEMM on the MFMA units (csrc/conv3d.hip,
functional.conv3d; geometries the implicit kernels do not cover: im2col + GEMM)."""
import numpy as np

from ... import functional as F
from .. import param as _param
from .. import initial_values_needed as _draw


def Conv3D(name, filter_len, input_dim, output_dim, filter_size, inputs, he_init=True, stride=1, stride_len=1, biases=True,
           activation=None, alpha=0.2, grad_rows=None):
    """inputs: (N, L, H, W, C) -> (N, ceil(L/stride_len), ceil(H/stride), ceil(W/stride), output_dim).  Same signature as the
    reference; `activation` / `alpha` are an optional fused epilogue and `grad_rows` an optional backward pruning hint (extensions, as
    on Linear / Conv2D: only volumes [0, grad_rows) of `inputs` need a gradient)."""
    fan_in = input_dim * filter_size ** 2 * filter_len
    fan_out = output_dim * filter_size ** 2 / (stride ** 2) * filter_len / stride_len
    filters_stdev = np.sqrt(4. / (fan_in + fan_out)) if he_init else np.sqrt(2. / (fan_in + fan_out))
    shape = (filter_len, filter_size, filter_size, input_dim, output_dim)
    if _draw(name + '.Filters'):
        filter_values = np.random.uniform(low=-filters_stdev * np.sqrt(3), high=filters_stdev * np.sqrt(3), size=shape).astype('float32')
    else:
        filter_values = None
    filters = _param(name + '.Filters', filter_values)
    b = _param(name + '.Biases', np.zeros((1, 1, 1, 1, output_dim), dtype='float32')) if biases else None
    assert inputs.dim() == 5 and inputs.shape[-1] == input_dim, (name, tuple(inputs.shape), input_dim)
    return F.conv3d(inputs, filters, b, stride_len, stride, F.ACT_NONE if activation is None else activation, alpha, grad_rows)
