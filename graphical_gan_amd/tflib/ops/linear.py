"""Linear (tflib/ops/linear.py:24-148): y = x @ W + b with W [in,out]; MFMA GEMM through ggan_gemm."""
import numpy as np

from ... import functional as F
from .. import param as _param
from .. import initial_values_needed as _draw

_default_weightnorm = False


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


def disable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = False


_weights_stdev = None


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def _uniform(stdev, size):
    if _weights_stdev is not None:
        stdev = _weights_stdev
    return np.random.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')


def _initial(initialization, input_dim, output_dim):
    """The six schemes of tflib/ops/linear.py:48-104; values drawn from numpy's global RNG on every call
    (even when the parameter already exists), exactly as the reference consumes RNG state."""
    if initialization == 'lecun':
        return _uniform(np.sqrt(1. / input_dim), (input_dim, output_dim))
    if initialization == 'glorot' or initialization is None:
        return _uniform(np.sqrt(2. / (input_dim + output_dim)), (input_dim, output_dim))
    if initialization == 'he':
        return _uniform(np.sqrt(2. / input_dim), (input_dim, output_dim))
    if initialization == 'glorot_he':
        return _uniform(np.sqrt(4. / (input_dim + output_dim)), (input_dim, output_dim))
    if initialization == 'orthogonal':
        a = np.random.normal(0.0, 1.0, (input_dim, output_dim))
        u, _, v = np.linalg.svd(a, full_matrices=False)
        q = u if u.shape == (input_dim, output_dim) else v
        return q.reshape((input_dim, output_dim)).astype('float32')
    if initialization[0] == 'uniform':
        return np.random.uniform(low=-initialization[1], high=initialization[1],
                                 size=(input_dim, output_dim)).astype('float32')
    raise Exception('Invalid initialization!')


def Linear(name, input_dim, output_dim, inputs, biases=True, initialization=None, weightnorm=None, gain=1.,
           activation=None, alpha=0.2, out=None):
    """Same signature as the reference; `activation`/`alpha` are an optional fused epilogue and `out` an optional
    functional.RowSlot the result is written into (extensions)."""
    if _draw(name + '.W'):
        weight_values = _initial(initialization, input_dim, output_dim)
        weight_values *= gain
    else:
        weight_values = None          # session.run time: the graph exists, nothing is drawn (tflib.end_build_phase)
    weight = _param(name + '.W', weight_values)
    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        norm_values = np.sqrt(np.sum(np.square(weight_values), axis=0)) if weight_values is not None else None
        target_norms = _param(name + '.g', norm_values)
        norms = (weight * weight).sum(dim=0).sqrt()
        weight = weight * (target_norms / norms)
    b = _param(name + '.b', np.zeros((output_dim,), dtype='float32')) if biases else None
    act = F.ACT_NONE if activation is None else activation
    if isinstance(inputs, (tuple, list)):
        # extension: a pair (x1, x2) stands for tf.concat([x1, x2], 1) -- the GEMM reads both operands in place
        x1, x2 = inputs
        assert x1.dim() == 2 and x2.dim() == 2 and x1.shape[1] + x2.shape[1] == input_dim, (name, x1.shape, x2.shape, input_dim)
        if out is None and F.Gemm2.usable(x1, x2):
            return F.Gemm2.apply(x1, x2, weight, b, act, float(alpha))
        import torch
        inputs = torch.cat([x1, x2], 1)
    x = inputs if inputs.dim() == 2 else inputs.reshape(-1, input_dim)
    result = F.Gemm.apply(x, weight, b, False, False, act, float(alpha), out)
    if inputs.dim() != 2:
        result = result.reshape(tuple(inputs.shape[:-1]) + (output_dim,))
    return result


def LinearLReLULinear(name1, input_dim, hidden_dim, name2, inputs, alpha=0.2, differentiable=False):
    """Extension: Linear(name1, input_dim, hidden_dim) -> LeakyReLU(alpha) -> Linear(name2, hidden_dim, 1), the tail of every
    critic of the scripts (gan_inference_cifar10.py:246-254 'Discriminator.zx1' / 'Discriminator.Output',
    gmgan_inference_cifar10.py:288-291 'Discriminator.Hyper3' / 'Discriminator.HyperOutput'), as ONE op
    (functional.CriticHead).  Same registry keys, shapes and initial-value draws (in the same order) as the two Linear calls.
    inputs: a tensor or a pair standing for tf.concat([x1, x2], 1).  differentiable=True composes the plain layers (needed
    where the result is differentiated twice: the gradient-penalty pass).  Returns the logits [rows]."""
    pair = isinstance(inputs, (tuple, list))
    x1, x2 = inputs if pair else (inputs, None)
    usable = (not differentiable and x1.dim() == 2 and hidden_dim % 4 == 0
              and (x2 is None or (x2.dim() == 2 and F.Gemm2.usable(x1, x2))))
    if not usable:
        h = Linear(name1, input_dim, hidden_dim, inputs, activation=F.ACT_LRELU, alpha=alpha)
        return Linear(name2, hidden_dim, 1, h).reshape(-1)
    w1 = _param(name1 + '.W', _initial(None, input_dim, hidden_dim) if _draw(name1 + '.W') else None)
    b1 = _param(name1 + '.b', np.zeros((hidden_dim,), dtype='float32'))
    w2 = _param(name2 + '.W', _initial(None, hidden_dim, 1) if _draw(name2 + '.W') else None)
    b2 = _param(name2 + '.b', np.zeros((1,), dtype='float32'))
    assert x1.shape[1] + (x2.shape[1] if x2 is not None else 0) == input_dim, (name1, x1.shape, input_dim)
    return F.CriticHead.apply(x1, x2, w1, b1, w2, b2, float(alpha))


def linear_params(name, input_dim, output_dim):
    """(W, b) of Linear(name, input_dim, output_dim) without applying it: same registry keys, shapes and initial-value draws"""
    w = _param(name + '.W', _initial(None, input_dim, output_dim) if _draw(name + '.W') else None)
    b = _param(name + '.b', np.zeros((output_dim,), dtype='float32'))
    return w, b


def LinearBatchnormRows(name, input_dim, output_dim, inputs, bn_name, activation=None, alpha=0.2):
    """Extension: Linear(name, input_dim, output_dim, inputs) followed by Batchnorm(bn_name, [0], .) and an activation -- the head
    of every Generator of the image scripts (gan_inference_cifar10.py:134-138) -- as ONE op (functional.LinearBatchNormRows) where
    the shapes allow (whole minibatch in one workgroup: at most 128 rows, a multiple of 16); otherwise the two calls.  Same registry
    keys, shapes and initial-value draws, in the same order, as the two calls."""
    from . import batchnorm
    w = _param(name + '.W', _initial(None, input_dim, output_dim) if _draw(name + '.W') else None)
    b = _param(name + '.b', np.zeros((output_dim,), dtype='float32'))
    x = inputs if inputs.dim() == 2 else inputs.reshape(-1, input_dim)
    act = F.ACT_NONE if activation is None else activation
    if batchnorm._SYNC_GROUP in (None, False) and F.LinearBatchNormRows.usable(x, w):
        shape = [1, output_dim]
        offset = _param(bn_name + '.offset', np.zeros(shape, dtype='float32'))
        scale = _param(bn_name + '.scale', np.ones(shape, dtype='float32'))
        return F.LinearBatchNormRows.apply(x, w, b, scale, offset, 1e-5, act, float(alpha))
    out = F.Gemm.apply(x, w, b, False, False, F.ACT_NONE, 0.0, None)
    return batchnorm.Batchnorm(bn_name, [0], out, activation=activation, alpha=alpha)
