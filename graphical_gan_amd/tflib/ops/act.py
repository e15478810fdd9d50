"""Pointwise glue the reference scripts write inline with raw tf ops (LeakyReLU = tf.maximum(alpha*x, x) at
gmgan_inference_cifar10.py:122-123, tf.nn.relu, tf.tanh, tf.nn.sigmoid, the int->float input scaling at :342)."""
from ... import functional as F

LRELU, RELU, TANH, SIGMOID = F.ACT_LRELU, F.ACT_RELU, F.ACT_TANH, F.ACT_SIGMOID


def LeakyReLU(x, alpha=0.2):
    return F.leaky_relu(x, alpha)


def relu(x):
    return F.relu(x)


def tanh(x):
    return F.tanh(x)


def sigmoid(x):
    return F.sigmoid(x)


def dropout(x, rate=0.0, training=False):
    """tf.layers.dropout as the reference calls it (no training=True anywhere) is the identity."""
    if training:
        raise NotImplementedError('the reference never enables dropout; parity target is identity')
    return x


def cast_scale(x_int, div=255., mul=2., noise=None, out=None, ring=None, defer=False):
    """2*((float(x)/255.)-.5) (+ dequantisation noise for the 64x64 scripts).  out: optional functional.RowSlot.
    ring: read the minibatch from a device-resident ring instead of x_int (functional.CastScaleI32).
    defer (with a ring): returns a functional.PendingCast -- the first conv layer that consumes it scales the minibatch while it stages its
    input (no launch of its own); `.out` is the float tensor, valid once that layer (or .materialize()) has run."""
    if defer and ring is not None:
        return F.PendingCast(x_int, noise, div, mul, out, ring)
    return F.CastScaleI32.apply(x_int, noise, float(div), float(mul), out, ring)
