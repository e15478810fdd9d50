"""Tiny reverse-mode autodiff over numpy, closed under differentiation.

TEST INFRASTRUCTURE (see oracle/__init__.py).  It plays the role tf.gradients plays in the
reference (tflib/objs/gan_inference.py:108-117 via AdamOptimizer.minimize, and explicitly at
gan_inference_cifar10.py:362 for the gradient penalty).  Every vjp is written in terms of
taped ops, so grad-of-grad (the wali-gp double backward) just works.
Validated by float64 central finite differences in tests/test_oracle_grad.py.
"""
import numpy as np
from . import ops as O


class T:
    """A taped tensor: value + how to push a cotangent to its parents."""
    __slots__ = ('v', 'parents', 'vjp')

    def __init__(self, v, parents=(), vjp=None):
        self.v = np.asarray(v)
        self.parents = parents
        self.vjp = vjp

    @property
    def shape(self):
        return self.v.shape

    def __add__(self, o): return add(self, o)
    __radd__ = __add__
    def __sub__(self, o): return add(self, neg(_t(o, self)))
    def __rsub__(self, o): return add(_t(o, self), neg(self))
    def __mul__(self, o): return mul(self, o)
    __rmul__ = __mul__
    def __neg__(self): return neg(self)
    def __truediv__(self, o):
        if isinstance(o, T):
            return mul(self, power(o, -1.0))
        return scale(self, 1.0 / o)


def _t(x, like=None):
    if isinstance(x, T):
        return x
    dt = like.v.dtype if like is not None else None
    return T(np.asarray(x, dtype=dt))


const = _t


# ---- broadcasting helpers -------------------------------------------------------------
def sum_to(a, shape):
    shape = tuple(shape)
    if a.shape == shape:
        return a
    v = a.v
    lead = v.ndim - len(shape)
    axes = tuple(range(lead)) + tuple(i + lead for i, s in enumerate(shape) if s == 1 and v.shape[i + lead] != 1)
    out = v.sum(axis=axes, keepdims=True).reshape(shape)
    src = a.shape
    return T(out, (a,), lambda g: (broadcast_to(g, src),))


def broadcast_to(a, shape):
    shape = tuple(shape)
    if a.shape == shape:
        return a
    src = a.shape
    return T(np.broadcast_to(a.v, shape).copy(), (a,), lambda g: (sum_to(g, src),))


def add(a, b):
    a = _t(a, b if isinstance(b, T) else None); b = _t(b, a)
    sa, sb = a.shape, b.shape
    return T(a.v + b.v, (a, b), lambda g: (sum_to(g, sa), sum_to(g, sb)))


def neg(a):
    return T(-a.v, (a,), lambda g: (neg(g),))


def scale(a, c):
    c = float(c)
    return T(a.v * a.v.dtype.type(c), (a,), lambda g: (scale(g, c),))


def mul(a, b):
    if not isinstance(b, T) and np.isscalar(b):
        return scale(a, b)
    a = _t(a, b); b = _t(b, a)
    sa, sb = a.shape, b.shape
    return T(a.v * b.v, (a, b), lambda g: (sum_to(mul(g, b), sa), sum_to(mul(g, a), sb)))


def power(a, p):
    p = float(p)
    return T(a.v ** a.v.dtype.type(p), (a,), lambda g: (mul(g, scale(power(a, p - 1.0), p)),))


def sqrt(a): return power(a, 0.5)
def square(a): return mul(a, a)


def exp(a):
    out = T(np.exp(a.v), (a,), None)
    out.vjp = lambda g: (mul(g, out),)
    return out


def log(a):
    return T(np.log(a.v), (a,), lambda g: (mul(g, power(a, -1.0)),))


def log1p(a):
    return T(np.log1p(a.v), (a,), lambda g: (mul(g, power(add(a, 1.0), -1.0)),))


def absolute(a):
    s = T(np.sign(a.v))
    return T(np.abs(a.v), (a,), lambda g: (mul(g, s),))


KINK_LOG = None      # tests: set to [] to collect the 2-D (Linear-layer) ReLU / LeakyReLU inputs of a forward pass
# Kink tables / forced branches of the image-shaped activations (tests/golden/make_golden_full.py, the state-space fixture; the image
# scripts' fixtures use the same scheme in oracle/torch_cpu.py): with KINKS a dict, every keyed ReLU / LeakyReLU on a 4-D tensor records the
# KINK_K units nearest zero (flat positions, float64 values, the layer's rms) under key + KINK_TAG[0]; with FORCE a dict key -> (positions,
# take-the-positive-branch flags) those units take the given branch whatever the sign of their input.
KINKS = None
FORCE = None
KINK_TAG = ['']
KINK_K = 48


def _kink(a, key, slope):
    """-> the 0/1 (or slope/1) derivative mask of the activation, with forced units, and records the kink table"""
    x = a.v
    pos = x > 0
    if key is not None and x.ndim == 4:
        k = key + KINK_TAG[0]
        if KINKS is not None:
            f = x.reshape(-1).astype(np.float64)
            n = min(KINK_K, f.size)
            idx = np.sort(np.argpartition(np.abs(f), n - 1)[:n])
            KINKS[k] = dict(idx=idx.astype(np.int64), val=f[idx], rms=float(np.sqrt(np.mean(f * f))), shape=tuple(x.shape))
        if FORCE and k in FORCE:
            idx, p = FORCE[k]
            pos = pos.copy().reshape(-1)
            pos[np.asarray(idx, dtype=np.int64)] = np.asarray(p, dtype=bool)
            pos = pos.reshape(x.shape)
    return np.where(pos, 1.0, slope).astype(x.dtype)


def relu(a, key=None):
    """tf.nn.relu; gradient g*(x>0).  key: see KINKS / FORCE (a forced unit's value is x * its branch's slope, |x| ~ 1e-7 rms there)."""
    if KINK_LOG is not None and a.v.ndim == 2:
        KINK_LOG.append(a.v)
    mv = _kink(a, key, 0.0)
    m = T(mv)
    forced = key is not None and FORCE and (key + KINK_TAG[0]) in FORCE
    return T(a.v * mv if forced else np.maximum(a.v, 0), (a,), lambda g: (mul(g, m),))


def leaky_relu(a, alpha=0.2, key=None):
    """tf.maximum(alpha*x, x) (gmgan_inference_cifar10.py:122-123).  Piecewise linear:
    slope 1 for x>0 else alpha; second derivative zero a.e."""
    if KINK_LOG is not None and a.v.ndim == 2:
        KINK_LOG.append(a.v)
    mv = _kink(a, key, alpha)
    m = T(mv)
    forced = key is not None and FORCE and (key + KINK_TAG[0]) in FORCE
    return T(a.v * mv if forced else O.leaky_relu(a.v, alpha), (a,), lambda g: (mul(g, m),))


def tanh(a):
    out = T(np.tanh(a.v), (a,), None)
    out.vjp = lambda g: (mul(g, add(1.0, neg(square(out)))),)
    return out


def sigmoid(a):
    out = T(O.sigmoid(a.v), (a,), None)
    out.vjp = lambda g: (mul(g, mul(out, add(1.0, neg(out)))),)
    return out


# ---- shape ops ------------------------------------------------------------------------
def reshape(a, shape):
    src = a.shape
    return T(a.v.reshape(shape), (a,), lambda g: (reshape(g, src),))


def transpose(a, perm):
    inv = np.argsort(perm)
    return T(a.v.transpose(perm), (a,), lambda g: (transpose(g, inv),))


def concat(xs, axis=1):
    sizes = [x.shape[axis] for x in xs]
    offs = np.cumsum([0] + sizes)

    def vjp(g):
        return tuple(slice_axis(g, axis, int(offs[i]), int(offs[i + 1])) for i in range(len(xs)))
    return T(np.concatenate([x.v for x in xs], axis=axis), tuple(xs), vjp)


def slice_axis(a, axis, lo, hi):
    src = a.shape

    def vjp(g):
        return (pad_axis(g, axis, lo, src[axis] - hi),)
    idx = [slice(None)] * a.v.ndim
    idx[axis] = slice(lo, hi)
    return T(a.v[tuple(idx)].copy(), (a,), vjp)


def pad_axis(a, axis, before, after):
    pads = [(0, 0)] * a.v.ndim
    pads[axis] = (before, after)
    n = a.shape[axis]
    return T(np.pad(a.v, pads), (a,), lambda g: (slice_axis(g, axis, before, before + n),))


def reduce_sum(a, axes=None, keepdims=False):
    src = a.shape
    if axes is None:
        axes = tuple(range(a.v.ndim))
    axes = tuple(axes)
    kshape = tuple(1 if i in axes else s for i, s in enumerate(src))

    def vjp(g):
        return (broadcast_to(reshape(g, kshape), src),)
    return T(a.v.sum(axis=axes, keepdims=keepdims), (a,), vjp)


def reduce_mean(a, axes=None, keepdims=False):
    if axes is None:
        axes = tuple(range(a.v.ndim))
    cnt = 1
    for i in axes:
        cnt *= a.shape[i]
    return scale(reduce_sum(a, axes, keepdims), 1.0 / cnt)


# ---- contractions ---------------------------------------------------------------------
def matmul(a, b):
    """tf.matmul(inputs, W) (tflib/ops/linear.py:133)."""
    return T(a.v @ b.v, (a, b),
             lambda g: (matmul(g, transpose(b, (1, 0))), matmul(transpose(a, (1, 0)), g)))


def conv2d(x, w, stride, padding='SAME'):
    hw = x.shape[2:]
    k = w.shape[0]
    return T(O.conv2d(x.v, w.v, stride, padding), (x, w),
             lambda g: (conv2d_bwd_data(g, w, hw, stride, padding),
                        conv2d_bwd_filter(x, g, k, stride, padding)))


def conv2d_bwd_data(gy, w, in_hw, stride, padding='SAME'):
    k = w.shape[0]
    return T(O.conv2d_bwd_data(gy.v, w.v, in_hw, stride, padding), (gy, w),
             lambda h: (conv2d(h, w, stride, padding),
                        conv2d_bwd_filter(h, gy, k, stride, padding)))


def conv2d_bwd_filter(x, gy, k, stride, padding='SAME'):
    hw = x.shape[2:]
    return T(O.conv2d_bwd_filter(x.v, gy.v, k, stride, padding), (x, gy),
             lambda h: (conv2d_bwd_data(gy, h, hw, stride, padding),
                        conv2d(x, h, stride, padding)))


def deconv2d(x, w, stride=2, padding='SAME'):
    """tf.nn.conv2d_transpose as called by tflib/ops/deconv2d.py:101-107."""
    oh, ow = O.deconv_out_hw(x.shape[2], x.shape[3], w.shape[0], stride, padding)
    return conv2d_bwd_data(x, w, (oh, ow), stride, padding)


# ---- composites the reference uses ----------------------------------------------------
def conv3d(x, w, stride_len, stride):
    """tf.nn.conv3d NDHWC SAME; closed under differentiation through its two gradient maps"""
    fl, fs = w.v.shape[0], w.v.shape[1]
    return T(O.conv3d(x.v, w.v, stride_len, stride), (x, w),
             lambda g: (conv3d_bwd_data(g, w, x.v.shape, stride_len, stride), conv3d_bwd_filter(x, g, fl, fs, stride_len, stride)))


def conv3d_bwd_data(gy, w, in_shape, stride_len, stride):
    fl, fs = w.v.shape[0], w.v.shape[1]
    return T(O.conv3d_bwd_data(gy.v, w.v, in_shape, stride_len, stride), (gy, w),
             lambda g: (conv3d(g, w, stride_len, stride), conv3d_bwd_filter(g, gy, fl, fs, stride_len, stride)))


def conv3d_bwd_filter(x, gy, fl, fs, stride_len, stride):
    return T(O.conv3d_bwd_filter(x.v, gy.v, fl, fs, stride_len, stride), (x, gy),
             lambda g: (conv3d_bwd_data(gy, g, x.v.shape, stride_len, stride), conv3d(x, g, stride_len, stride)))


def bce_with_logits(x, z):
    """tf.nn.sigmoid_cross_entropy_with_logits, z a python float 0/1."""
    return add(add(relu(x), scale(x, -float(z))), log1p(exp(neg(absolute(x)))))


def batchnorm_train(x, scale_p, offset_p, axes, eps=1e-5):
    """tflib/ops/batchnorm.py:29-30 / :74-87 -- batch statistics, biased variance."""
    axes = tuple(axes)
    shp = tuple(1 if i in axes else s for i, s in enumerate(x.shape))
    mean = reduce_mean(x, axes, keepdims=True)
    xc = add(x, neg(mean))
    var = reduce_mean(square(xc), axes, keepdims=True)
    inv = power(add(var, eps), -0.5)
    return add(mul(mul(xc, inv), reshape(scale_p, shp)), reshape(offset_p, shp))


def softmax(a, axis=-1):
    m = T(a.v.max(axis=axis, keepdims=True))
    e = exp(add(a, neg(m)))
    return mul(e, power(reduce_sum(e, (a.v.ndim - 1,) if axis == -1 else (axis,), keepdims=True), -1.0))


# ---- backward -------------------------------------------------------------------------
def grad(out, wrt, seed=None):
    """d out / d wrt (list of T).  Returns taped T's, so the result can be differentiated
    again.  Unreached inputs get None (TF: minimize drops (None, var) pairs)."""
    order, seen = [], set()

    def visit(n):
        stack = [(n, False)]
        while stack:
            node, done = stack.pop()
            if done:
                order.append(node)
                continue
            if id(node) in seen:
                continue
            seen.add(id(node))
            stack.append((node, True))
            for p in node.parents:
                if id(p) not in seen:
                    stack.append((p, False))
    visit(out)
    g = {id(out): seed if seed is not None else T(np.ones_like(out.v))}
    for node in reversed(order):
        gn = g.get(id(node))
        if gn is None or node.vjp is None:
            continue
        for p, gp in zip(node.parents, node.vjp(gn)):
            if gp is None:
                continue
            g[id(p)] = gp if id(p) not in g else add(g[id(p)], gp)
    return [g.get(id(w)) for w in wrt]
