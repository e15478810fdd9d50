"""A TensorFlow-1.x graph-mode API surface on top of the oracle tape -- TEST INFRASTRUCTURE ONLY.

Purpose: run the REFERENCE'S OWN Python (the ten driver scripts and tflib/{__init__, ops/*, objs/*, utils/*} under
/root/reference, converted from Python 2 in memory) in the build container, where TensorFlow is absent, so that what the
restatement in oracle/nets.py / objs.py / step.py / ssgan.py claims about the reference's STRUCTURE is checked against the
reference itself: which `lib.param` names exist with which shapes, which layers a net is made of and in which order, how a
cost is composed, which variables an optimizer owns and with which hyper-parameters, which session.run happens when.
tests/golden/make_reference_trace.py drives it and commits what it records (tests/golden/reference_trace_*.npz,
param_manifest.json).

What this does NOT pin: the arithmetic of the TensorFlow primitives themselves.  `tf.nn.conv2d`, `conv2d_transpose`,
`fused_batch_norm`, `AdamOptimizer` ... are bound here to the same numpy restatements (oracle/ops.py, oracle/tape.py; SURVEY.md
Appendix A) the oracle uses, so a wrong reading of a TF primitive would be wrong on both sides.  The parity status at the TF
boundary therefore stays "unpinned" (oracle/__init__.py); the composition above the primitives is pinned.

Graph mode: every tf call builds a node (`Tensor`); `Session.run(fetches, feed_dict)` evaluates the fetched nodes on the tape
(one tape per run, so `tf.gradients` and `Optimizer.minimize` differentiate what that run computed, as TF does).  A node's
static shape comes from evaluating it once on dummy inputs when it is created.  Random ops draw from `Session.noise(node)`
(a caller-supplied source) and every draw of a run is logged, so the same draws can be fed to the restatement.
"""
import contextlib
import sys
import numpy as np

from . import tape as tp
from . import ops as O

DTYPE = np.float64            # arithmetic type of every float tensor (variables are converted on creation)

float32 = 'float32'
int32 = 'int32'
int64 = 'int64'


class TensorShape(object):
    def __init__(self, dims):
        self.dims = tuple(int(d) for d in dims)

    ndims = property(lambda self: len(self.dims))

    def as_list(self):
        return list(self.dims)

    def __getitem__(self, i):
        return TensorShape(self.dims[i]) if isinstance(i, slice) else self.dims[i]

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def __eq__(self, o):
        return tuple(self) == tuple(o)

    def __repr__(self):
        return 'TensorShape(%r)' % (self.dims,)


_GRAPH = None


class Graph(object):
    """Everything one script builds: nodes in creation order, variables, optimizers."""

    def __init__(self, seed=0):
        self.nodes, self.variables, self.optimizers, self.scopes = [], [], [], []
        self.dummy = np.random.RandomState(seed)


def reset(seed=0):
    global _GRAPH
    _GRAPH = Graph(seed)
    return _GRAPH


def graph():
    return _GRAPH


class Tensor(object):
    """A node: kind in {'op','var','placeholder','random','grad','item','train','assign','group'}."""

    def __init__(self, kind, fn=None, inputs=(), name=None, static=None, **attrs):
        self.kind, self.fn, self.inputs, self.name, self.attrs = kind, fn, list(inputs), name, attrs
        self.id = len(_GRAPH.nodes)
        self.scope = '/'.join(_GRAPH.scopes)
        _GRAPH.nodes.append(self)
        if static is None:
            with np.errstate(all='ignore'):
                static = fn(*[i.static for i in self.inputs])
        # a static is a VALUE (shape / dtype carrier): the tape behind it is dropped, or every node would pin the whole dummy graph
        self.static = tp.T(static.v) if isinstance(static, tp.T) else static

    # -- static shape --------------------------------------------------------------------------------
    def get_shape(self):
        return TensorShape(self.static.v.shape)

    shape = property(get_shape)
    dtype = property(lambda self: self.static.v.dtype)

    # -- operators -----------------------------------------------------------------------------------
    def __add__(self, o): return add(self, o)
    def __radd__(self, o): return add(o, self)
    def __sub__(self, o): return subtract(self, o)
    def __rsub__(self, o): return subtract(o, self)
    def __mul__(self, o): return multiply(self, o)
    def __rmul__(self, o): return multiply(o, self)
    def __truediv__(self, o): return divide(self, o)
    def __rtruediv__(self, o): return divide(o, self)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __neg__(self): return _op(tp.neg, self)
    def __pow__(self, p): return pow(self, p)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        return _op(lambda a: _t_index(a, idx), self)

    def __iter__(self):
        raise TypeError('a Tensor is not iterable')

    def __bool__(self):
        raise TypeError('a Tensor has no truth value in graph mode')

    __hash__ = object.__hash__

    def __repr__(self):
        return '<tf1_shim.Tensor #%d %s %s %r>' % (self.id, self.kind, self.name or '', tuple(self.static.v.shape) if self.static is not None else None)


class Variable(Tensor):
    """tf.Variable(initial_value, name=..., trainable=True)."""

    def __init__(self, initial_value, name=None, trainable=True, **kw):
        v = np.asarray(initial_value)
        if v.dtype.kind == 'f':
            v = v.astype(DTYPE)
        self.value = v.copy()
        self.initial_value = v.copy()
        self.trainable = trainable
        Tensor.__init__(self, 'var', name=name, static=tp.T(self.value))
        _GRAPH.variables.append(self)


# ---- tape helpers the tf surface needs beyond oracle/tape.py ----------------------------------------
def _t_index(a, idx):
    src = a.shape
    return tp.T(np.array(a.v[idx], copy=True), (a,), lambda g: (_t_unindex(g, idx, src),))


def _t_unindex(g, idx, shape):
    z = np.zeros(shape, g.v.dtype)
    z[idx] = g.v
    return tp.T(z, (g,), lambda h: (_t_index(h, idx),))


def _t_maximum(a, b):
    """tf.maximum; MaximumGrad sends the gradient to x where x >= y, else to y."""
    a = tp._t(a, b if isinstance(b, tp.T) else None); b = tp._t(b, a)
    m = np.broadcast_to(a.v >= b.v, np.broadcast(a.v, b.v).shape)
    ma, mb = tp.T(m.astype(a.v.dtype)), tp.T((~m).astype(a.v.dtype))
    sa, sb = a.shape, b.shape
    return tp.T(np.maximum(a.v, b.v), (a, b), lambda g: (tp.sum_to(tp.mul(g, ma), sa), tp.sum_to(tp.mul(g, mb), sb)))


def _t_tile(a, mult):
    """tf.tile as a broadcast over inserted axes."""
    shp = a.shape
    inter = []
    for s in shp:
        inter += [1, s]
    x = tp.reshape(a, tuple(inter))
    full = []
    for m, s in zip(mult, shp):
        full += [m, s]
    x = tp.broadcast_to(x, tuple(full))
    return tp.reshape(x, tuple(m * s for m, s in zip(mult, shp)))


def _t_const(v):
    v = np.asarray(v)
    if v.dtype.kind == 'f':
        v = v.astype(DTYPE)
    return tp.T(v)


def _lift(x):
    """python number / ndarray / Tensor -> Tensor"""
    if isinstance(x, Tensor):
        return x
    if isinstance(x, ShapeVec):
        x = list(x)
    if isinstance(x, (int, float)) and not isinstance(x, bool):        # python scalars are float constants; integer ARRAYS stay integer
        x = np.asarray(x, dtype=DTYPE)
    return Tensor('op', static=_t_const(x), fn=None, name='const')


def _op(fn, *inputs, **kw):
    return Tensor('op', fn, [_lift(i) for i in inputs], **kw)


def _axes(t, axis):
    if axis is None:
        return None
    nd = t.static.v.ndim
    if isinstance(axis, (list, tuple)):
        return tuple(a % nd for a in axis)
    return (axis % nd,)


# ---- tf.* : construction ----------------------------------------------------------------------------
def constant(value, dtype=None, shape=None, name=None):
    v = np.asarray(value)
    if dtype is not None:
        v = v.astype(DTYPE if str(dtype).startswith('float') else dtype)
    return _lift(v)


def placeholder(dtype, shape=None, name=None):
    kind = 'i' if 'int' in str(dtype) else 'f'
    dummy = _GRAPH.dummy.randint(0, 256, size=shape).astype(np.int32) if kind == 'i' else _GRAPH.dummy.uniform(0.1, 0.9, size=shape).astype(DTYPE)
    return Tensor('placeholder', static=tp.T(dummy), name=name, dkind=kind, pshape=tuple(shape))


class ShapeVec(list):
    """tf.shape(x): the static shape as a list of ints (every shape in these scripts is static)."""


def shape(x, name=None):
    return ShapeVec(_lift(x).static.v.shape)


def stack(values, axis=0, name=None):
    if all(not isinstance(v, Tensor) for v in values):
        return ShapeVec(int(v) for v in values)
    def f(*vs):
        vs = [v if v.v.dtype.kind == 'f' else tp.T(v.v.astype(DTYPE)) for v in vs]
        return tp.concat([tp.reshape(v, v.shape[:axis] + (1,) + v.shape[axis:]) for v in vs], axis)
    return _op(f, *values)


def pack(values, axis=0, name=None):
    return stack(values, axis)


def unstack(value, num=None, axis=0, name=None):
    if isinstance(value, list):
        return list(value)
    n = value.static.v.shape[axis]
    return [value[(slice(None),) * axis + (i,)] for i in range(n)]


unpack = unstack


def _shape_arg(s):
    if isinstance(s, Tensor):
        raise NotImplementedError('dynamic shapes')
    return tuple(int(d) for d in s)


def _random(kind, shp, **attrs):
    shp = _shape_arg(shp)
    dummy = _GRAPH.dummy.uniform(0.1, 0.9, size=shp).astype(DTYPE)
    return Tensor('random', static=tp.T(dummy), rkind=kind, rshape=shp, **attrs)


def random_normal(shape, mean=0.0, stddev=1.0, dtype=None, seed=None, name=None):
    z = _random('normal', shape)
    if stddev != 1.0:
        z = z * float(stddev)
    if mean != 0.0:
        z = z + float(mean)
    return z


def random_uniform(shape, minval=0, maxval=1, dtype=None, seed=None, name=None):
    u = _random('uniform', shape)
    if float(maxval) - float(minval) != 1.0:
        u = u * (float(maxval) - float(minval))
    if float(minval) != 0.0:
        u = u + float(minval)
    return u


class _Categorical(object):
    def __init__(self, probs=None, logits=None):
        self.probs = probs

    def sample(self, n):
        k = int(_lift(self.probs).static.v.shape[-1])
        dummy = _GRAPH.dummy.randint(0, k, size=(int(n),)).astype(np.int64)
        return Tensor('random', static=tp.T(dummy), rkind='categorical', rshape=(int(n),), classes=k)


class _NS(object):
    pass


distributions = _NS()
distributions.Categorical = _Categorical


# ---- tf.* : pointwise and shape ops ------------------------------------------------------------------
def add(a, b, name=None): return _op(tp.add, a, b)
def subtract(a, b, name=None): return _op(lambda x, y: tp.add(x, tp.neg(y)), a, b)
def multiply(a, b, name=None): return _op(tp.mul, a, b)
def scalar_mul(s, x): return multiply(s, x)
def divide(a, b, name=None): return _op(lambda x, y: tp.mul(x, tp.power(y, -1.0)), a, b)
def square(x, name=None): return _op(tp.square, x)
def sqrt(x, name=None): return _op(tp.sqrt, x)
def exp(x, name=None): return _op(tp.exp, x)
def log(x, name=None): return _op(tp.log, x)
def tanh(x, name=None): return _op(tp.tanh, x)
def abs(x, name=None): return _op(tp.absolute, x)              # noqa: A001
def maximum(a, b, name=None): return _op(_t_maximum, a, b)
def identity(x, name=None): return _op(lambda a: a, x)
def stop_gradient(x, name=None): return _op(lambda a: tp.T(a.v), x, blocks_gradient=True)


def pow(x, p, name=None):                                       # noqa: A001
    if isinstance(p, Tensor):
        raise NotImplementedError('tensor exponent')
    return _op(lambda a: tp.power(a, float(p)), x)


def cast(x, dtype, name=None):
    if 'float' in str(dtype):
        return _op(lambda a: tp.T(a.v.astype(DTYPE)) if a.v.dtype.kind != 'f' else a, x)
    return _op(lambda a: tp.T(a.v.astype(str(dtype))), x)


def ones_like(x, name=None): return _op(lambda a: tp.T(np.ones_like(a.v)), x)
def zeros_like(x, name=None): return _op(lambda a: tp.T(np.zeros_like(a.v)), x)
def ones(shape, dtype=None, name=None): return _lift(np.ones(_shape_arg(shape), DTYPE))
def eye(n, dtype=None, name=None): return _lift(np.eye(n, dtype=DTYPE))


def reshape(x, shape, name=None):
    shp = _shape_arg(shape)
    return _op(lambda a: tp.reshape(a, shp), x)


def transpose(x, perm=None, name=None):
    return _op(lambda a: tp.transpose(a, tuple(perm) if perm is not None else tuple(reversed(range(a.v.ndim)))), x)


def expand_dims(x, axis=None, name=None, dim=None):
    ax = axis if axis is not None else dim
    def f(a):
        k = ax % (a.v.ndim + 1)
        return tp.reshape(a, a.shape[:k] + (1,) + a.shape[k:])
    return _op(f, x)


def squeeze(x, axis=None, name=None, squeeze_dims=None):
    ax = axis if axis is not None else squeeze_dims
    return _op(lambda a: tp.reshape(a, np.squeeze(a.v, axis=tuple(ax) if ax is not None else None).shape), x)


def concat(values, axis, name=None):
    if not isinstance(values, (list, tuple)):                  # tf < 1.0 order: concat(axis, values)
        values, axis = axis, values
    return _op(lambda *vs: tp.concat(list(vs), axis % vs[0].v.ndim), *values)


def tile(x, multiples, name=None):
    m = tuple(int(v) for v in multiples)
    return _op(lambda a: _t_tile(a, m), x)


def one_hot(indices, depth, name=None, **kw):
    def f(i):
        out = np.zeros(i.v.shape + (depth,), DTYPE)
        np.put_along_axis(out, i.v.astype(np.int64)[..., None], 1.0, axis=-1)
        return tp.T(out)
    return _op(f, indices)


def argmax(x, axis=None, name=None, dimension=None):
    ax = axis if axis is not None else dimension
    return _op(lambda a: tp.T(np.argmax(a.v, axis=ax)), x)


def clip_by_value(x, lo, hi, name=None):
    return _op(lambda a: tp.T(np.clip(a.v, lo, hi)), x)             # (only the weight clipping of MODE wali: no gradient flows here)


def _reduce(which, x, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
    ax = axis if axis is not None else reduction_indices
    kd = keep_dims if keepdims is None else keepdims
    if isinstance(x, (list, tuple)) and any(isinstance(e, Tensor) for e in x):
        x = stack(list(x))
    x = _lift(x)
    axes = _axes(x, ax)
    return _op(lambda a: which(a, axes, kd), x)


def reduce_sum(x, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
    return _reduce(tp.reduce_sum, x, axis, keep_dims, name, reduction_indices, keepdims)


def reduce_mean(x, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
    return _reduce(tp.reduce_mean, x, axis, keep_dims, name, reduction_indices, keepdims)


def _t_reduce_max(a, axes, kd):
    axes = tuple(range(a.v.ndim)) if axes is None else axes
    mx = a.v.max(axis=axes, keepdims=True)
    hit = (a.v == mx).astype(a.v.dtype)
    hit = hit / hit.sum(axis=axes, keepdims=True)              # TF splits the gradient evenly between ties
    out = tp.reduce_sum(tp.mul(a, tp.T(hit)), axes, keepdims=kd)
    return out


def reduce_prod(x, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
    """(parameter-count logging of the state-space scripts: integer shapes in, no gradient)"""
    return _reduce(lambda a, axes, kd: tp.T(np.prod(a.v, axis=axes, keepdims=kd)), x, axis, keep_dims, name, reduction_indices, keepdims)


def reduce_max(x, axis=None, keep_dims=False, name=None, reduction_indices=None, keepdims=None):
    return _reduce(_t_reduce_max, x, axis, keep_dims, name, reduction_indices, keepdims)


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    def f(x, y):
        if transpose_a:
            x = tp.transpose(x, (1, 0))
        if transpose_b:
            y = tp.transpose(y, (1, 0))
        return tp.matmul(x, y)
    return _op(f, a, b)


def diag_part(x, name=None):
    def f(a):
        n = a.v.shape[0]
        return tp.reduce_sum(tp.mul(a, tp.T(np.eye(n, dtype=a.v.dtype))), (1,))
    return _op(f, x)


def trace(x, name=None):
    return reduce_sum(diag_part(x))


@contextlib.contextmanager
def name_scope(name, *a, **k):
    _GRAPH.scopes.append(str(name))
    try:
        yield name
    finally:
        _GRAPH.scopes.pop()


@contextlib.contextmanager
def control_dependencies(deps):
    yield


def cond(pred, true_fn, false_fn, name=None):
    raise NotImplementedError('tf.cond: no script passes is_training (SURVEY.md 0.1)')


# ---- tf.nn -------------------------------------------------------------------------------------------
nn = _NS()
layers = _NS()


def _relu(x, name=None): return _op(tp.relu, x)
def _sigmoid(x, name=None): return _op(tp.sigmoid, x)


def _softmax(x, dim=-1, name=None, axis=None):
    ax = axis if axis is not None else dim
    return _op(lambda a: tp.softmax(a, ax), x)


def _softsign(x, name=None):
    return _op(lambda a: tp.mul(a, tp.power(tp.add(tp.absolute(a), 1.0), -1.0)), x)


def _sigmoid_ce(_sentinel=None, labels=None, logits=None, name=None):
    """max(x, 0) - x * z + log(1 + exp(-|x|))   (TF's documented, numerically stable form)"""
    return _op(lambda x, z: tp.add(tp.add(tp.relu(x), tp.neg(tp.mul(x, z))), tp.log1p(tp.exp(tp.neg(tp.absolute(x))))), logits, labels)


def _softmax_ce(_sentinel=None, labels=None, logits=None, dim=-1, name=None):
    def f(x, z):
        m = tp.T(x.v.max(axis=-1, keepdims=True))
        xs = tp.add(x, tp.neg(m))
        lse = tp.log(tp.reduce_sum(tp.exp(xs), (x.v.ndim - 1,), keepdims=True))
        return tp.neg(tp.reduce_sum(tp.mul(z, tp.add(xs, tp.neg(lse))), (x.v.ndim - 1,)))
    return _op(f, logits, labels)


def _conv2d(input=None, filter=None, strides=None, padding='SAME', data_format='NHWC', name=None, **kw):   # noqa: A002
    if data_format != 'NCHW':
        raise NotImplementedError('conv2d NHWC')
    assert strides[0] == 1 and strides[1] == 1 and strides[2] == strides[3]
    s = int(strides[2])
    return _op(lambda x, w: tp.conv2d(x, w, s, padding), input, filter, prim='conv2d')


def _conv2d_transpose(value=None, filter=None, output_shape=None, strides=None, padding='SAME', data_format='NHWC', name=None):   # noqa: A002
    """NHWC in, NHWC out (tflib/ops/deconv2d.py:101-107); filter [k, k, out, in].  By definition the input-gradient of the
    forward conv that maps output_shape -> value's shape."""
    assert data_format == 'NHWC' and strides[0] == 1 and strides[3] == 1 and strides[1] == strides[2]
    s = int(strides[1])
    osh = _shape_arg(output_shape)

    def f(x, w):
        xn = tp.transpose(x, (0, 3, 1, 2))
        y = tp.conv2d_bwd_data(xn, w, (osh[1], osh[2]), s, padding)
        assert y.v.shape[1] == osh[3], (y.v.shape, osh)
        return tp.transpose(y, (0, 2, 3, 1))
    return _op(f, value, filter, prim='conv2d_transpose')


def _conv3d(input=None, filter=None, strides=None, padding='SAME', data_format='NDHWC', name=None):   # noqa: A002
    assert data_format == 'NDHWC' and padding == 'SAME' and strides[0] == 1 and strides[4] == 1 and strides[2] == strides[3]
    return _op(lambda x, w: tp.conv3d(x, w, int(strides[1]), int(strides[2])), input, filter, prim='conv3d')


def _bias_add(value, bias, data_format=None, name=None):
    def f(x, b):
        if data_format == 'NCHW':
            return tp.add(x, tp.reshape(b, (1, -1) + (1,) * (x.v.ndim - 2)))
        return tp.add(x, b)
    return _op(f, value, bias)


def _moments(x, axes, shift=None, name=None, keep_dims=False):
    x = _lift(x)
    ax = _axes(x, list(axes))
    mean = _op(lambda a: tp.reduce_mean(a, ax, keepdims=True), x)
    var = _op(lambda a, m: tp.reduce_mean(tp.square(tp.add(a, tp.neg(m))), ax, keepdims=True), x, mean)
    if not keep_dims:
        mean, var = squeeze(mean, list(ax)), squeeze(var, list(ax))
    return mean, var


def _batch_normalization(x, mean, variance, offset, scale, variance_epsilon, name=None):
    def f(a, m, v, o, s):
        inv = tp.power(tp.add(v, float(variance_epsilon)), -0.5)
        if s is not None:
            inv = tp.mul(inv, s)
        y = tp.mul(tp.add(a, tp.neg(m)), inv)
        return tp.add(y, o) if o is not None else y
    return _op(f, x, mean, variance, offset, scale)


def _fused_batch_norm(x, scale, offset, mean=None, variance=None, epsilon=0.001, data_format='NHWC', is_training=True, name=None):
    """-> (y, batch_mean, batch_var); normalisation uses the biased batch variance (the Bessel-corrected value TF returns as
    batch_var is not used by any script)."""
    assert data_format == 'NCHW' and is_training and mean is None
    x = _lift(x)
    y = _op(lambda a, s, o: tp.batchnorm_train(a, s, o, (0, 2, 3), float(epsilon)), x, scale, offset, prim='fused_batch_norm')
    bm = _op(lambda a: tp.reduce_mean(a, (0, 2, 3)), x)
    bv = _op(lambda a, m: tp.reduce_mean(tp.square(tp.add(a, tp.neg(tp.reshape(m, (1, -1, 1, 1))))), (0, 2, 3)), x, bm)
    return y, bm, bv


def _embedding_lookup(params, ids, name=None):
    return _op(lambda p, i: _t_index(p, (i.v.astype(np.int64),)), params, ids)


def _dropout_layer(inputs, rate=0.5, noise_shape=None, seed=None, training=False, name=None):
    """tf.layers.dropout: identity unless training=True (no script passes it)."""
    if training is not False:
        raise NotImplementedError('dropout with training=%r' % (training,))
    return identity(inputs)


nn.relu, nn.sigmoid, nn.softmax, nn.softsign = _relu, _sigmoid, _softmax, _softsign
nn.sigmoid_cross_entropy_with_logits, nn.softmax_cross_entropy_with_logits = _sigmoid_ce, _softmax_ce
nn.conv2d, nn.conv2d_transpose, nn.conv3d, nn.bias_add = _conv2d, _conv2d_transpose, _conv3d, _bias_add
nn.moments, nn.batch_normalization, nn.fused_batch_norm, nn.embedding_lookup = _moments, _batch_normalization, _fused_batch_norm, _embedding_lookup
layers.dropout = _dropout_layer
sigmoid = _sigmoid


# ---- gradients, optimizers, assignment -----------------------------------------------------------------
def gradients(ys, xs, grad_ys=None, name=None, **kw):
    """-> [d sum(ys) / dx for x in xs]; entries with no path are None (known when the node is created)."""
    ys = list(ys) if isinstance(ys, (list, tuple)) else [ys]
    xs = list(xs) if isinstance(xs, (list, tuple)) else [xs]
    assert grad_ys is None
    ny = len(ys)

    def f(*vals):
        total = None
        for y in vals[:ny]:
            s = tp.reduce_sum(y)
            total = s if total is None else tp.add(total, s)
        return tp.grad(total, list(vals[ny:]))
    node = Tensor('grad', f, ys + xs, static=tp.T(np.zeros(())), ny=ny)
    # which xs have a differentiable path to some y is a property of the graph: walk the producers of ys
    reach, stack = set(), list(ys)
    while stack:
        t = stack.pop()
        if t.id in reach:
            continue
        reach.add(t.id)
        if t.kind in ('op', 'item', 'grad') and not t.attrs.get('blocks_gradient') and t.static.v.dtype.kind == 'f':
            stack.extend(t.inputs)
    out = []
    for i, x in enumerate(xs):
        out.append(Tensor('item', None, [node], static=tp.T(np.zeros(x.static.v.shape, DTYPE)), index=i) if x.id in reach else None)
    return out


class _Optimizer(object):
    def __init__(self, **hp):
        self.hp = hp
        self.var_list = None
        self.state = {}
        self.t = 0
        self.cost = None
        _GRAPH.optimizers.append(self)

    def minimize(self, loss, var_list=None, **kw):
        assert var_list is not None, 'the scripts always pass var_list'
        self.var_list = list(var_list)
        self.cost = loss
        return Tensor('train', None, [loss] + self.var_list, static=tp.T(np.zeros(())), opt=self)


class AdamOptimizer(_Optimizer):
    """t += 1; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m, v moving averages; var -= lr_t*m/(sqrt(v)+eps)   """
    kind = 'adam'

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **kw):
        _Optimizer.__init__(self, lr=float(learning_rate), beta1=float(beta1), beta2=float(beta2), eps=float(epsilon))

    def apply(self, grads):
        hp = self.hp
        self.t += 1
        lr_t = hp['lr'] * np.sqrt(1.0 - hp['beta2'] ** self.t) / (1.0 - hp['beta1'] ** self.t)
        for var, g in zip(self.var_list, grads):
            if g is None:                                      # minimize drops (None, var) pairs
                continue
            m, v = self.state.get(var.id, (np.zeros_like(var.value), np.zeros_like(var.value)))
            m = hp['beta1'] * m + (1.0 - hp['beta1']) * g
            v = hp['beta2'] * v + (1.0 - hp['beta2']) * g * g
            self.state[var.id] = (m, v)
            var.value = var.value - lr_t * m / (np.sqrt(v) + hp['eps'])


class RMSPropOptimizer(_Optimizer):
    """TF defaults: decay .9, momentum 0, epsilon 1e-10, mean-square slot initialised to ONE."""
    kind = 'rmsprop'

    def __init__(self, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10, **kw):
        assert momentum == 0.0
        _Optimizer.__init__(self, lr=float(learning_rate), decay=float(decay), eps=float(epsilon))

    def apply(self, grads):
        hp = self.hp
        self.t += 1
        for var, g in zip(self.var_list, grads):
            if g is None:
                continue
            ms = self.state.get(var.id, np.ones_like(var.value))
            ms = hp['decay'] * ms + (1.0 - hp['decay']) * g * g
            self.state[var.id] = ms
            var.value = var.value - hp['lr'] * g / np.sqrt(ms + hp['eps'])


class _Saver(object):
    def __init__(self, *a, **k): pass
    def save(self, *a, **k): return None
    def restore(self, *a, **k): raise NotImplementedError


train = _NS()
train.AdamOptimizer, train.RMSPropOptimizer, train.Saver = AdamOptimizer, RMSPropOptimizer, _Saver


def assign(ref, value, name=None):
    assert isinstance(ref, Variable)
    return Tensor('assign', None, [_lift(value)], static=ref.static, ref=ref)


def group(*ops, **kw):
    return Tensor('group', None, list(ops), static=tp.T(np.zeros(())))


def global_variables_initializer():
    return Tensor('group', None, [], static=tp.T(np.zeros(())), name='init')


def trainable_variables():
    return [v for v in _GRAPH.variables if v.trainable]


# ---- Session ---------------------------------------------------------------------------------------------
class Run(object):
    """One session.run: the tape values of the nodes it touched, the random draws, the gradients of its train ops."""

    def __init__(self, index, fetches, feed):
        self.index, self.fetches, self.feed = index, fetches, feed
        self.memo, self.draws, self.deferred = {}, [], []
        self.train = []                                        # (optimizer, cost value, {var name: grad ndarray | None})
        self.results = None


class Session(object):
    """noise(node) -> ndarray is the source of every random draw (default: numpy RandomState(seed))."""
    current = None
    on_create = None                                            # driver hook: called with the new session (the graph is complete then)

    def __init__(self, *a, **k):
        self.runs = []
        self.noise = k.pop('noise', None)
        self._rs = np.random.RandomState(k.pop('seed', 1234))
        self.keep_values = ()                                  # nodes whose per-run value the driver wants logged
        Session.current = self
        if Session.on_create is not None:
            Session.on_create(self)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def _draw(self, node):
        if self.noise is not None:
            v = self.noise(node)
            if v is not None:
                return np.asarray(v)
        k, shp = node.attrs['rkind'], node.attrs['rshape']
        if k == 'normal':
            return self._rs.standard_normal(shp).astype(np.float32).astype(DTYPE)
        if k == 'uniform':
            return self._rs.random_sample(shp).astype(np.float32).astype(DTYPE)
        return self._rs.randint(0, node.attrs['classes'], size=shp).astype(np.int64)

    def _ev(self, t, run):
        memo = run.memo
        if t.id in memo:
            return memo[t.id]
        k = t.kind
        if k == 'var':
            v = tp.T(t.value)
        elif k == 'placeholder':
            if t not in run.feed:
                raise KeyError('placeholder %r is not fed' % (t,))
            a = np.asarray(run.feed[t])
            assert tuple(a.shape) == t.attrs['pshape'], (a.shape, t.attrs['pshape'])
            v = tp.T(a.astype(DTYPE) if t.attrs['dkind'] == 'f' else a)
        elif k == 'random':
            a = self._draw(t)
            if a.dtype.kind == 'f':
                a = a.astype(DTYPE)                            # (a float32 draw would drag products of two draws down to float32)
            run.draws.append((t.id, t.attrs['rkind'], a))
            v = tp.T(a)
        elif k == 'op':
            v = t.static if t.fn is None else t.fn(*[self._ev(i, run) for i in t.inputs])
        elif k == 'grad':
            v = t.fn(*[self._ev(i, run) for i in t.inputs])
        elif k == 'item':
            v = self._ev(t.inputs[0], run)[t.attrs['index']]
        elif k == 'train':
            opt = t.attrs['opt']
            cost = self._ev(t.inputs[0], run)
            vs = [self._ev(x, run) for x in opt.var_list]
            gs = tp.grad(cost, vs)
            grads = [None if g is None else g.v for g in gs]
            run.train.append((opt, float(cost.v), {x.name: g for x, g in zip(opt.var_list, grads)}))
            run.deferred.append(lambda: opt.apply(grads))
            v = None
        elif k == 'assign':
            val = self._ev(t.inputs[0], run)
            ref = t.attrs['ref']
            run.deferred.append(lambda: setattr(ref, 'value', np.array(val.v, copy=True)))
            v = val
        elif k == 'group':
            for i in t.inputs:
                self._ev(i, run)
            v = None
        else:
            raise ValueError(k)
        memo[t.id] = v
        return v

    def run(self, fetches, feed_dict=None):
        lim = sys.getrecursionlimit()
        sys.setrecursionlimit(max(lim, 20000))
        run = Run(len(self.runs), fetches, dict(feed_dict or {}))

        def go(f):
            if isinstance(f, (list, tuple)):
                return [go(x) for x in f]
            v = self._ev(f, run)
            if v is None:
                return None
            if isinstance(v, list):
                return [None if x is None else np.array(x.v) for x in v]
            a = np.array(v.v)
            return a if a.ndim else a[()]
        with np.errstate(over='ignore'):
            run.results = go(fetches)
        run.kept = {n.id: np.array(run.memo[n.id].v) for n in self.keep_values if n.id in run.memo and run.memo[n.id] is not None}
        for d in run.deferred:
            d()
        run.memo = None                                        # (the tape of a run can be large)
        self.runs.append(run)
        sys.setrecursionlimit(lim)
        return run.results
