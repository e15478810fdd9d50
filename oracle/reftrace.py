"""Rules shared by the reference-trace generator (tests/golden/make_reference_trace.py, build container only) and the tests
that replay a trace on the restatement / on the HIP path -- TEST INFRASTRUCTURE ONLY.

A trace fixture holds no tensors, only what cannot be recomputed: names / shapes, the order of the reference's session.run
calls, the random nodes each run drew, and digests of what the reference computed.  Everything that goes IN is a function of
a key, defined here, so both sides regenerate identical weights, minibatches and noise:
  weights    det_weight(name, shape)                    (float32-representable; non-trivial biases and BatchNorm parameters)
  minibatch  det_batch(stream, index, spec)             (index = position in the loader's stream)
  noise      det_noise(run index, node id, kind, shape[, classes])
"""
import zlib
import numpy as np

SAMPLES = 8


def _rs(*key):
    return np.random.RandomState(zlib.crc32(repr(key).encode()) & 0x7fffffff)


def det_weight(name, shape, dtype=np.float64):
    """Deterministic stand-in for a trained-ish parameter `name` (the reference's initialisers draw from numpy's unseeded global
    RNG and start biases at 0 / scales at 1, which would leave those paths unexercised)."""
    rs = _rs('w', name)
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    if leaf in ('moving_mean',):
        v = np.zeros(shape)
    elif leaf in ('moving_variance',):
        v = np.ones(shape)
    elif leaf == 'scale':
        v = 1.0 + rs.uniform(-0.2, 0.2, size=shape)
    elif leaf in ('offset', 'Biases', 'b'):
        v = rs.uniform(-0.1, 0.1, size=shape)
    elif leaf == 'Mu':
        v = rs.standard_normal(shape)
    else:                                   # Filters / W: +-sqrt(3)*std, std = sqrt(4 / (fan_in + fan_out)) with the taps counted once
        if len(shape) >= 2:
            taps = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan = taps * (shape[-2] + shape[-1]) / (4.0 if len(shape) > 2 else 1.0)
        else:
            fan = shape[0]
        std = np.sqrt(4.0 / fan)
        v = rs.uniform(-std * np.sqrt(3), std * np.sqrt(3), size=shape)
    return v.astype(np.float32).astype(dtype)


def det_noise(run, node_id, kind, shape, classes=None):
    rs = _rs('n', int(run), int(node_id))
    shape = tuple(int(s) for s in shape)
    if kind == 'normal':
        return rs.standard_normal(shape).astype(np.float32)
    if kind == 'uniform':
        return rs.random_sample(shape).astype(np.float32)
    if kind == 'categorical':
        return rs.randint(0, int(classes), size=shape).astype(np.int64)
    raise ValueError(kind)


def det_batch(stream, index, spec):
    """spec: ('int', shape) -> int32 in [0, 256); ('unit', shape) -> float32 in [0, 1); ('label', n, classes) -> int labels."""
    rs = _rs('d', stream, int(index))
    if spec[0] == 'int':
        return rs.randint(0, 256, size=tuple(spec[1])).astype(np.int32)
    if spec[0] == 'unit':
        return rs.random_sample(tuple(spec[1])).astype(np.float32)
    if spec[0] == 'label':
        return rs.randint(0, int(spec[2]), size=(int(spec[1]),)).astype(np.int64)
    raise ValueError(spec)


FINAL_SAMPLES = 64      # entries kept of every tensor of the weights after the last run (round 4: the trajectory criterion needs entries, not a norm)


def sample_positions(name, size, count=SAMPLES):
    return _rs('s', name).randint(0, max(int(size), 1), size=count)


def digest(name, a, count=SAMPLES):
    """[L2 norm, max |a|, `count` entries at fixed positions] of a tensor (float64)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    if a.size == 0:
        return [0.0, 0.0] + [0.0] * count
    pos = sample_positions(name, a.size, count)
    return [float(np.sqrt((a * a).sum())), float(np.abs(a).max())] + [float(a[p]) for p in pos]
