"""Shared loader of the committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(HERE, name + '.npz'))


def traj_feeds(z):
    feeds, i = [], 0
    while any(k.startswith('feed%d/' % i) for k in z.files):
        f = {}
        for k in z.files:
            if k.startswith('feed%d/' % i):
                v = z[k]
                kk = k.split('/', 1)[1]
                f[kk] = v.astype(np.int32) if kk == 'real_x_int' else v
        feeds.append(f)
        i += 1
    return feeds


def digest(v):
    f = np.asarray(v, dtype=np.float64).reshape(-1)
    return np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]])
