"""SVHN minibatch generators with the interface of tflib/svhn.py:19-53 (`load(batch_size, data_dir)` -> train / test `get_epoch`
callables yielding (images[B,3072] uint8 in C,H,W order, labels[B] with the digit 0 stored as 0, not 10)), Python 3.  Reads the
cropped-digits `train_32x32.mat` / `test_32x32.mat` from `data_dir` (no download: there is no network); `data=` takes the raw
(X[32,32,3,N], y[N]) pairs directly."""
import os

import numpy as np


def _prepare(X, y):
    y = np.array(y, copy=True).reshape(-1)
    y[y == 10] = 0                                               # tflib/svhn.py:38,42
    X = np.transpose(np.asarray(X), [3, 2, 0, 1])                # [H,W,C,N] -> [N,C,H,W]  (:43-44)
    return np.ascontiguousarray(X).reshape(-1, 32 * 32 * 3), y


def svhn_generator(data, batch_size):
    images, labels = data

    def get_epoch():
        state = np.random.get_state()
        np.random.shuffle(images)
        np.random.set_state(state)
        np.random.shuffle(labels)
        for i in range(len(images) // batch_size):
            yield (images[i * batch_size:(i + 1) * batch_size], labels[i * batch_size:(i + 1) * batch_size])

    return get_epoch


def load(batch_size, data_dir, data=None):
    """data: optional ((X_train, y_train), (X_test, y_test)) in the .mat layout"""
    if data is None:
        paths = [os.path.join(data_dir, f) for f in ('train_32x32.mat', 'test_32x32.mat')]
        if not all(os.path.isfile(p) for p in paths):
            raise FileNotFoundError('SVHN .mat files not found under %s (no network here)' % data_dir)
        from scipy.io import loadmat
        mats = [loadmat(p) for p in paths]
        data = [(m['X'], m['y']) for m in mats]
    return svhn_generator(_prepare(*data[0]), batch_size), svhn_generator(_prepare(*data[1]), batch_size)
