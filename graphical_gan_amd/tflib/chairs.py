"""Rotating-chairs sequence generators with the interface of tflib/chairs.py:36-44 (`load(seq_length, batch_size, size, data_dir,
num_dev)` -> train / dev `get_epoch` callables yielding float32 [B, seq_length, size*size*3] batches in (C,H,W) order, values
0..255), Python 3.  Reads `chairs_<size>.npy` ([objects, 31 views, H, W, 3]) from `data_dir`; `data=` takes the array directly."""
import os

import numpy as np


def rand_clip(x, seq_length):
    start = np.random.randint(x.shape[0] - seq_length + 1)
    return x[start:start + seq_length]


def chair_generator(batch_size, seq_length, data, size):
    def get_epoch():
        if seq_length == 1:
            data_all = data.reshape((-1, size * size * 3))
        elif seq_length == 31:
            data_all = data.reshape((-1, 31, size * size * 3))
        elif seq_length == 4:                                   # a random 4-view clip per object (tflib/chairs.py:21-26)
            data_all = np.asarray([rand_clip(d, seq_length) for d in data])[:, :seq_length, :]
        else:
            data_all = data[:, :seq_length, :]
        data_all = np.array(data_all, copy=True)
        np.random.shuffle(data_all)
        for i in range(data_all.shape[0] // batch_size):
            yield np.ascontiguousarray(data_all[i * batch_size:(i + 1) * batch_size], dtype=np.float32)

    return get_epoch


def load(seq_length, batch_size, size, data_dir, num_dev=200, data=None):
    if data is None:
        path = os.path.join(data_dir, 'chairs_%d.npy' % size)
        if not os.path.isfile(path):
            raise FileNotFoundError('%s not found (no network here)' % path)
        data = np.load(path)
    data = np.transpose(np.asarray(data), [0, 1, 4, 2, 3])       # [N, 31, H, W, 3] -> [N, 31, 3, H, W]
    data = data.reshape((-1, 31, size * size * 3))
    data = np.array(data, copy=True)
    np.random.shuffle(data)
    return (chair_generator(batch_size, seq_length, data[num_dev:], size),
            chair_generator(batch_size, seq_length, data[:num_dev], size))
