"""CelebA 64x64 minibatch generators with the interface of tflib/celebA.py:11-35 (`load(batch_size, data_dir, num_dev)` ->
train / test `get_epoch` callables yielding images[B,12288] per minibatch), Python 3.  Reads `celebA_64x64.npy` from
`data_dir` (`data=` takes the array directly); the first `num_dev` rows after one global-RNG shuffle are the dev split."""
import os

import numpy as np


def celeba_generator(batch_size, images):
    def get_epoch():
        np.random.shuffle(images)
        for i in range(len(images) // batch_size):
            yield images[i * batch_size:(i + 1) * batch_size]

    return get_epoch


def load(batch_size, data_dir, num_dev=5000, data=None):
    if data is None:
        path = os.path.join(data_dir, 'celebA_64x64.npy')
        if not os.path.isfile(path):
            raise FileNotFoundError('%s not found (no network here)' % path)
        data = np.load(path)
    data = np.array(data, copy=True).reshape(len(data), -1)
    np.random.shuffle(data)
    return celeba_generator(batch_size, data[num_dev:]), celeba_generator(batch_size, data[:num_dev])
