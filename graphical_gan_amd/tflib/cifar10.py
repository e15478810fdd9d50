"""CIFAR-10 minibatch generators with the interface of tflib/cifar10.py:21-48 (`load(batch_size, data_dir)` -> train / test
`get_epoch` callables yielding (images[B,3072] uint8, labels[B]) per minibatch), Python 3.  Reads the python-version pickles
(`data_batch_1..5`, `test_batch`) from `data_dir`; `data=` takes the arrays directly."""
import os
import pickle

import numpy as np


def unpickle(file):
    with open(file, 'rb') as fo:
        d = pickle.load(fo, encoding='latin1')
    return np.asarray(d['data']), np.asarray(d['labels'])


def get_reconstruction_data(n_samples, data_dir):
    """fixed reconstruction samples for comparison (tflib/cifar10.py:14-19)"""
    np.random.seed(1234)
    data, _ = unpickle(os.path.join(data_dir, 'test_batch'))
    np.random.shuffle(data)
    return data[:n_samples]


def cifar_generator(filenames, batch_size, data_dir, data=None):
    if data is not None:
        images, labels = np.array(data[0], copy=True), np.array(data[1], copy=True)
    else:
        parts = [unpickle(os.path.join(data_dir, f)) for f in filenames]
        images = np.concatenate([p[0] for p in parts], axis=0)
        labels = np.concatenate([p[1] for p in parts], axis=0)

    def get_epoch():
        state = np.random.get_state()
        np.random.shuffle(images)
        np.random.set_state(state)
        np.random.shuffle(labels)
        for i in range(len(images) // batch_size):
            yield (images[i * batch_size:(i + 1) * batch_size], labels[i * batch_size:(i + 1) * batch_size])

    return get_epoch


def load(batch_size, data_dir, data=None):
    """data: optional ((train_images, train_labels), (test_images, test_labels))"""
    if data is None and not os.path.isfile(os.path.join(data_dir, 'data_batch_1')):
        raise FileNotFoundError('CIFAR-10 python batches not found under %s (no network here)' % data_dir)
    return (cifar_generator(['data_batch_1', 'data_batch_2', 'data_batch_3', 'data_batch_4', 'data_batch_5'], batch_size,
                            data_dir, None if data is None else data[0]),
            cifar_generator(['test_batch'], batch_size, data_dir, None if data is None else data[1]))
