"""MNIST minibatch generators with the interface of tflib/mnist.py:8-64 (`load(batch_size, test_batch_size, n_labelled)` ->
three `get_epoch` callables yielding (images[B,784] float32, targets[B] int) per minibatch), Python 3.

There is no download step (the GPU boxes have no network): `load` reads an mnist.pkl.gz that is already on disk
(`path=` or $GGAN_MNIST, default /tmp/mnist.pkl.gz as in the reference) or takes the three (images, targets) splits directly
through `data=`.  Shuffling follows the reference: images and targets are shuffled with the same numpy global-RNG state at
generator construction and again at the start of every epoch; a trailing partial minibatch is dropped by the reshape."""
import gzip
import os
import pickle

import numpy


def mnist_generator(data, batch_size, n_labelled, limit=None):
    images, targets = data
    images = numpy.array(images, dtype='float32', copy=True)
    targets = numpy.array(targets, copy=True)

    def shuffle_together(*arrays):
        state = numpy.random.get_state()
        for a in arrays:
            numpy.random.set_state(state)
            numpy.random.shuffle(a)

    shuffle_together(images, targets)
    if limit is not None:
        print('WARNING ONLY FIRST {} MNIST DIGITS'.format(limit))
        images = images[:limit]
        targets = targets.astype('int32')[:limit]
    labelled = None
    if n_labelled is not None:
        labelled = numpy.zeros(len(images), dtype='int32')
        labelled[:n_labelled] = 1

    def get_epoch():
        if labelled is not None:
            shuffle_together(images, targets, labelled)
        else:
            shuffle_together(images, targets)
        n = len(images) // batch_size
        image_batches = images[:n * batch_size].reshape(n, batch_size, 784)
        target_batches = targets[:n * batch_size].reshape(n, batch_size)
        for i in range(n):
            if labelled is not None:
                yield (numpy.copy(image_batches[i]), numpy.copy(target_batches[i]), numpy.copy(labelled))
            else:
                yield (numpy.copy(image_batches[i]), numpy.copy(target_batches[i]))

    return get_epoch


def read_pickle(path=None):
    """(train, dev, test), each (images[N,784] float32 in [0,1], targets[N]) -- the layout of the LISA-lab mnist.pkl.gz"""
    path = path or os.environ.get('GGAN_MNIST', '/tmp/mnist.pkl.gz')
    if not os.path.isfile(path):
        raise FileNotFoundError("MNIST pickle not found at %s (no network here: place mnist.pkl.gz there, set $GGAN_MNIST, "
                                "or pass data=(train, dev, test))" % path)
    with gzip.open(path, 'rb') as f:
        return pickle.load(f, encoding='latin1')


def load(batch_size, test_batch_size, n_labelled=None, path=None, data=None):
    train_data, dev_data, test_data = data if data is not None else read_pickle(path)
    return (mnist_generator(train_data, batch_size, n_labelled),
            mnist_generator(dev_data, test_batch_size, n_labelled),
            mnist_generator(test_data, test_batch_size, n_labelled))
