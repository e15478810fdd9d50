"""Moving-MNIST sequence synthesiser with the interface of tflib/simple_moving_mnist.py:9-112 (`load_video(seq_length,
batch_size, cla)` -> train / test `get_epoch` callables yielding (video[B,LEN,4096] float32, labels[B])), Python 3.

One digit per 64x64 canvas bounces off the walls (step 0.1 of the free canvas per frame).  The reference renders an epoch with
Python loops over every digit and frame (:62-89) -- at the speed of the accelerated step that host loop would be the
bottleneck -- so rendering here is vectorised: one fancy-index assignment per frame index.  The numpy global RNG is consumed
in exactly the reference's order (shuffle of images / labels, then y, x, theta), so a seeded run produces the same videos.
`data=` takes the MNIST splits directly (no download here, see tflib/mnist.py)."""
import numpy as np

from . import mnist as _mnist


def GetRandomTrajectory(step_length, seq_length, batch_size, image_size, digit_size):
    canvas_size = image_size - digit_size
    y = np.random.rand(batch_size)                       # initial position uniform inside the box
    x = np.random.rand(batch_size)
    theta = np.random.rand(batch_size) * 2 * np.pi       # random direction, unit speed
    v_y, v_x = np.sin(theta), np.cos(theta)
    start_y = np.zeros((seq_length, batch_size))
    start_x = np.zeros((seq_length, batch_size))
    for i in range(seq_length):
        y = y + v_y * step_length
        x = x + v_x * step_length
        # bounce off the edges (the reference's four per-digit ifs, vectorised; the two tests of a coordinate exclude each other)
        lo, hi = x <= 0, x >= 1.0
        x = np.where(lo, 0.0, np.where(hi, 1.0, x))
        v_x = np.where(lo | hi, -v_x, v_x)
        lo, hi = y <= 0, y >= 1.0
        y = np.where(lo, 0.0, np.where(hi, 1.0, y))
        v_y = np.where(lo | hi, -v_y, v_y)
        start_y[i, :] = y
        start_x[i, :] = x
    return (canvas_size * start_y).astype(np.int32), (canvas_size * start_x).astype(np.int32)


def Overlap(a, b):
    return np.maximum(a, b)


def render(images, start_y, start_x, image_size=64, digit_size=28):
    """images[n,28,28] placed at (start_y[i,j], start_x[i,j]) in frame i of video j -> [n, LEN, 64, 64] float32"""
    n, L = images.shape[0], start_y.shape[0]
    data = np.zeros((n, L, image_size, image_size), dtype=np.float32)
    jj = np.arange(n)[:, None, None]
    dy = np.arange(digit_size)[None, :, None]
    dx = np.arange(digit_size)[None, None, :]
    for i in range(L):
        rows = start_y[i][:, None, None] + dy
        cols = start_x[i][:, None, None] + dx
        data[jj, i, rows, cols] = Overlap(data[jj, i, rows, cols], images)
    return data


def moving_mnist_generator_video(data_all, seq_length, batch_size):
    images, labels = data_all
    images = np.array(images, dtype=np.float32, copy=True).reshape([-1, 28, 28])
    labels = np.array(labels, copy=True)
    image_size, step_length, digit_size = 64, 0.1, 28

    def get_epoch():
        state = np.random.get_state()
        np.random.shuffle(images)
        np.random.set_state(state)
        np.random.shuffle(labels)
        start_y, start_x = GetRandomTrajectory(step_length=step_length, seq_length=seq_length, batch_size=images.shape[0],
                                               image_size=image_size, digit_size=digit_size)
        data = render(images, start_y, start_x, image_size, digit_size).reshape(images.shape[0], seq_length,
                                                                                image_size * image_size)
        for ind in range(data.shape[0] // batch_size):
            yield data[ind * batch_size:(ind + 1) * batch_size], labels[ind * batch_size:(ind + 1) * batch_size]

    return get_epoch


def load_video(seq_length, batch_size, cla=None, path=None, data=None):
    train_data, dev_data, test_data = data if data is not None else _mnist.read_pickle(path)
    train_all_x = np.concatenate([train_data[0], dev_data[0]], axis=0)
    train_all_y = np.concatenate([train_data[1], dev_data[1]], axis=0)
    if cla is not None:
        train_all_x, train_all_y = train_all_x[train_all_y == cla], train_all_y[train_all_y == cla]
        test_x, test_y = test_data
        test_data = (test_x[test_y == cla], test_y[test_y == cla])
    return (moving_mnist_generator_video((train_all_x, train_all_y), seq_length, batch_size),
            moving_mnist_generator_video(test_data, seq_length, batch_size))
