"""Moving-MNIST synthesiser restated literally (per-digit / per-frame loops) from tflib/simple_moving_mnist.py:9-89.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The product's vectorised renderer (graphical_gan_amd/tflib/
simple_moving_mnist.py) is checked against this with the same numpy global-RNG seed: identical videos, bit for bit."""
import numpy as np


def get_random_trajectory(step_length, seq_length, batch_size, image_size, digit_size):      # :9-48
    canvas_size = image_size - digit_size
    y = np.random.rand(batch_size)
    x = np.random.rand(batch_size)
    theta = np.random.rand(batch_size) * 2 * np.pi
    v_y = np.sin(theta)
    v_x = np.cos(theta)
    start_y = np.zeros((seq_length, batch_size))
    start_x = np.zeros((seq_length, batch_size))
    for i in range(seq_length):
        y += v_y * step_length
        x += v_x * step_length
        for j in range(batch_size):
            if x[j] <= 0:
                x[j] = 0
                v_x[j] = -v_x[j]
            if x[j] >= 1.0:
                x[j] = 1.0
                v_x[j] = -v_x[j]
            if y[j] <= 0:
                y[j] = 0
                v_y[j] = -v_y[j]
            if y[j] >= 1.0:
                y[j] = 1.0
                v_y[j] = -v_y[j]
            start_y[i, :] = y
            start_x[i, :] = x
    return (canvas_size * start_y).astype(np.int32), (canvas_size * start_x).astype(np.int32)


def epoch(images, labels, seq_length, batch_size):                                          # :54-89, one epoch, num_digits = 1
    images = np.array(images, dtype=np.float32, copy=True).reshape([-1, 28, 28])
    labels = np.array(labels, copy=True)
    state = np.random.get_state()
    np.random.shuffle(images)
    np.random.set_state(state)
    np.random.shuffle(labels)
    start_y, start_x = get_random_trajectory(0.1, seq_length, images.shape[0], 64, 28)
    data = np.zeros((images.shape[0], seq_length, 64, 64), dtype=np.float32)
    for j in range(images.shape[0]):
        for i in range(seq_length):
            top, left = start_y[i, j], start_x[i, j]
            data[j, i, top:top + 28, left:left + 28] = np.maximum(data[j, i, top:top + 28, left:left + 28], images[j])
    data = data.reshape(images.shape[0], seq_length, 64 * 64)
    return [(data[k * batch_size:(k + 1) * batch_size], labels[k * batch_size:(k + 1) * batch_size])
            for k in range(data.shape[0] // batch_size)]
