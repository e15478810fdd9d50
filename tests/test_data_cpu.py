"""Data loaders of the hot path's callers (tflib/mnist.py, cifar10.py, celebA.py, simple_moving_mnist.py counterparts): interface,
epoch semantics, and the vectorised moving-MNIST renderer against its literal restatement (same seed -> identical videos)."""
import numpy as np


def _fake_mnist(n=96, seed=0):
    rng = np.random.default_rng(seed)
    mk = lambda m: (rng.random((m, 784), dtype=np.float32), rng.integers(0, 10, size=m))
    return mk(n), mk(n // 2), mk(n // 2)


def test_mnist_loader_pairs_images_with_targets():
    from graphical_gan_amd.tflib import mnist
    data = _fake_mnist()
    tag = {tuple(np.round(x[:4], 6)): int(t) for x, t in zip(*data[0])}
    np.random.seed(3)
    train, dev, test = mnist.load(16, 8, data=data)
    seen = 0
    for epoch in range(2):
        for x, t in train():
            assert x.shape == (16, 784) and x.dtype == np.float32 and t.shape == (16,)
            for xi, ti in zip(x, t):
                assert tag[tuple(np.round(xi[:4], 6))] == int(ti)       # the pair survived both shuffles
            seen += 1
    assert seen == 2 * (96 // 16)
    assert sum(1 for _ in dev()) == 48 // 8


def test_cifar_and_celeba_loaders():
    from graphical_gan_amd.tflib import cifar10, celebA
    rng = np.random.default_rng(1)
    imgs = rng.integers(0, 256, size=(50, 3072)).astype(np.uint8)
    labs = np.arange(50)
    imgs[:, 0] = labs                                                      # marker to check pairing
    tr, te = cifar10.load(8, '/nonexistent', data=((imgs, labs), (imgs[:16], labs[:16])))
    n = 0
    for x, y in tr():
        assert x.shape == (8, 3072) and np.array_equal(x[:, 0], y.astype(np.uint8))
        n += 1
    assert n == 50 // 8 and sum(1 for _ in te()) == 2
    faces = rng.integers(0, 256, size=(40, 64, 64, 3)).astype(np.uint8)
    tr, te = celebA.load(4, '/nonexistent', num_dev=8, data=faces)
    assert sum(1 for _ in tr()) == 8 and sum(1 for _ in te()) == 2
    assert next(iter(tr())).shape == (4, 12288)
    import pytest
    with pytest.raises(FileNotFoundError):
        cifar10.load(8, '/nonexistent')


def test_moving_mnist_matches_literal_restatement():
    from graphical_gan_amd.tflib import simple_moving_mnist as M
    from oracle import moving_mnist as O
    train, dev, test = _fake_mnist(40, seed=5)
    np.random.seed(11)
    ref = O.epoch(test[0], test[1], seq_length=6, batch_size=5)
    np.random.seed(11)
    gen = M.moving_mnist_generator_video(test, 6, 5)
    got = list(gen())
    assert len(got) == len(ref) == 4
    for (v, y), (rv, ry) in zip(got, ref):
        assert v.shape == (5, 6, 4096) and v.dtype == np.float32
        assert np.array_equal(v, rv) and np.array_equal(y, ry)
    # the digit stays inside the canvas and keeps its mass from frame to frame
    v = got[0][0].reshape(5, 6, 64, 64)
    assert np.allclose(v.sum(axis=(2, 3)), v.sum(axis=(2, 3))[:, :1], rtol=1e-6)
    tr, te = M.load_video(6, 5, data=(train, dev, test))
    x, y = next(iter(tr()))
    assert x.shape == (5, 6, 4096) and y.shape == (5,)


def test_svhn_loader_layout_labels_and_mat_files(tmp_path):
    """tflib/svhn.py:33-53: X[32,32,3,N] -> rows in (C,H,W) order, label 10 -> 0, pairs survive the shuffles; also through real
    .mat files written with scipy."""
    from graphical_gan_amd.tflib import svhn
    from scipy.io import savemat
    rng = np.random.default_rng(2)
    X = rng.integers(0, 256, size=(32, 32, 3, 24)).astype(np.uint8)
    y = (np.arange(24) % 10 + 1).reshape(-1, 1).astype(np.uint8)           # 1..10 as in the files
    X[0, 0, 0, :] = y.reshape(-1)                                          # marker: row[0] (c=0,h=0,w=0) carries the raw label
    tr, te = svhn.load(8, '/nonexistent', data=((X, y), (X[..., :8], y[:8])))
    n = 0
    for x, t in tr():
        assert x.shape == (8, 3072) and x.dtype == np.uint8
        assert np.array_equal(x[:, 0] % 10, t)                            # 10 -> 0, everything else unchanged
        n += 1
    assert n == 3 and sum(1 for _ in te()) == 1
    # element order: row[c*1024 + h*32 + w] == X[h, w, c, i]
    tr2, _ = svhn.load(24, '/nonexistent', data=((X, y), (X[..., :8], y[:8])))
    np.random.seed(0)
    xb, tb = next(iter(tr2()))
    i = int(np.where((X[0, 0, 0, :] == xb[0, 0]) & (X[5, 7, 2, :] == xb[0, 2 * 1024 + 5 * 32 + 7]))[0][0])
    assert np.array_equal(xb[0].reshape(3, 32, 32), X[..., i].transpose(2, 0, 1))
    d = tmp_path / 'svhn'
    d.mkdir()
    savemat(str(d / 'train_32x32.mat'), {'X': X, 'y': y})
    savemat(str(d / 'test_32x32.mat'), {'X': X[..., :8], 'y': y[:8]})
    tr3, te3 = svhn.load(8, str(d))
    assert sum(1 for _ in tr3()) == 3 and next(iter(te3()))[0].shape == (8, 3072)
    import pytest
    with pytest.raises(FileNotFoundError):
        svhn.load(8, '/nonexistent')


def test_chairs_loader_layout_and_clips(tmp_path):
    """tflib/chairs.py:36-44: [objects, 31, H, W, 3] -> [B, seq, 3*H*W] in (C,H,W) order, dev split first, whole sequences / leading
    views / random 4-view clips; also through a real .npy file."""
    from graphical_gan_amd.tflib import chairs
    rng = np.random.default_rng(4)
    size, n = 8, 12
    data = rng.integers(0, 256, size=(n, 31, size, size, 3)).astype(np.uint8)
    data[:, :, 0, 0, 0] = np.arange(n)[:, None]               # object id in pixel (0,0) of channel 0
    data[:, :, 0, 1, 0] = np.arange(31)[None, :]              # view index in pixel (0,1) of channel 0
    np.random.seed(1)
    tr, dev = chairs.load(31, 4, size, '/nonexistent', num_dev=4, data=data)
    xb = next(iter(tr()))
    assert xb.shape == (4, 31, 3 * size * size) and xb.dtype == np.float32
    assert np.array_equal(xb[:, :, 1], np.tile(np.arange(31, dtype=np.float32), (4, 1)))      # views in order, (C,H,W) layout
    obj = int(xb[0, 0, 0])
    assert np.array_equal(xb[0, 5].reshape(3, size, size), data[obj, 5].transpose(2, 0, 1).astype(np.float32))
    assert sum(1 for _ in tr()) == (n - 4) // 4 and sum(1 for _ in dev()) == 1
    tr16, _ = chairs.load(16, 4, size, '/nonexistent', num_dev=4, data=data)
    x16 = next(iter(tr16()))
    assert x16.shape == (4, 16, 3 * size * size) and np.array_equal(x16[0, :, 1], np.arange(16, dtype=np.float32))
    tr4, _ = chairs.load(4, 4, size, '/nonexistent', num_dev=4, data=data)
    x4 = next(iter(tr4()))
    assert x4.shape == (4, 4, 3 * size * size) and np.all(np.diff(x4[:, :, 1], axis=1) == 1)   # consecutive views of one object
    np.save(str(tmp_path / ('chairs_%d.npy' % size)), data)
    tr2, _ = chairs.load(31, 2, size, str(tmp_path), num_dev=2)
    assert next(iter(tr2())).shape == (2, 31, 3 * size * size)
    import pytest
    with pytest.raises(FileNotFoundError):
        chairs.load(31, 2, size, '/nonexistent')
