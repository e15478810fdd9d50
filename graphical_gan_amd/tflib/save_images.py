"""Sample / reconstruction grid dump with the interface of tflib/save_images.py:53-87 (`save_images(X, save_path, size=None)`;
`large_image` returns the grid array).  Floats in [0,1] are mapped with 255.99*x as in the reference; the grid is nh x nw with
nh the largest divisor of the sample count not above its square root; [B,C,H,W], [B,H,W] and flattened [B,H*W] inputs.
The PNG is written with zlib directly (no scipy.misc / imageio in this image)."""
import struct
import zlib

import numpy as np


def large_image(X, size=None):
    X = np.asarray(X)
    if np.issubdtype(X.dtype, np.floating):
        X = (255.99 * X).astype('uint8')
    n_samples = X.shape[0]
    if size is None:
        rows = int(np.sqrt(n_samples))
        while n_samples % rows != 0:
            rows -= 1
        nh, nw = rows, n_samples // rows
    else:
        nh, nw = size
        assert nh * nw == n_samples
    if X.ndim == 2:
        s = int(np.sqrt(X.shape[1]))
        X = X.reshape(X.shape[0], s, s)
    if X.ndim == 4:
        X = X.transpose(0, 2, 3, 1)          # BCHW -> BHWC
        h, w = X[0].shape[:2]
        img = np.zeros((h * nh, w * nw, X.shape[3]), dtype='uint8')
    elif X.ndim == 3:
        h, w = X[0].shape[:2]
        img = np.zeros((h * nh, w * nw), dtype='uint8')
    else:
        raise ValueError('unsupported sample array shape %r' % (X.shape,))
    for n, x in enumerate(X):
        j, i = n // nw, n % nw
        img[j * h:j * h + h, i * w:i * w + w] = x
    return img


def write_png(path, img):
    """8-bit greyscale [H,W] / [H,W,1] or RGB [H,W,3]"""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    h, w = img.shape[:2]
    ctype = 0 if img.ndim == 2 else 2
    raw = b''.join(b'\x00' + img[r].tobytes() for r in range(h))

    def chunk(tag, data):
        c = struct.pack('>I', len(data)) + tag + data
        return c + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def save_images(X, save_path, size=None):
    write_png(save_path, large_image(X, size))
