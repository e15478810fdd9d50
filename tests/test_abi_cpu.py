"""not-gpu: libggan builds for gfx950 without a GPU, loads, and exports every symbol include/ggan.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'ggan.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ggan_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound(lib_built):
    import ctypes
    from graphical_gan_amd import _lib
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(lib_built)
    for n in names:
        assert hasattr(lib, n), 'libggan.so does not export %s' % n
    assert sorted(_lib.SIGNATURES) == names, set(names) ^ set(_lib.SIGNATURES)
    L = _lib.load()
    assert L.ggan_version() >= 100
    assert L.ggan_last_error() is not None


def test_graft_entry_build():
    import __graft_entry__ as ge
    path = ge.build()
    assert os.path.exists(path)


def test_argument_errors_do_not_need_a_gpu(lib_built):
    """Argument validation happens before any launch: bad geometry -> negative rc + message."""
    import ctypes as C
    from graphical_gan_amd import _lib
    L = _lib.load()
    g = _lib.ConvGeom(1, 1, 4, 4, 1, 9, 9, 5, 2, 1, 1)       # output grid larger than the input allows
    rc = L.ggan_conv2d_fwd(C.byref(g), C.c_void_p(8), C.c_void_p(8), None, C.c_void_p(8), 0, 0.0, None, 0, None)
    assert rc < 0 and b'geometry' in L.ggan_last_error()
    assert L.ggan_gemm(0, 0, 0, 1, 1, C.c_void_p(8), C.c_void_p(8), None, C.c_void_p(8), 0, 0.0, None, 0, None) < 0


def test_ops_refuse_cpu_tensors(lib_built):
    """No CPU fallback: the product ops raise on CPU tensors instead of computing something else."""
    import pytest
    import torch
    from graphical_gan_amd import functional as F, _lib
    x = torch.zeros(2, 3)
    with pytest.raises(_lib.GganError):
        F.relu(x)
    with pytest.raises(_lib.GganError):
        F.Gemm.apply(x, torch.zeros(3, 4), None, False, False, 0, 0.0)
