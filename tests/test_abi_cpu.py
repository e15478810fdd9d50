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
    assert L.ggan_version() == _lib.ABI_VERSION == int(re.search(r'#define GGAN_ABI_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'ggan.h')).read()).group(1))
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


def test_launch_plan_travels_in_the_geometry_struct(lib_built):
    """Round 4: no process-wide launch-plan setters (ggan_set_target_workgroups*, ggan_set_naive are gone); the plan is three fields of
    ggan_conv_geom, filled per call from functional.target_workgroups / launch_hint / force_plain -- thread-local on the Python side."""
    import ctypes as C
    import re
    import threading
    from graphical_gan_amd import functional as F, _lib
    L = _lib.load()
    for gone in ('ggan_set_naive', 'ggan_set_target_workgroups', 'ggan_set_target_workgroups_filter_grad'):
        assert not hasattr(L, gone) or gone not in _lib.SIGNATURES
        assert gone not in open(os.path.join(ROOT, 'include', 'ggan.h')).read()
    hdr = open(os.path.join(ROOT, 'include', 'ggan.h')).read()
    body = re.search(r'typedef struct \{(.*?)\} ggan_conv_geom;', hdr, re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = [f.strip() for decl in re.findall(r'int ([^;]*);', body) for f in decl.split(',')]
    assert fields == [n for n, _ in _lib.ConvGeom._fields_], fields
    assert C.sizeof(_lib.ConvGeom) == 4 * len(fields)
    t = F.conv_geom(4, 8, 16, 16, 16, 5, 2)
    g = F._geom(t)
    assert (g.plan_wgs, g.plan_wgs_filter, g.plan_flags) == (0, 0, 0)
    with F.launch_hint(128):
        assert (F._geom(t).plan_wgs, F._geom(t).plan_wgs_filter) == (128, 128)       # (round 5: the hint plans the filter gradient too)
        with F._planned_for(96):
            assert (F._geom(t).plan_wgs, F._geom(t).plan_wgs_filter) == (96, 96)
        seen = []
        th = threading.Thread(target=lambda: seen.append(F._geom(t).plan_wgs))     # another thread: its own (default) plan
        th.start(); th.join()
        assert seen == [0]
    old = F.force_plain(True)
    try:
        assert F._geom(t).plan_flags == _lib.PLAN_PLAIN
    finally:
        F.force_plain(old)


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
