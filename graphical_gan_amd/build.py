"""Build libggan.so (all HIP kernels + the C ABI of include/ggan.h) in-tree for gfx950.

hipcc cross-compiles without a GPU; the resulting .so sits next to csrc/ so that it travels with the
repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libggan.so')
STAMP = os.path.join(HERE, '.libggan.stamp')
SOURCES = ['runtime.hip', 'pointwise.hip', 'bn.hip', 'linear_bn.hip', 'gemm.hip', 'conv_naive.hip', 'conv_corr.hip', 'conv_dg16.hip', 'conv_wgrad.hip', 'conv_wgrad_split.hip', 'conv_thin.hip', 'conv_api.hip', 'conv3d.hip', 'dynscan.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-Wall', '-Wno-unused-function']
EXTRA_FLAGS = {'conv_wgrad_split.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}
if os.environ.get('GGAN_BUILD_DIAG'):       # timing experiments only: compiles GGAN_SKIP_KERNELS in
    FLAGS.append('-DGGAN_DIAG')


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, '..', 'include')):
        for fn in sorted(os.listdir(root)):
            with open(os.path.join(root, fn), 'rb') as f:
                h.update(fn.encode())
                h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build_id():
    """what a profile of the timed step depends on: the digest of csrc/ + include/ + flags, and the committed launch-site plans
    (profiles/pmc_traffic.json carries it as `_build`; bench.py says whether its static figures belong to the build it runs)"""
    h = hashlib.sha256()
    try:
        with open(os.path.join(HERE, 'site_plans.json'), 'rb') as f:
            h.update(f.read())
    except OSError:
        pass
    return _digest()[:16] + '/' + h.hexdigest()[:8]


def build(force=False, verbose=False):
    """Compile every .hip translation unit and link libggan.so.  Returns the library path.  Safe to call from several
    processes at once (one rank per GPU does): an exclusive file lock serialises them and the later ones find the stamp."""
    import fcntl
    with open(os.path.join(HERE, '.libggan.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src.replace('.hip', '.o'))
        os.makedirs(os.path.dirname(obj), exist_ok=True)
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write('hipcc failed for %s:\n%s\n' % (src, out.decode(errors='replace')))
        elif verbose and out:
            sys.stderr.write(out.decode(errors='replace'))
    if failed:
        raise RuntimeError('libggan build failed')
    tmp = LIB + '.tmp.%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)                       # atomic: a process that already mapped the old file keeps it
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
