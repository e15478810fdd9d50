import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def lib_built():
    """libggan.so built in-tree (hipcc cross-compiles without a GPU)."""
    from graphical_gan_amd import build
    return build.build()


@pytest.fixture(scope='session')
def gpu(lib_built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    from graphical_gan_amd import _lib
    _lib.load()          # fail loudly if the extension is missing on a GPU box
    if os.environ.get('GGAN_TEST_DP_ONE_RANK'):
        # a one-rank RCCL process group for this pytest process (tests/test_golden_full_gpu.py::test_full_size_dp_schedule_vs_fixture starts
        # such a process): with GGAN_FORCE_ALLREDUCE=1 every Trainer then builds the DATA-PARALLEL schedule -- gradient exchange captured
        # inside the step graph, pack -> all-reduce -> Adam
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(29800 + os.getpid() % 150))
            torch.cuda.set_device(0)
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    return torch.device('cuda:0')
