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
    return torch.device('cuda:0')
