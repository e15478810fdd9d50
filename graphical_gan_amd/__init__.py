"""graphical_gan_amd -- MI355X-native hot path of zhenxuan00/graphical-gan.

  csrc/ + libggan.so   hand-written gfx950 HIP kernels behind the C ABI of include/ggan.h
  functional/          torch.autograd bindings, one module per op family (torch = memory / streams / tape only)
  tflib/               host-side mirror of the reference's `tflib` operator API (drop-in module paths)
  optim.py             TF-flavoured Adam on flat buffers + RCCL gradient exchange
  engine.py            step scheduler: gen step / critic steps, HIP-graph capture
  models/              counterparts of the reference driver scripts' net definitions

Importing this package also registers `tflib` (and its submodules) in sys.modules so that code written
against the reference -- `import tflib as lib; import tflib.ops.linear` -- resolves to this implementation.
"""
import sys as _sys

from . import _lib  # noqa: F401
from . import tflib as _tflib


def _alias_tflib():
    prefix = __name__ + '.tflib'
    for name, mod in list(_sys.modules.items()):
        if name == prefix or name.startswith(prefix + '.'):
            _sys.modules.setdefault('tflib' + name[len(prefix):], mod)


_alias_tflib()
