"""CPU oracle for the Graphical-GAN training step -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference (/root/reference, Python 2 + TensorFlow 1.x) can be
neither imported nor compiled in this image, and it ships no tests, golden vectors or
fixtures.  This package is therefore a *restatement* of
the arithmetic the reference's TF call sites select, pinned only by
  (i)   analytic known-answer tests (SAME-pad tables, delta responses, Adam step-1 form),
  (ii)  float64 finite-difference gradient checks (incl. the GP double-backward),
  (iii) an independent cross-check against PyTorch-CPU primitives composed to TF semantics.
Those checks live in tests/test_oracle_*.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (graphical_gan_amd/) never does; it fails loudly when the HIP
library is missing.
"""
from . import ops, tape, nets, objs, step  # noqa: F401
