"""CPU oracle for the Graphical-GAN training step -- TEST INFRASTRUCTURE ONLY.

PARITY: pinned to the reference at the level of COMPOSITION, UNPINNED at the TensorFlow-primitive boundary.
The reference (/root/reference, Python 2 + TensorFlow 1.x) ships no tests, golden vectors or fixtures,
and TensorFlow is absent, so the arithmetic of tf.nn.conv2d / conv2d_transpose / fused_batch_norm / AdamOptimizer ... cannot be
executed here: ops.py / tape.py restate it from the documented semantics, pinned only by
  (i)   analytic known-answer tests (SAME-pad tables, delta responses, Adam step-1 form),
  (ii)  float64 finite-difference gradient checks (incl. the GP double-backward),
  (iii) an independent cross-check against PyTorch-CPU primitives composed to TF semantics,
  (iv)  the numeric statements TensorFlow itself publishes for these ops (its conv2d_transpose SAME / stride-2 unit test, the SAME
        padding rule, the fused_batch_norm training formula, the AdamOptimizer update rule, the sigmoid cross-entropy formula):
        tests/test_oracle_cpu.py::test_tf_published_*, tests/test_ops_gpu.py::test_tf_published_conv2d_transpose_known_answer.
Everything ABOVE those primitives -- nets.py, objs.py, step.py, ssgan.py: which layers a net is made of, parameter names and
shapes, loss composition, var_lists, optimizer settings, the order of session.run calls -- is checked against the reference's OWN
Python: tests/golden/make_reference_trace.py runs the ten driver scripts and tflib (converted from Python 2 in memory) under
tf1_shim.py, a TF1 graph-mode API surface bound to this package's primitives, and commits param_manifest.json /
reference_trace.json; tests/test_reference_trace_cpu.py replays them on the restatement (float64, 1e-9).
Those checks live in tests/test_oracle_*.py and tests/test_reference_trace_*.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (graphical_gan_amd/) never does; it fails loudly when the HIP
library is missing.
"""
from . import ops, tape, nets, objs, step  # noqa: F401   (tf1_shim / reftrace: imported by the trace generator and its tests)
