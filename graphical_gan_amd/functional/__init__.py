"""torch.autograd bindings of the libggan C ABI.

torch supplies device memory, the current HIP stream and the autograd tape (the role tf.gradients plays in
the reference); every piece of arithmetic is a HIP kernel reached through include/ggan.h.  Each backward
is itself built from these Functions, so the gradient-penalty double backward works:
  conv_fwd'   = (conv_dgrad, conv_wgrad)        conv_dgrad' = (conv_fwd, conv_wgrad)
  conv_wgrad' = (conv_dgrad, conv_fwd)          gemm'       = (gemm, gemm)
There is no CPU path: tensors must live on a HIP device and libggan.so must load.
"""

# one module per op family (round-4 review: the 2 190-line functional.py split); this package re-exports every name, private helpers
# included, so that `from graphical_gan_amd import functional as F` reads as before
from ._core import (  # noqa: F401
    C, os, weakref, torch, Function, once_differentiable, _lib, ACT_NONE, ACT_LRELU, ACT_RELU, ACT_TANH, ACT_SIGMOID, ConvGeom,
    check, _WS, _WS_BYTES, _L, _p, _stream, _dev, _c, _os, FUSED_CONV_BWD, _DEFER, _DATA_ONLY, _is_param, data_grad_only,
    _skip_undefined, defer_wgrad_reduce, _wgrad_parts, _STREAMS, shared_stream, workspace, _TARGET, _SERIAL_BWD, serial_backward,
    _bwd_target, target_workgroups, _threading, _PLAN, _HINT_FILTER, _PLAIN, force_plain, launch_hint, _carries_hint, _planned_for,
    same_geometry, conv_geom, _geom, _SITE, site_scope, set_site_plan, record_sites, site_log, site_mismatches, RowSlot, _new_out, _adjacent, HEAD_LOGITS, _PENDING_COSTS, _tail_value, settle_cost,
    pending_costs, drop_pending_costs, UNIT_SEEDS, unit_seed, is_unit_seed, LATE_EXT, _LATE_TERMS, mark_ready, wait_ready, add_late_terms,
    drop_late_terms)
from .pointwise import ActFwd, ActBwd, leaky_relu, relu, tanh, sigmoid  # noqa: F401
from .conv import (  # noqa: F401
    DEBUG_POISON_CHECK, PendingCast, ConvFwd, _fused_conv_backward, ConvDgrad, ConvDgradMasked, ConvWgrad, ChanSum, TALL_ROWS,
    ColSum)
from .linear import (  # noqa: F401
    Gemm, _fused_linear_backward, Gemm2, _CONSTS, cached_const, Gemm2Dgrad, _HEAD_HINT, head_bce_hint, CriticHead, DynScan,
    gemm_colsum_, linear)
from .norm import BatchNormTrain, BatchNormBwd, LinearBatchNormRows, _all_gather_rows, SyncBatchNormTrain  # noqa: F401
from .rows import (  # noqa: F401
    JoinRows, Fanout, fanout, SplitRows, CastScaleI32, Axpby, MixMean, GmmLatent, MixRbfMmd2, Reparam, AGG_KL, AGG_IKL, AGG_JSD,
    AggDiv, RowLerp)
from .conv3d import _dims3, Im2Col3d, Col2Im3d, _conv3d_patch, _igemm_ok, Conv3dImplicit, conv3d  # noqa: F401
from .losses import BceSum, Distance, MeanSum, GradPenalty  # noqa: F401
from .optim_ops import adam_step_, rmsprop_step_, NOISE_NORMAL, NOISE_UNIFORM, NOISE_ONEHOT, noise_state, noise_fill_, pack_  # noqa: F401
