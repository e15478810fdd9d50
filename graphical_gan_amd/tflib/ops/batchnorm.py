"""Batchnorm (tflib/ops/batchnorm.py:6-87).  The scripts never pass `is_training`, so the reference always
runs the training branch with batch statistics (:51-52); moving statistics are created (and returned by
params_with_name) but never written.  axes [0,2,3] -> fused NCHW branch, axes [0] -> the non-fused branch with
[1,F]-shaped parameters."""
import numpy as np

from ... import functional as F
from .. import param as _param
from .. import tap as _tap

_SYNC_GROUP = None      # False/None: per-replica statistics (the reference's single-GPU behaviour); else a process group


def set_sync_group(group):
    """Batch statistics over the global batch of `group` (torch.distributed process group, or True for the default group) from
    now on; None switches back.  SURVEY.md 8(e): the mode in which N GPUs x B/N reproduce 1 GPU x B.  Returns the old value."""
    global _SYNC_GROUP
    old, _SYNC_GROUP = _SYNC_GROUP, group
    if group is None or group is False:
        from ... import rccl
        rccl.release_stats()         # (the shared communicator leaves serial mode with the statistics exchange)
    return old


def _bn(x, scale, offset, act, alpha):
    if _SYNC_GROUP is None or _SYNC_GROUP is False:
        return F.BatchNormTrain.apply(x, scale, offset, 1e-5, act, float(alpha))
    import torch.distributed as dist
    group = dist.group.WORLD if _SYNC_GROUP is True else _SYNC_GROUP
    return F.SyncBatchNormTrain.apply(x, scale, offset, 1e-5, act, float(alpha), group)


def Batchnorm(name, axes, inputs, is_training=None, stats_iter=None, update_moving_stats=True, fused=True,
              activation=None, alpha=0.2):
    act = F.ACT_NONE if activation is None else activation
    if is_training is not None:
        raise NotImplementedError('Batchnorm(is_training=...) is not used by any reference script and is not built')
    axes = list(axes)
    if (axes == [0, 2, 3] or axes == [0, 2]) and fused is True:
        x = inputs.unsqueeze(3) if axes == [0, 2] else inputs
        c = x.shape[1]
        offset = _param(name + '.offset', np.zeros(c, dtype='float32'))
        scale = _param(name + '.scale', np.ones(c, dtype='float32'))
        _param(name + '.moving_mean', np.zeros(c, dtype='float32'), trainable=False)
        _param(name + '.moving_variance', np.ones(c, dtype='float32'), trainable=False)
        out = _bn(x, scale, offset, act, alpha)
        if act in (F.ACT_LRELU, F.ACT_RELU):
            _tap(name, out)
        return out[:, :, :, 0] if axes == [0, 2] else out
    if axes == [0] and inputs.dim() == 2:
        shape = [1, inputs.shape[1]]
        offset = _param(name + '.offset', np.zeros(shape, dtype='float32'))
        scale = _param(name + '.scale', np.ones(shape, dtype='float32'))
        return _bn(inputs, scale, offset, act, alpha)
    raise Exception('unsupported')
