"""batch normalisation: NCHW and row-major (fused with Linear), sync-BN over the data-parallel group."""
import os as _os
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from .._lib import ACT_NONE, check  # noqa: F401
from ._core import _L, _p, _stream, _c, _skip_undefined  # noqa: F401
from .linear import Gemm, gemm_colsum_  # noqa: F401


@_skip_undefined
class BatchNormTrain(Function):
    """Training-mode BN over all axes but channel axis 1 (NCHW) or over axis 0 of [N,C]."""

    @staticmethod
    def forward(ctx, x, scale, offset, eps, act, alpha):
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        sc, of = _c(scale).reshape(-1), _c(offset).reshape(-1)
        y = torch.empty_like(x)
        mean = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        check(_L().ggan_bn_fwd_train(_p(x), _p(sc), _p(of), _p(y), _p(mean), _p(invstd), N, Cc, HW, eps, act, alpha,
                                     _stream()), 'ggan_bn_fwd_train')
        ctx.dims = (N, Cc, HW)
        ctx.act, ctx.alpha = act, alpha
        ctx.pshape = tuple(scale.shape)
        ctx.save_for_backward(x, sc, mean, invstd, y if act != ACT_NONE else None, scale)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, sc, mean, invstd, y, scale = ctx.saved_tensors
        N, Cc, HW = ctx.dims
        if torch.is_grad_enabled():        # a double backward is being recorded (gradient penalty through this layer)
            gx, gs, go = BatchNormBwd.apply(x, gy, y, scale.reshape(-1), mean, invstd, ctx.act, ctx.alpha, ctx.dims)
            return gx, gs.view(ctx.pshape), go.view(ctx.pshape), None, None, None
        gy = _c(gy)
        gx = torch.empty_like(x)
        gs = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        go = torch.empty_like(gs)
        csum = torch.empty_like(gs) if HW > 1 else None
        check(_L().ggan_bn_bwd_act(_p(x), _p(gy), _p(y) if ctx.act != ACT_NONE else _p(None), ctx.act, ctx.alpha, _p(sc),
                                   _p(mean), _p(invstd), _p(gx), _p(gs), _p(go), _p(csum), N, Cc, HW, _stream()),
              'ggan_bn_bwd_act')
        if csum is not None:
            gx._ggan_chansum = csum      # picked up by the producing layer's backward if gx reaches it unchanged
        return gx, gs.view(ctx.pshape), go.view(ctx.pshape), None, None, None


class BatchNormBwd(Function):
    """The first backward of BatchNormTrain as a differentiable op: only on the tape while a double backward is recorded (MODE
    vegan-wgan-gp: gradient penalty on a critic with BatchNorm).  Its own backward (ggan_bn_bwd_bwd) covers gradients arriving
    at gx; the scale / offset gradients are not differentiated again (no objective of the reference needs that)."""

    @staticmethod
    def forward(ctx, x, gy, y, sc, mean, invstd, act, alpha, dims):
        N, Cc, HW = dims
        gy = _c(gy)
        gx = torch.empty_like(x)
        gs = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        go = torch.empty_like(gs)
        check(_L().ggan_bn_bwd_act(_p(x), _p(gy), _p(y) if act != ACT_NONE else _p(None), act, alpha, _p(sc), _p(mean), _p(invstd),
                                   _p(gx), _p(gs), _p(go), _p(None), N, Cc, HW, _stream()), 'ggan_bn_bwd_act')
        ctx.dims, ctx.act, ctx.alpha = dims, act, alpha
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, gy, y, sc, mean, invstd)
        return gx, gs, go

    @staticmethod
    @once_differentiable
    def backward(ctx, h, hs, ho):
        if hs is not None or ho is not None:
            raise NotImplementedError('second derivative of the BatchNorm scale/offset gradients')
        if h is None:
            return (None,) * 9
        x, gy, y, sc, mean, invstd = ctx.saved_tensors
        N, Cc, HW = ctx.dims
        h = _c(h)
        ggy, gx2 = torch.empty_like(x), torch.empty_like(x)
        gsc = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        check(_L().ggan_bn_bwd_bwd(_p(x), _p(gy), _p(y) if ctx.act != ACT_NONE else _p(None), ctx.act, ctx.alpha, _p(h), _p(sc),
                                   _p(mean), _p(invstd), _p(ggy), _p(gx2), _p(gsc), N, Cc, HW, _stream()), 'ggan_bn_bwd_bwd')
        return gx2, ggy, None, gsc.view(sc.shape), None, None, None, None, None


class LinearBatchNormRows(Function):
    """y = act(BN_rows(x @ W + b)) in ONE launch (ggan_linear_bn_rows_fwd): Linear 'Generator.Input' + Batchnorm 'Generator.BN1' over the
    batch axis + relu (gan_inference_cifar10.py:134-138).  The backward is the composition's: ggan_bn_bwd_act on the kept Linear
    output, then the Linear layer's gradients (dW and db in one launch, the data gradient when the input needs one)."""

    @staticmethod
    def usable(x, w):
        """mirrors the limits of ggan_linear_bn_rows_fwd (linear_bn.hip): whole minibatch in one workgroup's LDS -- x [M, K + 4], the weight
        slice [K, 32] and the row-group partials must fit 160 KB (M = 128 with K >= 252 does not) -- rows in 16 equal groups, 16-byte
        aligned operands; anything else takes Linear + Batchnorm"""
        if x.dim() != 2 or _os.environ.get('GGAN_NO_LINEAR_BN'):
            return False
        M, K = x.shape
        N = w.shape[1]
        lds = (M * (K + 4) + K * 32 + 16 * 32) * 4
        return (M <= 128 and M % 16 == 0 and K <= 256 and K % 4 == 0 and N % 32 == 0 and x.is_contiguous() and w.is_contiguous()
                and lds <= 160 * 1024 and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)

    @staticmethod
    def forward(ctx, x, w, b, scale, offset, eps, act, alpha):
        x, w = _c(x), _c(w)
        M, K = x.shape
        N = w.shape[1]
        sc, of = _c(scale).reshape(-1), _c(offset).reshape(-1)
        bp = _p(_c(b)) if b is not None else _p(None)
        h = torch.empty((M, N), dtype=torch.float32, device=x.device)
        y = torch.empty_like(h)
        mean = torch.empty((N,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        check(_L().ggan_linear_bn_rows_fwd(_p(x), _p(w), bp, _p(sc), _p(of), _p(h), _p(y), _p(mean),
                                           _p(invstd), M, K, N, eps, act, alpha, _stream()), 'ggan_linear_bn_rows_fwd')
        ctx.act, ctx.alpha, ctx.has_bias, ctx.pshape = act, alpha, b is not None, tuple(scale.shape)
        ctx.save_for_backward(x, w, h, sc, mean, invstd, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, h, sc, mean, invstd, y = ctx.saved_tensors
        M, N = h.shape
        gy = _c(gy)
        gs = torch.empty((N,), dtype=torch.float32, device=h.device)
        go = torch.empty_like(gs)
        need = ctx.needs_input_grad
        if (not need[0] and need[1] and mean.dim() == 1 and x.shape[1] in (64, 128, 256) and M <= 128 and M % 16 == 0 and N % 32 == 0
                and x.data_ptr() % 16 == 0 and not _os.environ.get('GGAN_NO_LINEAR_BN_BWD')):
            # the input is noise (no data gradient): BatchNorm's backward and the weight-gradient product in one launch
            dw = torch.empty((x.shape[1], N), dtype=torch.float32, device=h.device)
            db = torch.empty((N,), dtype=torch.float32, device=h.device) if (ctx.has_bias and need[2]) else None
            rc = _L().ggan_linear_bn_rows_bwd(_p(x), _p(gy), _p(h), _p(y) if ctx.act != ACT_NONE else _p(None), _p(sc), _p(mean), _p(invstd),
                                              _p(dw), _p(db), _p(gs), _p(go), M, x.shape[1], N, ctx.act, ctx.alpha, _stream())
            if rc != 1:
                check(rc, 'ggan_linear_bn_rows_bwd')
                return (None, dw, db, gs.view(ctx.pshape) if need[3] else None, go.view(ctx.pshape) if need[4] else None, None, None, None)
        gh = torch.empty_like(h)
        check(_L().ggan_bn_bwd_act(_p(h), _p(gy), _p(y) if ctx.act != ACT_NONE else _p(None), ctx.act, ctx.alpha, _p(sc), _p(mean),
                                   _p(invstd), _p(gh), _p(gs), _p(go), _p(None), M, N, 1, _stream()), 'ggan_bn_bwd_act')
        dx = dw = db = None
        if need[1] or (ctx.has_bias and need[2]):
            dw, db = gemm_colsum_(x, gh, True)                         # dW = x^T gh and db = column sums of gh in one launch
            if not need[1]:
                dw = None
            if not (ctx.has_bias and need[2]):
                db = None
        if need[0]:
            dx = Gemm.apply(gh, w, None, False, True, ACT_NONE, 0.0)    # gh W^T
        return (dx, dw, db, gs.view(ctx.pshape) if need[3] else None, go.view(ctx.pshape) if need[4] else None, None, None, None)


def _all_gather_rows(t, group):
    """[world, *t.shape]: every replica's `t`, in rank order"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    from .. import rccl
    # (rccl.get_stats, called by Trainer(sync_bn=True): the gradient buckets' communicator in serial mode -- this gather waits for a bucket
    #  that is still on the wire on the communicator's stream, and the next bucket waits for it)
    comm = rccl._STATS[0] if (t.is_cuda and (group is None or group is dist.group.WORLD)) else None
    if comm is not None:             # an enqueue on the current stream (capturable: cross-replica BatchNorm inside a step graph)
        return comm.all_gather(out, t.contiguous())
    try:
        dist.all_gather_into_tensor(out, t, group=group)
    except (RuntimeError, NotImplementedError):
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=group)
        out = torch.stack(parts)
    return out


class SyncBatchNormTrain(Function):
    """BatchNormTrain with statistics over the global batch of a process group of equal-sized replicas: local
    statistics kernel -> all-gather of 2*C floats -> normalisation kernel, the same split in the backward."""

    @staticmethod
    def forward(ctx, x, scale, offset, eps, act, alpha, group):
        import torch.distributed as dist
        x = _c(x)
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        y = torch.empty_like(x)
        mean = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        sc, of = _c(scale).reshape(-1), _c(offset).reshape(-1)
        st = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        check(_L().ggan_bn_sync_stats(_p(x), _p(st), N, Cc, HW, _stream()), 'ggan_bn_sync_stats')
        allst = _all_gather_rows(st, group)
        world = allst.shape[0]
        check(_L().ggan_bn_sync_apply(_p(x), _p(allst), world, _p(sc), _p(of), _p(y), _p(mean), _p(invstd), N, Cc, HW, eps, act,
                                      alpha, _stream()), 'ggan_bn_sync_apply')
        ctx.dims = (N, Cc, HW)
        ctx.act, ctx.alpha = act, alpha
        ctx.group, ctx.world, ctx.rank = group, world, dist.get_rank(group)
        ctx.pshape = tuple(scale.shape)
        ctx.save_for_backward(x, sc, mean, invstd, y if act != ACT_NONE else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, sc, mean, invstd, y = ctx.saved_tensors
        N, Cc, HW = ctx.dims
        gy = _c(gy)
        yp = _p(y) if ctx.act != ACT_NONE else _p(None)
        sums = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        check(_L().ggan_bn_sync_bwd_stats(_p(x), _p(gy), yp, ctx.act, ctx.alpha, _p(mean), _p(invstd), _p(sums), N, Cc, HW,
                                          _stream()), 'ggan_bn_sync_bwd_stats')
        allsums = _all_gather_rows(sums, ctx.group)
        gx = torch.empty_like(x)
        gs = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        go = torch.empty_like(gs)
        check(_L().ggan_bn_sync_bwd_apply(_p(x), _p(gy), yp, ctx.act, ctx.alpha, _p(sc), _p(mean), _p(invstd), _p(allsums),
                                          ctx.world, ctx.rank, _p(gx), _p(gs), _p(go), N, Cc, HW, _stream()),
              'ggan_bn_sync_bwd_apply')
        return gx, gs.view(ctx.pshape), go.view(ctx.pshape), None, None, None, None
