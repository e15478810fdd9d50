"""row glue of the objectives: joins / splits / fan-out of row blocks, mixture and reparameterisation ops, the wali-gp interpolates."""
import ctypes as C
import os
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from .._lib import ACT_NONE, check  # noqa: F401
from ._core import _L, _p, _stream, _dev, _c, _DATA_ONLY, _is_param, _skip_undefined, _new_out, _adjacent  # noqa: F401
from .linear import Gemm  # noqa: F401


class JoinRows(Function):
    """cat([a, b], 0) for the critic evaluated once on [fake; real].  When the two operands already sit back to back in one
    buffer (their producers were handed RowSlots) the result is an alias of that memory: no copy kernel; the backward hands
    out the two row ranges of the incoming gradient (views)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.n = a.shape[0]
        if _adjacent(a, b):
            out = torch.empty(0, dtype=a.dtype, device=a.device)
            out.set_(a.untyped_storage(), a.storage_offset(), (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())
            return out
        return torch.cat([a, b], 0)

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.n], g[ctx.n:]


class Fanout(Function):
    """n aliases of a tensor that several branches of a step read (the mixture scripts: the code p_z feeds the Generator and the
    critics, q_z the mixture posterior and the critics, [p_z; q_z] both critics, the component means both hyper nets).  Their
    gradients are summed HERE, in alias order, by this library's pointwise launch (ggan_axpby through Axpby, differentiable) --
    not by at::add wherever autograd happens to meet the second contribution."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)      # (an alias nobody differentiates contributes None, not a zero tensor + an addition launch)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            acc = Axpby.apply(acc, g, 1.0, 1.0, 0.0)
        return acc, None


def fanout(x, n=2):
    """n aliases of x whose gradients meet in one Fanout node (x itself n times where no gradient can flow)"""
    if (not torch.is_tensor(x) or not x.is_cuda or not x.requires_grad or not torch.is_grad_enabled() or os.environ.get('GGAN_NO_FANOUT')):
        return (x,) * n
    return Fanout.apply(x, n)


class SplitRows(Function):
    """(x[:n], x[n:]) for the critic evaluated once on [fake; real]; the backward is ONE concatenation instead of two
    zero-padded slice gradients and their sum."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.rows = n, x.shape[0]
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        like = ga if ga is not None else gb
        if ga is None:
            ga = like.new_zeros((ctx.n,) + tuple(like.shape[1:]))
        if gb is None:
            gb = like.new_zeros((ctx.rows - ctx.n,) + tuple(like.shape[1:]))
        if (ga.is_contiguous() and gb.is_contiguous() and ga.dtype == gb.dtype
                and ga.untyped_storage().data_ptr() == gb.untyped_storage().data_ptr()
                and gb.storage_offset() == ga.storage_offset() + ga.numel()):
            # the two halves already sit back to back in one buffer (BceSum.backward): no copy
            return torch.as_strided(ga, (ctx.rows,) + tuple(ga.shape[1:]), ga.stride(), ga.storage_offset()), None
        return torch.cat([ga, gb], 0), None


@_skip_undefined
class CastScaleI32(Function):
    """real_x = mul*(float(x)/div - .5) + noise  (no gradient: the input is data)."""

    @staticmethod
    def forward(ctx, x_int, noise, div, mul, slot=None, ring=None):
        """ring: (int32 [R, ...] tensor of pre-staged minibatches, counter a, counter b, offset) -- the minibatch is slot
        (a + b + offset) mod R of the ring instead of x_int (ggan_cast_scale_ring_i32)"""
        _dev(x_int)
        assert x_int.dtype == torch.int32
        x_int = x_int.contiguous()
        y = _new_out(slot, x_int.shape, x_int.device)
        nz = _p(_c(noise)) if noise is not None else _p(None)
        if ring is not None:
            rt, ca, cb, off = ring
            assert rt.dtype == torch.int32 and rt.is_contiguous() and rt[0].numel() == x_int.numel()
            check(_L().ggan_cast_scale_ring_i32(_p(rt), rt.shape[0], _p(ca), _p(cb), int(off), nz, _p(y), x_int.numel(), div, mul,
                                                _stream()), 'ggan_cast_scale_ring_i32')
            return y
        check(_L().ggan_cast_scale_i32(_p(x_int), nz, _p(y), x_int.numel(), div, mul, _stream()), 'ggan_cast_scale_i32')
        return y

    @staticmethod
    def backward(ctx, g):
        return (None,) * len(ctx.needs_input_grad)


@_skip_undefined
class Axpby(Function):
    """out = a*x + b*y + c"""

    @staticmethod
    def forward(ctx, x, y, a, b, c, slot=None):
        x = _c(x)
        y = _c(y) if y is not None else None
        out = _new_out(slot, x.shape, x.device)
        check(_L().ggan_axpby(_p(x), _p(y), _p(out), x.numel(), a, b, c, _stream()), 'ggan_axpby')
        ctx.a, ctx.b, ctx.has_y = a, b, y is not None
        return out

    @staticmethod
    def backward(ctx, g):
        gx = Axpby.apply(g, None, ctx.a, 0.0, 0.0) if ctx.needs_input_grad[0] else None
        gy = Axpby.apply(g, None, ctx.b, 0.0, 0.0) if (ctx.has_y and ctx.needs_input_grad[1]) else None
        return (gx, gy) + (None,) * (len(ctx.needs_input_grad) - 2)


@_skip_undefined
class MixMean(Function):
    """p_z[B,D] = k[B,K] @ mu[K,D] + noise[B,D]: HyperGenerator of the gmgan scripts (gmgan_inference_cifar10.py:150-153) as ONE pointwise
    launch (ggan_mix_mean) instead of Gemm + Axpby at the head of the Generator chain; backward: d mu = k^T g (one product), d noise = g."""

    @staticmethod
    def usable(k, mu, noise):
        return k.dim() == 2 and mu.dim() == 2 and mu.shape[1] % 4 == 0 and k.is_cuda

    @staticmethod
    def forward(ctx, k, mu, noise, slot=None):
        k, mu, noise = _c(k), _c(mu), _c(noise)
        B, K = k.shape
        D = mu.shape[1]
        assert mu.shape[0] == K and tuple(noise.shape) == (B, D), (k.shape, mu.shape, noise.shape)
        out = _new_out(slot, (B, D), k.device)
        check(_L().ggan_mix_mean(_p(k), _p(mu), _p(noise), _p(out), B, K, D, _stream()), 'ggan_mix_mean')
        ctx.mu_param = _is_param(mu)
        ctx.save_for_backward(k, mu)
        return out

    @staticmethod
    def backward(ctx, g):
        k, mu = ctx.saved_tensors
        dk = dmu = None
        if ctx.needs_input_grad[1] and not (_DATA_ONLY[0] and ctx.mu_param):
            dmu = Gemm.apply(k, g, None, True, False, ACT_NONE, 0.0)          # k^T g
        if ctx.needs_input_grad[0]:
            dk = Gemm.apply(g, mu, None, False, True, ACT_NONE, 0.0)          # g mu^T
        return dk, dmu, (g if ctx.needs_input_grad[2] else None), None


class GmmLatent(Function):
    """HyperExtractor of the gmgan scripts in one launch per direction (ggan_gmm_latent_*): component logits of z under the
    mixture prior and the Gumbel-softmax relaxation of the component assignment.  Returns (logits, k)."""

    @staticmethod
    def forward(ctx, z, mu, gumbel_u, log_pi, temp, slot=None):
        z, mu, gumbel_u = _c(z), _c(mu), _c(gumbel_u)
        B, D = z.shape
        K = mu.shape[0]
        assert tuple(mu.shape) == (K, D) and tuple(gumbel_u.shape) == (B, K)
        logits = torch.empty((B, K), dtype=torch.float32, device=z.device)
        k = _new_out(slot, (B, K), z.device)
        check(_L().ggan_gmm_latent_fwd(_p(z), _p(mu), _p(gumbel_u), _p(logits), _p(k), B, K, D, float(log_pi), float(temp),
                                       _stream()), 'ggan_gmm_latent_fwd')
        ctx.temp = float(temp)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(z, mu, k)
        return logits, k

    @staticmethod
    @once_differentiable
    def backward(ctx, g_logits, g_k):
        z, mu, k = ctx.saved_tensors
        n_in = len(ctx.needs_input_grad)
        if (g_logits is None and g_k is None) or not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return (None,) * n_in
        B, D = z.shape
        K = mu.shape[0]
        dz = torch.empty_like(z) if ctx.needs_input_grad[0] else None
        dmu = torch.empty_like(mu) if ctx.needs_input_grad[1] else None
        gl = _c(g_logits) if g_logits is not None else None
        gk = _c(g_k) if g_k is not None else None
        check(_L().ggan_gmm_latent_bwd(_p(z), _p(mu), _p(k), _p(gl), _p(gk), _p(dz), _p(dmu), B, K, D, ctx.temp, _stream()),
              'ggan_gmm_latent_bwd')
        return (dz, dmu) + (None,) * (n_in - 2)


class MixRbfMmd2(Function):
    """biased MMD^2 between two sets of codes under a mixture of RBF kernels (tflib/objs/mmd.py:65-67) -> 0-dim tensor"""

    @staticmethod
    def forward(ctx, x, y, sigmas, wts):
        x, y = _c(x), _c(y)
        m, d = x.shape
        n = y.shape[0]
        assert y.shape[1] == d
        ns = len(sigmas)
        sg = (C.c_float * ns)(*[float(v) for v in sigmas])
        wt = (C.c_float * ns)(*[float(v) for v in wts]) if wts is not None else None
        out = torch.empty((), dtype=torch.float32, device=x.device)
        scratch = torch.empty((m + n,), dtype=torch.float32, device=x.device)
        check(_L().ggan_mix_rbf_mmd2_fwd(_p(x), _p(y), m, n, d, sg, wt, ns, _p(out), _p(scratch), _stream()), 'ggan_mix_rbf_mmd2_fwd')
        ctx.sg, ctx.wt, ctx.ns = sg, wt, ns
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        m, d = x.shape
        n = y.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        if dx is None and dy is None:
            return None, None, None, None
        check(_L().ggan_mix_rbf_mmd2_bwd(_p(x), _p(y), m, n, d, ctx.sg, ctx.wt, ctx.ns, _p(_c(g)), _p(dx), _p(dy), _stream()),
              'ggan_mix_rbf_mmd2_bwd')
        return dx, dy, None, None


class Reparam(Function):
    """(z, std) = (mean + eps * exp(log_std), exp(log_std)): the stochastic encoder head (gan_inference_cifar10.py:173-188)"""

    @staticmethod
    def forward(ctx, mean, log_std, eps):
        mean, log_std, eps = _c(mean), _c(log_std), _c(eps)
        z, sd = torch.empty_like(mean), torch.empty_like(mean)
        check(_L().ggan_reparam_fwd(_p(mean), _p(log_std), _p(eps), _p(z), _p(sd), mean.numel(), _stream()), 'ggan_reparam_fwd')
        ctx.save_for_backward(eps, sd)
        return z, sd

    @staticmethod
    @once_differentiable
    def backward(ctx, gz, gsd):
        eps, sd = ctx.saved_tensors
        gmean, glog = torch.empty_like(sd), torch.empty_like(sd)
        check(_L().ggan_reparam_bwd(_p(_c(gz)) if gz is not None else _p(None), _p(_c(gsd)) if gsd is not None else _p(None), _p(eps), _p(sd),
                                    _p(gmean), _p(glog), sd.numel(), _stream()), 'ggan_reparam_bwd')
        return gmean, glog, None


AGG_KL, AGG_IKL, AGG_JSD = 0, 1, 2


class AggDiv(Function):
    """Monte-Carlo KL / inverse KL / JSD between the aggregated posterior (mixture of the minibatch's diagonal Gaussians mu, sd [nx, d])
    and N(0, I) (tflib/objs/kl_aggregated.py:46-74) -> 0-dim tensor.  k_onehot [nz, nx], eps_q [nz, d]: the component draws and noise
    of the samples from q (kl, jsd); z_p [nz, d]: the samples from the prior (ikl, jsd)."""

    @staticmethod
    def forward(ctx, mu, sd, k_onehot, eps_q, z_p, kind, n_coms):
        mu, sd = _c(mu), _c(sd)
        nx, d = mu.shape
        nz = (z_p if kind != AGG_KL else eps_q).shape[0]
        ns = 2 * nz if kind == AGG_JSD else nz
        k_onehot = _c(k_onehot) if kind != AGG_IKL else None
        eps_q = _c(eps_q) if kind != AGG_IKL else None
        z_p = _c(z_p) if kind != AGG_KL else None
        assert k_onehot is None or (tuple(k_onehot.shape) == (nz, nx) and tuple(eps_q.shape) == (nz, d))
        assert z_p is None or tuple(z_p.shape) == (nz, d)
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=mu.device)
        out, Z, A, Bv, T = new(), new(ns, d), new(ns, nx), new(ns), new(ns)
        check(_L().ggan_agg_div_fwd(kind, _p(mu), _p(sd), _p(k_onehot), _p(eps_q), _p(z_p), nx, nz, d, int(n_coms), _p(out), _p(Z), _p(A),
                                    _p(Bv), _p(T), _stream()), 'ggan_agg_div_fwd')
        ctx.dims = (kind, nx, nz, d, int(n_coms), ns)
        ctx.save_for_backward(mu, sd, k_onehot, eps_q, Z, A, Bv)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        mu, sd, k_onehot, eps_q, Z, A, Bv = ctx.saved_tensors
        kind, nx, nz, d, n_coms, ns = ctx.dims
        gmu, gsd = torch.empty_like(mu), torch.empty_like(sd)
        W = torch.empty((ns, nx), dtype=torch.float32, device=mu.device)
        GZ = torch.empty((nz, d), dtype=torch.float32, device=mu.device)
        check(_L().ggan_agg_div_bwd(kind, _p(mu), _p(sd), _p(k_onehot), _p(eps_q), nx, nz, d, n_coms, _p(Z), _p(A), _p(Bv), _p(_c(g)),
                                    _p(W), _p(GZ), _p(gmu), _p(gsd), _stream()), 'ggan_agg_div_bwd')
        return gmu, gsd, None, None, None, None, None




@_skip_undefined
class RowLerp(Function):
    """out[r,:] = x[r,:] + alpha[r]*(y[r,:]-x[r,:])  (the wali-gp interpolates)."""

    @staticmethod
    def forward(ctx, x, y, alpha):
        x, y, alpha = _c(x), _c(y), _c(alpha)
        rows, cols = x.shape
        out = torch.empty_like(x)
        check(_L().ggan_row_lerp(_p(x), _p(y), _p(alpha), _p(out), rows, cols, _stream()), 'ggan_row_lerp')
        ctx.save_for_backward(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        z = torch.zeros_like(g)
        gx = RowLerp.apply(g, z, alpha) if ctx.needs_input_grad[0] else None
        gy = RowLerp.apply(z, g, alpha) if ctx.needs_input_grad[1] else None
        return gx, gy, None
