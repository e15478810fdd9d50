"""tflib/objs/kl_aggregated.py: the critic-free objectives of MODE vegan-kl / vegan-ikl / vegan-jsd (gan_inference_cifar10.py:331-341):
lamb * D(q(z) || p(z)) + rec_penalty with q(z) the aggregated posterior -- the equal-weight mixture of the minibatch's diagonal Gaussians
(q_z_mean, q_z_std) -- and D estimated on z_samples Monte-Carlo samples.  One fused kernel pair per divergence (ggan_agg_div_fwd/bwd).

Same signatures as the reference plus `draws`: the reference samples inside the objective (Categorical / random_normal nodes); here the
draws are inputs -- (k_onehot [z_samples, batchsize], eps [z_samples, dim_z], z_p [z_samples, dim_z]) -- so that a step graph refreshes
them with the rest of the step's noise (functional.noise_fill_).  The prior is N(0, I): the scripts pass p_z_mean = 0, p_z_std = 1
constants (gan_inference_cifar10.py:269-270) and the kernels have that prior built in; other values are rejected at build time."""
import torch

from ... import functional as F
from ...optim import TrainOp, get_optimizer


def _draws(draws, z_samples, n_coms, dim_z, like):
    if draws is not None:
        return draws
    k = torch.empty((z_samples, n_coms), dtype=torch.float32, device=like.device)
    eps, z_p = (torch.empty((z_samples, dim_z), dtype=torch.float32, device=like.device) for _ in range(2))
    F.noise_fill_(F.noise_state(like.device), [(k, F.NOISE_ONEHOT, 0., 0.), (eps, F.NOISE_NORMAL, 0., 1.), (z_p, F.NOISE_NORMAL, 0., 1.)])
    return k, eps, z_p


def _check_prior(p_z_mean, p_z_std):
    from .. import _build_phase
    if p_z_mean is None and p_z_std is None:
        return
    if _build_phase[0] and not torch.cuda.is_current_stream_capturing():       # (a host read: only while the step is being built)
        if float(p_z_mean.abs().max()) != 0.0 or float((p_z_std - 1).abs().max()) != 0.0:
            raise NotImplementedError('kl_aggregated: only the N(0, I) prior of the scripts is built')


def kl_q_aggregated_p_diagonal_gaussian(q_z_mean, q_z_std, p_z_mean, p_z_std, n_samples, n_coms, dim_z, draws=None):
    """:46-51"""
    _check_prior(p_z_mean, p_z_std)
    k, eps, z_p = _draws(draws, n_samples, n_coms, dim_z, q_z_mean)
    return F.AggDiv.apply(q_z_mean, q_z_std, k, eps, z_p, F.AGG_KL, n_coms)


def ikl_q_aggregated_p_diagonal_gaussian(q_z_mean, q_z_std, p_z_mean, p_z_std, n_samples, dim_z, draws=None):
    """:53-58"""
    _check_prior(p_z_mean, p_z_std)
    k, eps, z_p = _draws(draws, n_samples, q_z_mean.shape[0], dim_z, q_z_mean)
    return F.AggDiv.apply(q_z_mean, q_z_std, k, eps, z_p, F.AGG_IKL, q_z_mean.shape[0])


def jsd_q_aggregated_p_diagonal_gaussian(q_z_mean, q_z_std, p_z_mean, p_z_std, n_samples, n_coms, dim_z, draws=None):
    """:60-71"""
    _check_prior(p_z_mean, p_z_std)
    k, eps, z_p = _draws(draws, n_samples, n_coms, dim_z, q_z_mean)
    return F.AggDiv.apply(q_z_mean, q_z_std, k, eps, z_p, F.AGG_JSD, n_coms)


def _finish(div, rec_penalty, gen_params, lamb, lr, beta1):
    gen_cost = div * float(lamb) + rec_penalty
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=0.999)
    return gen_cost, TrainOp(gen_opt, gen_cost)


def vegan_jsd(q_z_mean, q_z_std, p_z_mean, p_z_std, rec_penalty, gen_params, z_samples, batchsize, dim_z, lamb, lr=2e-4, beta1=.5, draws=None):
    """:73-82"""
    return _finish(jsd_q_aggregated_p_diagonal_gaussian(q_z_mean, q_z_std, p_z_mean, p_z_std, z_samples, batchsize, dim_z, draws),
                   rec_penalty, gen_params, lamb, lr, beta1)


def vegan_kl(q_z_mean, q_z_std, p_z_mean, p_z_std, rec_penalty, gen_params, z_samples, batchsize, dim_z, lamb, lr=2e-4, beta1=.5, draws=None):
    """:84-93"""
    return _finish(kl_q_aggregated_p_diagonal_gaussian(q_z_mean, q_z_std, p_z_mean, p_z_std, z_samples, batchsize, dim_z, draws),
                   rec_penalty, gen_params, lamb, lr, beta1)


def vegan_ikl(q_z_mean, q_z_std, p_z_mean, p_z_std, rec_penalty, gen_params, z_samples, dim_z, lamb, lr=2e-4, beta1=.5, draws=None):
    """:95-102"""
    return _finish(ikl_q_aggregated_p_diagonal_gaussian(q_z_mean, q_z_std, p_z_mean, p_z_std, z_samples, dim_z, draws),
                   rec_penalty, gen_params, lamb, lr, beta1)
