"""Objectives of tflib/objs/gan_inference.py: ali (:47-79), local_ep (:81-119), local_ep_dynamic (:246-305), weighted_local_epce (:307-358),
wali_gp (:28-45), and the reconstruction variants local_epce (:121-160), alice (:162-195), vegan (:197-225),
vegan_wgan_gp (:227-244).  wali (:4-26, RMSProp + weight clipping).  Same signatures and return tuples; `*_train_op` are
callables (TrainOp) that run backward + one TF-flavoured Adam step, the costs are 0-dim device tensors.
Losses are single fused kernels (ggan_bce_logits_* / ggan_mean_*), not per-term pointwise graphs."""
import numpy as np

from ... import functional as F
from ...optim import TrainOp, get_optimizer


ONLY = [None]   # 'gen' | 'disc': build just that cost (what one session.run fetches); None builds both
_val = F.settle_cost      # a cost is READ here: a hinted critic head that still owes its value computes it first (functional.settle_cost)


def _bce_costs(fakes, reals, ratios):
    """gen: fake->1, real->0 ; disc: fake->0, real->1 (sigmoid cross-entropy, mean over the batch)."""
    logits, gl, dl, w = [], [], [], []
    for f, r, ratio in zip(fakes, reals, ratios):
        logits += [f, r]
        gl += [1.0, 0.0]
        dl += [0.0, 1.0]
        w += [float(ratio), float(ratio)]
    gen_cost = F.BceSum.apply(tuple(gl), tuple(w), *logits) if ONLY[0] != 'disc' else None
    disc_cost = F.BceSum.apply(tuple(dl), tuple(w), *logits) if ONLY[0] != 'gen' else None
    return gen_cost, disc_cost


def ali(disc_fake, disc_real, gen_params, disc_params, lr=2e-4, beta1=0.5, beta2=0.999, s_f=None):
    gen_cost, disc_cost = _bce_costs([disc_fake], [disc_real], [1.0])
    if s_f is not None and gen_cost is not None:
        gen_cost = _val(gen_cost) + s_f
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=beta2)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=beta2)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def local_ep(disc_fake_list, disc_real_list, gen_params, disc_params, lr=2e-4, beta1=0.5, beta2=.999, s_f=None):
    n = float(len(disc_fake_list))
    if s_f is None:
        gen_cost, disc_cost = _bce_costs(disc_fake_list, disc_real_list, [1.0 / n] * len(disc_fake_list))
    else:   # (sum + s_f) / n, as gan_inference.py:102-106
        gen_cost, disc_cost = _bce_costs(disc_fake_list, disc_real_list, [1.0] * len(disc_fake_list))
        gen_cost = (_val(gen_cost) + s_f) / n if gen_cost is not None else None
        disc_cost = _val(disc_cost) / n if disc_cost is not None else None
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=beta2)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=beta2)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def _plus(cost, *terms):
    for t in terms:
        if t is not None and cost is not None:
            cost = _val(cost) + t
    return cost


def local_epce(disc_fake_list, disc_real_list, rec_penalty, gen_params, disc_params, lr=2e-4, beta1=0.5, s_f=None):
    """tflib/objs/gan_inference.py:121-160: (sum of BCE terms (+ s_f)) / n, then + rec_penalty on the generator side."""
    n = float(len(disc_fake_list))
    if s_f is None:
        gen_cost, disc_cost = _bce_costs(disc_fake_list, disc_real_list, [1.0 / n] * len(disc_fake_list))
    else:
        gen_cost, disc_cost = _bce_costs(disc_fake_list, disc_real_list, [1.0] * len(disc_fake_list))
        gen_cost = (_val(gen_cost) + s_f) / n if gen_cost is not None else None
        disc_cost = _val(disc_cost) / n if disc_cost is not None else None
    gen_cost = _plus(gen_cost, rec_penalty)
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=0.999)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=0.999)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def alice(disc_fake, disc_real, rec_penalty, gen_params, disc_params, lr=2e-4, beta1=0.5, s_f=None):
    """tflib/objs/gan_inference.py:162-195: the ali costs, generator side + s_f + rec_penalty."""
    gen_cost, disc_cost = _bce_costs([disc_fake], [disc_real], [1.0])
    gen_cost = _plus(gen_cost, s_f, rec_penalty)
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=0.999)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=0.999)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def vegan(disc_fake, disc_real, rec_penalty, gen_params, disc_params, lamb, lr=2e-4, beta1=.5, s_f=None):
    """tflib/objs/gan_inference.py:197-225: gen = lamb*(BCE(fake,1) + s_f) + rec_penalty; disc = lamb/2 * ali critic cost."""
    gen_cost = disc_cost = None
    if ONLY[0] != 'disc':
        gen_cost = F.BceSum.apply((1.0,), (1.0,), disc_fake)
        gen_cost = _val(_plus(gen_cost, s_f)) * float(lamb)
        gen_cost = _plus(gen_cost, rec_penalty)
    if ONLY[0] != 'gen':
        disc_cost = F.BceSum.apply((0.0, 1.0), (float(lamb) / 2, float(lamb) / 2), disc_fake, disc_real)
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=0.999)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=0.999)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def vegan_wgan_gp(disc_fake, disc_real, rec_penalty, gradient_penalty, gen_params, disc_params, lamb, lr=2e-4, beta1=.5):
    """tflib/objs/gan_inference.py:227-244."""
    gen_cost = disc_cost = None
    if ONLY[0] != 'disc':
        gen_cost = _plus(F.MeanSum.apply((-float(lamb), float(lamb)), disc_fake, disc_real), rec_penalty)
    if ONLY[0] != 'gen':
        if gradient_penalty is not None:
            disc_cost = F.MeanSum.apply((float(lamb), -float(lamb), 1.0), disc_fake, disc_real, gradient_penalty)
        else:
            disc_cost = F.MeanSum.apply((float(lamb), -float(lamb)), disc_fake, disc_real)
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=0.999)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=0.999)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def local_ep_dynamic(disc_fake_zz, disc_real_zz, disc_fake_xz, disc_real_xz, gen_params, disc_params, lr=2e-4, beta1=0.5, beta2=.999,
                     rec_penalty=None):
    """tflib/objs/gan_inference.py:246-305: the transition factors' BCE pairs summed and divided by (their number + 1) -- not by their
    number: the reference's own normalisation -- plus the observation factor's pair at weight 1 (+ rec_penalty on the generator side).
    All terms of a cost are ONE launch: the division is folded into the terms' weights."""
    n = len(disc_fake_zz)
    assert n == len(disc_real_zz)
    w = 1.0 / (n + 1) if n > 0 else 1.0
    gen_cost, disc_cost = _bce_costs(list(disc_fake_zz) + [disc_fake_xz], list(disc_real_zz) + [disc_real_xz], [w] * n + [1.0])
    gen_cost = _plus(gen_cost, rec_penalty)
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=beta2)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=beta2)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def weighted_local_epce(disc_fake_list, disc_real_list, ratio_list, gen_params, disc_params, lr=2e-4, beta1=0.5,
                        rec_penalty=None):
    ratio_list = np.asarray(ratio_list)
    assert len(disc_fake_list) == ratio_list.shape[0]
    gen_cost, disc_cost = _bce_costs(disc_fake_list, disc_real_list, list(ratio_list))
    gen_debug_list, disc_debug_list = [], []   # per-factor terms are debug-only in the reference (:321-343)
    if rec_penalty is not None and gen_cost is not None:
        gen_cost = _val(gen_cost) + rec_penalty
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=beta1, beta2=0.999)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=beta1, beta2=0.999)
    return (gen_cost, disc_cost, gen_debug_list, disc_debug_list,
            TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost))


def wali(disc_fake, disc_real, gen_params, disc_params, lr=5e-5):
    """tflib/objs/gan_inference.py:4-26: Wasserstein costs, two RMSProp optimizers, critic weights clipped to [-.01, .01].
    Returns the reference's 6-tuple; `clip_disc_weights` / `clip_ops` are no-op callables here because the clipping is part
    of the critic's update kernel (the reference runs it as a separate session.run right after the critic step)."""
    gen_cost = F.MeanSum.apply((-1.0, -1.0), disc_fake, disc_real) if ONLY[0] != 'disc' else None
    disc_cost = F.MeanSum.apply((1.0, -1.0), disc_fake, disc_real) if ONLY[0] != 'gen' else None
    gen_opt = get_optimizer('gen', gen_params, kind='rmsprop', lr=lr)
    disc_opt = get_optimizer('disc', disc_params, kind='rmsprop', lr=lr, clip=(-.01, .01))
    noop = lambda: None
    return gen_cost, disc_cost, noop, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost), [noop]


def wali_gp(disc_fake, disc_real, gradient_penalty, gen_params, disc_params, lr=1e-4):
    gen_cost = F.MeanSum.apply((-1.0, 1.0), disc_fake, disc_real) if ONLY[0] != 'disc' else None
    disc_cost = None
    if ONLY[0] != 'gen':
        if gradient_penalty is not None:     # (the penalty is a one-element term of weight 1 of the same cost launch: no addition launch)
            disc_cost = F.MeanSum.apply((1.0, -1.0, 1.0), disc_fake, disc_real, gradient_penalty)
        else:
            disc_cost = F.MeanSum.apply((1.0, -1.0), disc_fake, disc_real)
    gen_opt = get_optimizer('gen', gen_params, lr=lr, beta1=0.5, beta2=0.9)
    disc_opt = get_optimizer('disc', disc_params, lr=lr, beta1=0.5, beta2=0.9)
    return gen_cost, disc_cost, TrainOp(gen_opt, gen_cost), TrainOp(disc_opt, disc_cost)


def gradient_penalty(critic, real_x, fake_x, q_z, p_z, alpha, lam=10.):
    """The wali-gp construction of gan_inference_cifar10.py:353-364 (script-level code in the reference):
    interpolate, third critic pass, x-gradient only, lam*mean((||g||-1)^2).  create_graph=True keeps the
    backward kernels on the tape so minimize() differentiates THROUGH g (double backward)."""
    import torch
    x_hat = F.RowLerp.apply(real_x, fake_x, alpha)
    z_hat = F.RowLerp.apply(q_z, p_z, alpha)
    if not x_hat.requires_grad:
        x_hat.requires_grad_(True)
    d_hat = critic(x_hat, z_hat)
    ones = F.cached_const(1.0, d_hat.shape, d_hat.device)      # (persistent: no fill launch per critic step)
    with F.data_grad_only():             # only d/d x_hat is asked for: the layers skip their parameter gradients
        (g,) = torch.autograd.grad(d_hat, [x_hat], grad_outputs=ones, create_graph=True)
    return F.GradPenalty.apply(g.reshape(g.shape[0], -1), float(lam))
