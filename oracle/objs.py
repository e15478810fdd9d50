"""Objectives of tflib/objs/gan_inference.py restated on the oracle tape.

TEST INFRASTRUCTURE (see oracle/__init__.py: composition pinned by tests/golden/reference_trace.json, TF primitives unpinned).
"""
import numpy as np
from . import tape as tp


def _bce_mean(logits, label):
    return tp.reduce_mean(tp.bce_with_logits(logits, label))


def local_ep_costs(disc_fake_list, disc_real_list):
    """tflib/objs/gan_inference.py:81-106 -> (gen_cost, disc_cost)."""
    gen, disc = 0.0, 0.0
    for f, r in zip(disc_fake_list, disc_real_list):
        gen = tp.add(tp.add(_bce_mean(f, 1.0), _bce_mean(r, 0.0)), gen)
        disc = tp.add(tp.add(_bce_mean(f, 0.0), _bce_mean(r, 1.0)), disc)
    n = float(len(disc_fake_list))
    return tp.scale(gen, 1.0 / n), tp.scale(disc, 1.0 / n)


def ali_costs(disc_fake, disc_real):
    """tflib/objs/gan_inference.py:47-66 (== local_ep with one factor)."""
    gen = tp.add(_bce_mean(disc_fake, 1.0), _bce_mean(disc_real, 0.0))
    disc = tp.add(_bce_mean(disc_fake, 0.0), _bce_mean(disc_real, 1.0))
    return gen, disc


def weighted_local_epce_costs(disc_fake_list, disc_real_list, ratio_list):
    """tflib/objs/gan_inference.py:307-345 (rec_penalty=None)."""
    gen, disc = 0.0, 0.0
    for f, r, ratio in zip(disc_fake_list, disc_real_list, ratio_list):
        ratio = float(ratio)
        gen = tp.add(tp.scale(tp.add(_bce_mean(f, 1.0), _bce_mean(r, 0.0)), ratio), gen)
        disc = tp.add(tp.scale(tp.add(_bce_mean(f, 0.0), _bce_mean(r, 1.0)), ratio), disc)
    return gen, disc


def local_ep_dynamic_costs(disc_fake_zz, disc_real_zz, disc_fake_xz, disc_real_xz, rec_penalty=None):
    """tflib/objs/gan_inference.py:246-296: literally -- accumulate the transition factors' pairs, divide by (n + 1) when there are
    any, add the observation factor's pair, add rec_penalty to the generator side."""
    gen, disc = 0.0, 0.0
    for f, r in zip(disc_fake_zz, disc_real_zz):
        gen = tp.add(tp.add(_bce_mean(f, 1.0), _bce_mean(r, 0.0)), gen)
        disc = tp.add(tp.add(_bce_mean(f, 0.0), _bce_mean(r, 1.0)), disc)
    if len(disc_fake_zz) > 0:
        gen = tp.scale(gen, 1.0 / (len(disc_fake_zz) + 1))
        disc = tp.scale(disc, 1.0 / (len(disc_fake_zz) + 1))
    gen = tp.add(tp.add(_bce_mean(disc_fake_xz, 1.0), _bce_mean(disc_real_xz, 0.0)), gen)
    disc = tp.add(tp.add(_bce_mean(disc_fake_xz, 0.0), _bce_mean(disc_real_xz, 1.0)), disc)
    if rec_penalty is not None:
        gen = tp.add(gen, rec_penalty)
    return gen, disc


def distance(x, y, d_type):
    """tflib/utils/distance.py:3-17"""
    d = tp.add(x, tp.neg(y))
    return tp.reduce_mean(tp.square(d)) if d_type == 'l2' else tp.reduce_mean(tp.absolute(d))


def alice_costs(disc_fake, disc_real, rec_penalty):
    """tflib/objs/gan_inference.py:162-184"""
    g, d = ali_costs(disc_fake, disc_real)
    return tp.add(g, rec_penalty), d


def local_epce_costs(disc_fake_list, disc_real_list, rec_penalty):
    """tflib/objs/gan_inference.py:121-149"""
    g, d = local_ep_costs(disc_fake_list, disc_real_list)
    return tp.add(g, rec_penalty), d


def vegan_costs(disc_fake, disc_real, rec_penalty, lamb):
    """tflib/objs/gan_inference.py:194-212."""
    gen = tp.add(tp.scale(_bce_mean(disc_fake, 1.0), lamb), rec_penalty)
    disc = tp.scale(tp.add(_bce_mean(disc_fake, 0.0), _bce_mean(disc_real, 1.0)), lamb / 2)
    return gen, disc


def vegan_wgan_gp_costs(disc_fake, disc_real, rec_penalty, gradient_penalty, lamb):
    """tflib/objs/gan_inference.py:225-233."""
    gen = tp.add(tp.scale(tp.add(tp.neg(tp.reduce_mean(disc_fake)), tp.reduce_mean(disc_real)), lamb), rec_penalty)
    disc = tp.add(tp.scale(tp.add(tp.reduce_mean(disc_fake), tp.neg(tp.reduce_mean(disc_real))), lamb), gradient_penalty)
    return gen, disc


def latent_gradient_penalty(critic, q_z, p_z, alpha, lam=10.0):
    """gan_inference_cifar10.py:309-317: interpolates = p_z + alpha*(q_z - p_z); lam*mean((||dD/dz_hat|| - 1)^2)."""
    a = tp.T(np.asarray(alpha, dtype=q_z.v.dtype).reshape(-1, 1))
    z_hat = tp.add(p_z, tp.mul(a, tp.add(q_z, tp.neg(p_z))))
    g = tp.grad(tp.reduce_sum(critic(z_hat)), [z_hat])[0]
    slopes = tp.sqrt(tp.reduce_sum(tp.square(g), (1,)))
    return tp.scale(tp.reduce_mean(tp.square(tp.add(slopes, -1.0))), lam)


def mix_rbf_mmd2(X, Y, sigmas=(2., 5., 10., 20., 40., 80.)):
    """tflib/objs/mmd.py:20-67 (biased=True, wts = 1): Gram matrices from the matmul form -2*X.Y^T + |x|^2 + |y|^2."""
    def sq(a):
        return tp.reduce_sum(tp.square(a), (1,), keepdims=True)
    def gram(a, b):
        return tp.add(tp.add(tp.scale(tp.matmul(a, tp.transpose(b, (1, 0))), -2.0), sq(a)), tp.transpose(sq(b), (1, 0)))
    dxx, dxy, dyy = gram(X, X), gram(X, Y), gram(Y, Y)
    kxx = kxy = kyy = None
    for sg in sigmas:
        gm = 1.0 / (2.0 * sg * sg)
        exx, exy, eyy = (tp.exp(tp.scale(d, -gm)) for d in (dxx, dxy, dyy))
        kxx = exx if kxx is None else tp.add(kxx, exx)
        kxy = exy if kxy is None else tp.add(kxy, exy)
        kyy = eyy if kyy is None else tp.add(kyy, eyy)
    m, n = X.v.shape[0], Y.v.shape[0]
    return tp.add(tp.add(tp.scale(tp.reduce_sum(kxx), 1.0 / (m * m)), tp.scale(tp.reduce_sum(kyy), 1.0 / (n * n))),
                  tp.scale(tp.reduce_sum(kxy), -2.0 / (m * n)))


# ---- tflib/objs/kl_aggregated.py: divergences between the aggregated posterior (a mixture of the minibatch's diagonal Gaussians) and
# the prior, estimated on Z_SAMPLES Monte-Carlo samples (MODE vegan-kl / vegan-ikl / vegan-jsd) -------------------------------------
def mixture_gaussian(k_onehot, mu, std, eps):
    """:6-16: z = k @ mu + (k @ std) * eps, k one-hot rows drawn uniformly over the components"""
    return tp.add(tp.matmul(k_onehot, mu), tp.mul(tp.matmul(k_onehot, std), eps))


def log_likelihood_diagonal_gaussian(x, mu, std):
    """:18-20"""
    r = tp.mul(tp.add(x, tp.neg(mu)), tp.power(std, -1.0))
    res = tp.scale(tp.add(tp.add(tp.square(r), tp.scale(tp.log(std), 2.0)), tp.T(np.asarray(np.log(2 * np.pi), dtype=x.v.dtype))), -0.5)
    return tp.reduce_sum(res, (res.v.ndim - 1,))


def _log_mean_exp_rows(res_mat):
    """:26-29 (the subtracted row maximum is an additive constant of the total derivative)"""
    mx = tp.T(res_mat.v.max(axis=1, keepdims=True))
    return tp.add(tp.log(tp.reduce_mean(tp.exp(tp.add(res_mat, tp.neg(mx))), (1,))), tp.reshape(mx, (-1,)))


def log_likelihood_mixture_gaussian(x, mu, std):
    """:22-29: x [nz, d] under the equal-weight mixture of the nx components -> [nz]"""
    nz, d = x.v.shape
    nx = mu.v.shape[0]
    return _log_mean_exp_rows(log_likelihood_diagonal_gaussian(tp.reshape(x, (nz, 1, d)), tp.reshape(mu, (1, nx, d)), tp.reshape(std, (1, nx, d))))


def log_likelihood_mixture_mixture_gaussian(x, mu_q, std_q, mu_p, std_p, n_coms):
    """:31-44: the nx posterior components and n_coms copies of the prior term, equally weighted"""
    nz, d = x.v.shape
    nx = mu_q.v.shape[0]
    r1 = log_likelihood_diagonal_gaussian(tp.reshape(x, (nz, 1, d)), tp.reshape(mu_q, (1, nx, d)), tp.reshape(std_q, (1, nx, d)))
    r2 = log_likelihood_diagonal_gaussian(x, mu_p, std_p)
    r2 = tp.broadcast_to(tp.reshape(r2, (nz, 1)), (nz, n_coms))
    return _log_mean_exp_rows(tp.concat([r1, r2], axis=1))


def aggregated_divergence(kind, q_mean, q_std, k_onehot, eps_q, z_p, n_coms):
    """kl / ikl / jsd_q_aggregated_p_diagonal_gaussian (:46-74) with p = N(0, I) (gan_inference_cifar10.py:269-270); the script's
    random draws (component indices, eps, prior samples) are inputs"""
    dt = q_mean.v.dtype
    nz, d = z_p.v.shape
    p_mean, p_std = tp.T(np.zeros((nz, d), dt)), tp.T(np.ones((nz, d), dt))
    if kind == 'kl':
        z = mixture_gaussian(k_onehot, q_mean, q_std, eps_q)
        return tp.reduce_mean(tp.add(log_likelihood_mixture_gaussian(z, q_mean, q_std), tp.neg(log_likelihood_diagonal_gaussian(z, p_mean, p_std))))
    if kind == 'ikl':
        return tp.reduce_mean(tp.add(log_likelihood_diagonal_gaussian(z_p, p_mean, p_std), tp.neg(log_likelihood_mixture_gaussian(z_p, q_mean, q_std))))
    assert kind == 'jsd'
    z1 = mixture_gaussian(k_onehot, q_mean, q_std, eps_q)
    log_q = log_likelihood_mixture_gaussian(z1, q_mean, q_std)
    log_m1 = log_likelihood_mixture_mixture_gaussian(z1, q_mean, q_std, p_mean, p_std, n_coms)
    log_p = log_likelihood_diagonal_gaussian(z_p, p_mean, p_std)
    log_m2 = log_likelihood_mixture_mixture_gaussian(z_p, q_mean, q_std, p_mean, p_std, n_coms)
    return tp.reduce_mean(tp.scale(tp.add(tp.add(log_q, tp.neg(log_m1)), tp.add(log_p, tp.neg(log_m2))), 0.5))


def wali_costs(disc_fake, disc_real):
    """tflib/objs/gan_inference.py:5-6 (the generator cost really is -mean(fake) - mean(real) there)."""
    gen = tp.add(tp.neg(tp.reduce_mean(disc_fake)), tp.neg(tp.reduce_mean(disc_real)))
    disc = tp.add(tp.reduce_mean(disc_fake), tp.neg(tp.reduce_mean(disc_real)))
    return gen, disc


def wali_gp_costs(disc_fake, disc_real, gradient_penalty):
    """tflib/objs/gan_inference.py:28-32."""
    gen = tp.add(tp.neg(tp.reduce_mean(disc_fake)), tp.reduce_mean(disc_real))
    disc = tp.add(tp.add(tp.reduce_mean(disc_fake), tp.neg(tp.reduce_mean(disc_real))), gradient_penalty)
    return gen, disc


def gradient_penalty(critic, real_x, fake_x, q_z, p_z, alpha, lam=10.0):
    """gan_inference_cifar10.py:353-364.  critic(x, z) -> logits [B].  Only the x-gradient
    enters the penalty ([0] selects `interpolates`); no epsilon inside the sqrt."""
    a = tp.T(np.asarray(alpha, dtype=real_x.v.dtype).reshape(-1, 1))
    x_hat = tp.add(real_x, tp.mul(a, tp.add(fake_x, tp.neg(real_x))))
    z_hat = tp.add(q_z, tp.mul(a, tp.add(p_z, tp.neg(q_z))))
    d_hat = critic(x_hat, z_hat)
    g = tp.grad(tp.reduce_sum(d_hat), [x_hat])[0]
    slopes = tp.sqrt(tp.reduce_sum(tp.square(g), (1,)))
    return tp.scale(tp.reduce_mean(tp.square(tp.add(slopes, -1.0))), lam)


class RMSProp(object):
    """tf.train.RMSPropOptimizer(learning_rate) with TF's defaults (decay .9, momentum 0, epsilon 1e-10, mean-square slot
    initialised to ONE) for one var_list (tflib/objs/gan_inference.py:8-13); clip=(lo, hi) applies the weight clipping the
    scripts run right after every critic step (gan_inference.py:15-24, gan_inference_cifar10.py `session.run(clip_disc_weights)`)."""

    def __init__(self, names, lr=5e-5, decay=0.9, eps=1e-10, clip=None):
        self.names, self.lr, self.decay, self.eps, self.clip = list(names), lr, decay, eps, clip
        self.ms = {}

    def apply(self, P, grads):
        for n in self.names:
            g = grads.get(n)
            if g is None:
                continue
            g = g.astype(P[n].dtype)
            ms = self.decay * self.ms.get(n, np.ones_like(P[n])) + (1 - self.decay) * g * g
            self.ms[n] = ms
            P[n] = P[n] - self.lr * g / np.sqrt(ms + self.eps)
            if self.clip is not None:
                P[n] = np.clip(P[n], self.clip[0], self.clip[1])


class Adam(object):
    """tf.train.AdamOptimizer state for one var_list."""

    def __init__(self, names, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8):
        self.names, self.lr, self.b1, self.b2, self.eps = list(names), lr, beta1, beta2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def apply(self, P, grads):
        """P: name -> ndarray (updated in place: entries are replaced); grads: name -> ndarray|None."""
        from .ops import adam_update
        self.t += 1
        for n in self.names:
            g = grads.get(n)
            if g is None:                       # TF minimize drops (None, var) pairs
                continue
            m = self.m.get(n, np.zeros_like(P[n]))
            v = self.v.get(n, np.zeros_like(P[n]))
            P[n], self.m[n], self.v[n] = adam_update(P[n], g.astype(P[n].dtype), m, v, self.t,
                                                     self.lr, self.b1, self.b2, self.eps)
