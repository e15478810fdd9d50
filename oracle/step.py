"""The training iteration of the reference's driver scripts, restated on the oracle.

TEST INFRASTRUCTURE (see oracle/__init__.py: composition pinned by tests/golden/reference_trace.json, TF primitives unpinned).

Wiring follows gan_inference_cifar10.py:261-366 (MODE 'ali' / 'wali-gp') and
gmgan_inference_cifar10.py:341-398 (MODE 'local_ep'); the loop follows
gmgan_inference_cifar10.py:480-494: iteration 0 runs the critic step only, every
session.run draws a fresh minibatch and fresh noise, gen and disc have independent Adam state.
All randomness is injected through `feed` (the reference's TF RNG is unseeded).
"""
import numpy as np
from . import tape as tp
from . import nets as N
from . import objs as J


AGG_MODES = ('vegan-kl', 'vegan-ikl', 'vegan-jsd')


def make_feed(cfg, rng, mode='ali'):
    """Synthetic inputs for one session.run."""
    f = {}
    if cfg.dataset == 'mnist':
        f['real_x'] = rng.random((cfg.B, cfg.output_dim), dtype=np.float32)
    else:
        f['real_x_int'] = rng.integers(0, 256, size=(cfg.B, cfg.output_dim)).astype(np.int32)
    if cfg.dataset == 'face':
        f['dequant_u'] = (rng.random((cfg.B, cfg.output_dim), dtype=np.float32) / 128.0).astype(np.float32)
    f['p_z_noise'] = rng.standard_normal((cfg.B, cfg.dim_latent), dtype=np.float32)
    if cfg.K:
        f['k_idx'] = rng.integers(0, cfg.K, size=(cfg.B,)).astype(np.int64)
        f['gumbel_u'] = rng.random((cfg.B, cfg.K), dtype=np.float32)
    if mode in ('wali-gp', 'vegan-wgan-gp'):
        f['alpha'] = rng.random((cfg.B, 1), dtype=np.float32)
    if mode in ('vegan', 'vegan-wgan-gp'):          # the latent critic's Gaussian noise layers, one set per critic call
        for tag in ('f', 'r') + (('h',) if mode == 'vegan-wgan-gp' else ()):
            for i, w in enumerate((cfg.dim_latent, 1024, 512, 256)):
                f['dn_%s%d' % (tag, i)] = rng.standard_normal((cfg.B, w), dtype=np.float32)
    if mode in AGG_MODES:                           # the stochastic encoder's eps and the Monte-Carlo draws of tflib/objs/kl_aggregated.py
        Z = cfg.z_samples
        f['q_eps'] = rng.standard_normal((cfg.B, cfg.dim_latent), dtype=np.float32)
        k = np.zeros((Z, cfg.B), np.float32)
        k[np.arange(Z), rng.integers(0, cfg.B, size=Z)] = 1
        f['kl_k'] = k
        f['kl_eps'] = rng.standard_normal((Z, cfg.dim_latent), dtype=np.float32)
        f['kl_zp'] = rng.standard_normal((Z, cfg.dim_latent), dtype=np.float32)
    return f


def real_x_from_feed(cfg, feed, dtype):
    if cfg.dataset == 'mnist':                      # gmgan_inference_mnist.py:335
        return np.asarray(feed['real_x'], dtype=dtype)
    xi = feed['real_x_int'].astype(dtype)
    if cfg.dataset == 'face':                       # gmgan_inference_face.py:242-243
        return (dtype(2) * ((xi / dtype(256.)) - dtype(.5))) + feed['dequant_u'].astype(dtype)
    return dtype(2) * ((xi / dtype(255.)) - dtype(.5))   # gmgan_inference_cifar10.py:342


def forward(cfg, P, feed, mode='ali'):
    """P: name -> tape.T.  Returns dict of taped tensors incl. gen_cost / disc_cost."""
    dt = next(iter(P.values())).v.dtype.type
    real_x = tp.T(real_x_from_feed(cfg, feed, dt))
    if mode in AGG_MODES:                           # gan_inference_cifar10.py:263-270,331-341: no critic, stochastic encoder
        q_z, q_mean, q_std = N.Extractor(cfg, P, real_x, eps=tp.T(feed['q_eps'].astype(dt)))
        rec = J.distance(real_x, N.Generator(cfg, P, q_z), 'l2')
        div = J.aggregated_divergence(mode[6:], q_mean, q_std, tp.T(feed['kl_k'].astype(dt)), tp.T(feed['kl_eps'].astype(dt)),
                                      tp.T(feed['kl_zp'].astype(dt)), cfg.B)
        return {'real_x': real_x, 'q_z': q_z, 'q_z_mean': q_mean, 'q_z_std': q_std, 'rec_penalty': rec, 'divergence': div,
                'gen_cost': tp.add(tp.scale(div, 1.0), rec), 'disc_cost': None}
    q_z = N.Extractor(cfg, P, real_x)
    noise = tp.T(feed['p_z_noise'].astype(dt))
    out = {'real_x': real_x, 'q_z': q_z}
    if cfg.K:
        onehot = np.zeros((cfg.B, cfg.K), dtype=dt)
        onehot[np.arange(cfg.B), feed['k_idx']] = 1
        onehot = tp.T(onehot)
        _, q_k = N.HyperExtractor(cfg, P, q_z, feed['gumbel_u'])
        p_z = N.HyperGenerator(cfg, P, onehot, noise)
        out['q_k'] = q_k
    else:
        p_z = noise
    fake_x = N.Generator(cfg, P, p_z)
    out.update(p_z=p_z, fake_x=fake_x)

    def critic(x, z):
        return N.Discriminator(cfg, P, x, z)

    if mode == 'vegan-mmd':                         # gan_inference_cifar10.py:327-329: no critic
        rec = J.distance(real_x, N.Generator(cfg, P, q_z), 'l2')
        gen_cost = tp.add(tp.scale(J.mix_rbf_mmd2(q_z, p_z), 1.0), rec)
        out.update(gen_cost=gen_cost, disc_cost=None, rec_penalty=rec)
        return out
    if mode in ('vegan', 'vegan-wgan-gp'):          # gan_inference_cifar10.py:272-275,305-322: the critic discriminates codes
        noise = lambda tag: [feed['dn_%s%d' % (tag, i)] for i in range(4)]
        d_real = N.LatentDiscriminator(cfg, P, p_z, noise('r'))
        d_fake = N.LatentDiscriminator(cfg, P, q_z, noise('f'))
        rec = J.distance(real_x, N.Generator(cfg, P, q_z), 'l2')
        if mode == 'vegan':
            gen_cost, disc_cost = J.vegan_costs(d_fake, d_real, rec, 1.0)
        else:
            gp = J.latent_gradient_penalty(lambda zz: N.LatentDiscriminator(cfg, P, zz, noise('h')), q_z, p_z, feed['alpha'])
            gen_cost, disc_cost = J.vegan_wgan_gp_costs(d_fake, d_real, rec, gp, 1.0)
            out['gradient_penalty'] = gp
        out.update(disc_fake=d_fake, disc_real=d_real, gen_cost=gen_cost, disc_cost=disc_cost, rec_penalty=rec)
        return out

    def rec_penalty():
        """gan_inference_cifar10.py:264-304 / gmgan_inference_cifar10.py:348,399-403 (DISTANCE_X = 'l2')"""
        r = None
        if mode != 'alice-x':
            r = J.distance(real_x, N.Generator(cfg, P, q_z), 'l2')
        if mode in ('alice', 'alice-x'):
            rz = J.distance(p_z, N.Extractor(cfg, P, fake_x), 'l2')
            r = rz if r is None else tp.add(r, rz)
        return r

    if cfg.K:
        d_fake = [N.HyperDiscriminator(cfg, P, p_z, onehot), critic(fake_x, p_z)]
        d_real = [N.HyperDiscriminator(cfg, P, q_z, q_k), critic(real_x, q_z)]
        if mode == 'local_epce':
            gen_cost, disc_cost = J.local_epce_costs(d_fake, d_real, rec_penalty())
        else:
            gen_cost, disc_cost = J.local_ep_costs(d_fake, d_real)
    else:
        d_fake, d_real = critic(fake_x, p_z), critic(real_x, q_z)
        if mode == 'ali':
            gen_cost, disc_cost = J.ali_costs(d_fake, d_real)
        elif mode in ('alice', 'alice-z', 'alice-x'):
            gen_cost, disc_cost = J.alice_costs(d_fake, d_real, rec_penalty())
        elif mode == 'wali':
            gen_cost, disc_cost = J.wali_costs(d_fake, d_real)
        elif mode == 'wali-gp':
            gp = J.gradient_penalty(critic, real_x, fake_x, q_z, p_z, feed['alpha'])
            gen_cost, disc_cost = J.wali_gp_costs(d_fake, d_real, gp)
            out['gradient_penalty'] = gp
        else:
            raise ValueError(mode)
    out.update(disc_fake=d_fake, disc_real=d_real, gen_cost=gen_cost, disc_cost=disc_cost)
    return out


class Trainer(object):
    """Holds parameters + the two Adam states; gen_step / disc_step = one session.run each."""

    def __init__(self, cfg, params, mode='ali', dtype=np.float64):
        self.cfg, self.mode, self.dtype = cfg, mode, dtype
        self.P = {k: np.asarray(v, dtype=dtype).copy() for k, v in params.items()}
        gen_names = N.trainable(N.params_with_name(self.P, 'Generator') + N.params_with_name(self.P, 'Extractor'))
        disc_names = N.trainable(N.params_with_name(self.P, 'Discriminator'))
        if mode == 'wali-gp':                       # tflib/objs/gan_inference.py:34-43
            hp = dict(lr=1e-4, beta1=0.5, beta2=0.9)
        else:                                       # LR/BETA1 of the scripts, beta2 default
            hp = dict(lr=2e-4, beta1=0.5, beta2=0.999)
        if mode == 'wali':                          # tflib/objs/gan_inference.py:8-24: RMSProp 5e-5, critic weights clipped
            self.gen_opt = J.RMSProp(gen_names)
            self.disc_opt = J.RMSProp(disc_names, clip=(-.01, .01))
        else:
            self.gen_opt = J.Adam(gen_names, **hp)
            self.disc_opt = J.Adam(disc_names, **hp)
        self.critic_iters = 5 if mode in ('wali', 'wali-gp', 'vegan', 'vegan-wgan-gp') else (0 if mode == 'vegan-mmd' or mode in AGG_MODES else 1)   # :52-59

    def _run(self, feed, which):
        Pt = {k: tp.T(v) for k, v in self.P.items()}
        out = forward(self.cfg, Pt, feed, self.mode)
        opt = self.gen_opt if which == 'gen' else self.disc_opt
        cost = out[which + '_cost']
        gs = tp.grad(cost, [Pt[n] for n in opt.names])
        grads = {n: (g.v if g is not None else None) for n, g in zip(opt.names, gs)}
        return float(cost.v), grads, out

    def gen_step(self, feed):
        cost, grads, _ = self._run(feed, 'gen')
        self.gen_opt.apply(self.P, grads)
        return cost

    def disc_step(self, feed):
        cost, grads, _ = self._run(feed, 'disc')
        self.disc_opt.apply(self.P, grads)
        return cost

    def iteration(self, it, feeds):
        """feeds: iterator of feed dicts; consumes 1 (if it>0) + critic_iters of them.
        gmgan_inference_cifar10.py:480-494."""
        res = {}
        if it > 0:
            res['gen_cost'] = self.gen_step(next(feeds))
        for _ in range(self.critic_iters):
            res['disc_cost'] = self.disc_step(next(feeds))
        return res
