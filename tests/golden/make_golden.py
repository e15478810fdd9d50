#!/usr/bin/env python
"""Generate the committed golden fixtures from the float64 oracle.

BUILD-GENERATED, NOT REFERENCE-DERIVED: the reference (Python2 + TF1) cannot run here and ships no vectors
(SURVEY.md 8c), so these pin the oracle against regressions and give the -m gpu tests a fixed target.
    python tests/golden/make_golden.py          (rewrites tests/golden/*.npz; deterministic)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ops as O, nets as N, step as S, tape as tp  # noqa: E402

TRAJ = {   # name -> (dataset, B, K, mode, dim, dim_latent, iterations)
    'traj_cifar_ali': ('cifar10', 8, 0, 'ali', 8, 16, 3),
    'traj_cifar_gmgan': ('cifar10', 8, 5, 'ali', 8, 16, 3),
    'traj_cifar_wali_gp': ('cifar10', 6, 0, 'wali-gp', 8, 16, 2),
    'traj_mnist_gmgan': ('mnist', 6, 4, 'ali', 8, 16, 2),
    'traj_face_gmgan': ('face', 4, 6, 'ali', 4, 16, 2),
    'traj_svhn_gmgan': ('svhn', 6, 5, 'ali', 8, 16, 2),             # no BatchNorm
    'traj_cifar_wali': ('cifar10', 6, 0, 'wali', 8, 16, 2),         # RMSProp + weight clipping, 5 critic steps
    'traj_cifar_alice': ('cifar10', 6, 0, 'alice', 8, 16, 2),       # both reconstruction terms
    'traj_cifar_vegan': ('cifar10', 4, 0, 'vegan', 8, 16, 2),       # latent critic with BatchNorm + noise layers
    'traj_cifar_vegan_jsd': ('cifar10', 6, 0, 'vegan-jsd', 8, 8, 4),   # critic-free: stochastic encoder + aggregated-posterior JSD
    'traj_cifar_vegan_kl': ('cifar10', 6, 0, 'vegan-kl', 8, 8, 3),
    'traj_cifar_vegan_mmd': ('cifar10', 6, 0, 'vegan-mmd', 8, 16, 3),
}
Z_SAMPLES = 12      # Monte-Carlo samples of the aggregated-divergence trajectories (the scripts' 100 would only make the fixture larger)


def cfg_for(name):
    dataset, B, K, mode, dim, dl, iters = TRAJ[name]
    agg = mode in S.AGG_MODES
    return N.Cfg(dataset, batch_size=B, n_coms=K, dim=dim, dim_latent=dl, latent_critic=mode in ('vegan', 'vegan-wgan-gp'),
                 learn_std=agg, z_samples=Z_SAMPLES)


def perturbed_params(cfg, seed=0):
    P0 = N.init_params(cfg, seed)
    rng = np.random.default_rng(7)
    for k in P0:
        if P0[k].ndim <= 2 and ('Biases' in k or k.endswith('.b') or 'offset' in k):
            P0[k] = (0.1 * rng.standard_normal(P0[k].shape)).astype(np.float32)
        if k.endswith('.scale'):
            P0[k] = (1 + 0.1 * rng.standard_normal(P0[k].shape)).astype(np.float32)
    return P0


def make_ops():
    rng = np.random.default_rng(2026)
    out = {}
    for tag, (n, ci, h, co) in {'a': (2, 3, 8, 4), 'b': (2, 4, 7, 5), 'c': (1, 2, 28, 3)}.items():
        x = rng.standard_normal((n, ci, h, h)).astype(np.float32)
        w = rng.standard_normal((5, 5, ci, co)).astype(np.float32)
        ho = O.conv_geometry(h, 5, 2)[0]
        gy = rng.standard_normal((n, co, ho, ho)).astype(np.float32)
        out.update({'conv_%s_x' % tag: x, 'conv_%s_w' % tag: w, 'conv_%s_gy' % tag: gy,
                    'conv_%s_y' % tag: O.conv2d(x.astype(np.float64), w.astype(np.float64), 2),
                    'conv_%s_gx' % tag: O.conv2d_bwd_data(gy.astype(np.float64), w.astype(np.float64), (h, h), 2),
                    'conv_%s_gw' % tag: O.conv2d_bwd_filter(x.astype(np.float64), gy.astype(np.float64), 5, 2)})
    x = rng.standard_normal((3, 6, 4, 4)).astype(np.float32)
    w = rng.standard_normal((5, 5, 2, 6)).astype(np.float32)            # Deconv2D layout [k,k,out,in]
    out.update(deconv_x=x, deconv_w=w, deconv_y=O.deconv2d(x.astype(np.float64), w.astype(np.float64)))
    x = (rng.standard_normal((5, 3, 4, 4)) * 2 + 1).astype(np.float32)
    sc, of = rng.standard_normal(3).astype(np.float32), rng.standard_normal(3).astype(np.float32)
    out.update(bn_x=x, bn_scale=sc, bn_offset=of,
               bn_y=O.batchnorm_train(x.astype(np.float64), sc.astype(np.float64), of.astype(np.float64), (0, 2, 3)))
    lg = (rng.standard_normal(16) * 3).astype(np.float32)
    out.update(bce_x=lg, bce_1=O.bce_with_logits(lg.astype(np.float64), 1.0), bce_0=O.bce_with_logits(lg.astype(np.float64), 0.0))
    th, g = rng.standard_normal(9), rng.standard_normal((3, 9))
    m, v, seq = np.zeros(9), np.zeros(9), []
    t_ = th.copy()
    for t in range(1, 4):
        t_, m, v = O.adam_update(t_, g[t - 1], m, v, t, 2e-4, .5, .999)
        seq.append(t_.copy())
    out.update(adam_theta0=th, adam_g=g, adam_theta=np.stack(seq))
    np.savez_compressed(os.path.join(HERE, 'ops_small.npz'), **out)


def make_traj(name):
    dataset, B, K, mode, dim, dl, iters = TRAJ[name]
    cfg = cfg_for(name)
    P0 = perturbed_params(cfg)
    tr = S.Trainer(cfg, P0, mode, np.float64)
    n_feeds = iters * (1 + tr.critic_iters)
    feeds = [S.make_feed(cfg, np.random.default_rng(100 + i), mode) for i in range(n_feeds)]
    it_f = iter(feeds)
    costs = []
    for it in range(iters):
        r = tr.iteration(it, it_f)
        costs.append([r.get('gen_cost', np.nan), r.get('disc_cost', np.nan)])
    out = {'costs': np.asarray(costs)}
    # initial weights are NOT stored: perturbed_params(cfg) regenerates them (numpy legacy RandomState is
    # stream-stable); final weights are stored as per-tensor digests (sum, abs-sum, first 8 values)
    for k, v in tr.P.items():
        f = v.reshape(-1)
        out['p1/' + k] = np.concatenate([[f.sum(), np.abs(f).sum()], f[:8]])
    for i, f in enumerate(feeds):
        for k, v in f.items():
            out['feed%d/%s' % (i, k)] = v.astype(np.uint8) if k == 'real_x_int' else v
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)


if __name__ == '__main__':
    only = sys.argv[1:]                       # optional: just these trajectories (existing files stay byte-identical)
    if not only:
        make_ops()
    for n in (only or TRAJ):
        make_traj(n)
    print('wrote', sorted(f for f in os.listdir(HERE) if f.endswith('.npz')))
