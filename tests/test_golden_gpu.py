"""-m gpu: the HIP path against the COMMITTED golden fixtures (tests/golden/*.npz, build-generated from the
float64 oracle by tests/golden/make_golden.py) -- nothing from the oracle package is executed here except the
deterministic initial-weight generator."""
import os
import sys

import numpy as np
import pytest

from _golden import load, traj_feeds, digest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)


def _rel(a, ref):
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30))


def test_ops_vs_goldens(gpu):
    import torch
    from graphical_gan_amd import functional as F
    z = load('ops_small')
    for tag, (n, ci, h, co) in {'a': (2, 3, 8, 4), 'b': (2, 4, 7, 5), 'c': (1, 2, 28, 3)}.items():
        geom = F.conv_geom(n, ci, h, h, co, 5, 2)
        x, w, gy = (_t(z['conv_%s_%s' % (tag, k)], gpu) for k in ('x', 'w', 'gy'))
        assert _rel(F.ConvFwd.apply(x, w, None, geom, 0, 0.0).cpu().numpy(), z['conv_%s_y' % tag]) < 2e-5
        assert _rel(F.ConvDgrad.apply(gy, w, None, geom, 0, 0.0).cpu().numpy(), z['conv_%s_gx' % tag]) < 2e-5
        assert _rel(F.ConvWgrad.apply(x, gy, geom).cpu().numpy(), z['conv_%s_gw' % tag]) < 2e-5
    geom = F.conv_geom(3, 2, 8, 8, 6, 5, 2)
    y = F.ConvDgrad.apply(_t(z['deconv_x'], gpu), _t(z['deconv_w'], gpu), None, geom, 0, 0.0).cpu().numpy()
    assert _rel(y, z['deconv_y']) < 2e-5
    y = F.BatchNormTrain.apply(_t(z['bn_x'], gpu), _t(z['bn_scale'], gpu), _t(z['bn_offset'], gpu), 1e-5, 0, 0.0)
    assert _rel(y.cpu().numpy(), z['bn_y']) < 1e-5
    for lab in (1.0, 0.0):
        l = F.BceSum.apply((lab,), (1.0,), _t(z['bce_x'], gpu))
        assert abs(float(l) - z['bce_%d' % lab].mean()) < 1e-6
    th = _t(z['adam_theta0'], gpu)
    m, v = torch.zeros_like(th), torch.zeros_like(th)
    step = torch.zeros(1, dtype=torch.int32, device=gpu)
    for t in range(3):
        F.adam_step_(th, _t(z['adam_g'][t], gpu), m, v, step, 2e-4, .5, .999)
        assert np.abs(th.cpu().numpy() - z['adam_theta'][t]).max() < 5e-7


@pytest.mark.parametrize('name', ['traj_cifar_ali', 'traj_cifar_gmgan', 'traj_cifar_wali_gp', 'traj_mnist_gmgan',
                                  'traj_face_gmgan', 'traj_svhn_gmgan', 'traj_cifar_wali', 'traj_cifar_alice', 'traj_cifar_vegan',
                                  'traj_cifar_vegan_jsd', 'traj_cifar_vegan_kl', 'traj_cifar_vegan_mmd'])
def test_trajectory_vs_goldens(gpu, name):
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import make_golden as MG
    from oracle import nets as N
    from graphical_gan_amd import tflib as lib, optim
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models import Config
    dataset, B, K, mode, dim, dl, iters = MG.TRAJ[name]
    z = load(name)
    ocfg = MG.cfg_for(name)
    optim.reset_optimizers(); lib.delete_all_params()
    pmode = 'local_ep' if (K and mode == 'ali') else mode
    cfg = Config(dataset, batch_size=B, n_coms=K, mode=pmode, dim=dim, dim_latent=dl)
    cfg.z_samples = MG.Z_SAMPLES
    tr = Trainer(cfg, device=gpu, graph=False, inject_noise=True)
    tr.load_params(MG.perturbed_params(ocfg))
    feeds = iter(traj_feeds(z))
    for it in range(iters):
        r = tr.iteration(it, feeds)
        if it > 0:
            assert abs(float(r['gen_cost']) - z['costs'][it, 0]) <= 2e-3 * max(1.0, abs(z['costs'][it, 0]))
        if tr.cfg.critic_iters:
            assert abs(float(r['disc_cost']) - z['costs'][it, 1]) <= 2e-3 * max(1.0, abs(z['costs'][it, 1]))
        else:
            assert 'disc_cost' not in r and np.isnan(z['costs'][it, 1])
    P = tr.get_params()
    for k in z.files:
        if not k.startswith('p1/'):
            continue
        n = k[3:]
        if not tr.cfg.critic_iters and n.startswith('Discriminator'):
            continue   # critic-free modes: the oracle's parameter set carries an unused critic
        if ocfg.bn and (n.endswith('.Biases') or n == 'Generator.Input.b') and not n.startswith('Discriminator') \
                and n not in ('Extractor.1.Biases', 'Generator.5.Biases'):
            continue
        if mode == 'wali-gp' and n == 'Discriminator.Output.b':
            continue   # d(mean(D_fake) - mean(D_real) + GP)/d(output bias) == 0 exactly: Adam walks on rounding noise
        if mode == 'vegan' and ocfg.bn and n in ('Discriminator.Input.b', 'Discriminator.2.b', 'Discriminator.3.b', 'Discriminator.4.b'):
            continue   # latent critic: every hidden Linear feeds a BatchNorm (zero true gradient)
        d, ref = digest(P[n]), z[k]
        numel = P[n].size
        assert abs(d[1] - ref[1]) <= 1e-3 * ref[1] + 3e-4 * numel * 0.02 + 1e-6, (n, d[1], ref[1])
        assert np.abs(d[2:] - ref[2:]).max() <= 2.5 * tr.cfg.lr * iters * (1 + tr.cfg.critic_iters), n
    optim.reset_optimizers(); lib.delete_all_params()
