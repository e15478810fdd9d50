"""-m gpu: the HIP path at BASELINE sizes against the committed full-size fixtures (tests/golden/full_*.npz: float64 first-step
costs, critic logits and per-parameter gradient digests of every BASELINE configuration, generated offline by
tests/golden/make_golden_full.py).  Nothing of the oracle's arithmetic runs here: only its deterministic initial-weight and
feed generators, checked against the checksums in the fixture.  The fixtures' feeds keep every LeakyReLU input of the critics'
MLP layers clear of its kink (margin stored in the fixture), so the tolerances below carry no allowance for branch flips."""
import os
import sys

import numpy as np
import pytest

from _golden import load

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))


def _check_grads(z, which, names, grads, tol, allow_flip=False):
    """per tensor: the 64 fixture entries within tol * scale and the L2 norm within tol/3 (scale = max |g| of the tensor, floored at
    1e-2 of the largest gradient of the step for the mathematically-zero ones, e.g. a bias that feeds a BatchNorm)"""
    import make_golden_full as MG
    refs = {n: z['%s/g/%s' % (which, n)] for n in names if '%s/g/%s' % (which, n) in z.files}
    gmax = max(r[1] for r in refs.values())
    for n, g in zip(names, grads):
        if n not in refs:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        assert g is not None, n
        ref = refs[n]
        f = g.detach().cpu().numpy().astype(np.float64).reshape(-1)
        scale = max(ref[1], 1e-2 * gmax)
        idx = MG.sample_index(n, f.size)
        e = np.abs(f[idx] - ref[2:])
        err = e.max()
        if err > tol * scale and f.size <= 1024 and allow_flip:
            # A per-channel tensor (bias, BatchNorm offset / scale) shows ONE unit's activation-derivative flip in one entry: the
            # configurations with BatchNorm inside the critic (gan_inference_mnist.py) put ~6e6 ReLU / LeakyReLU units per step
            # behind fp32 batch statistics, the closest of them sits ~1e-6 (relative) from its kink, and which side single-precision
            # rounding takes there depends on the summation order of the statistics -- the fixture's CPU float32 screen cannot
            # predict the GPU's.  Granted only if it IS that: at most one distinct position off, by less than 1e-2 of the scale,
            # and the tensor's L2 norm (checked below, tol / 3) unaffected.
            bad = sorted(set(int(i) for i in idx[e > tol * scale]))
            assert len(bad) == 1 and err <= 1e-2 * scale, (which, n, 'entries', err, scale, bad)
            err = 0.0
        assert err <= tol * scale, (which, n, 'entries', err, scale)
        assert abs(np.linalg.norm(f) - ref[0]) <= tol / 3 * max(ref[0], 1e-2 * gmax * np.sqrt(f.size)), (which, n, 'l2', np.linalg.norm(f), ref[0])


@pytest.mark.parametrize('name', ['full_cifar_ali', 'full_cifar_wali_gp', 'full_cifar_gmgan_k30', 'full_cifar_gmgan_k10',
                                  'full_face_ali', 'full_face_gmgan_k100', 'full_mnist_ali', 'full_mnist_gmgan'])
def test_full_size_first_step_vs_fixture(gpu, name):
    import torch
    import make_golden_full as MG
    from oracle import nets as N, step as S
    from graphical_gan_amd import tflib as lib, optim
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models import Config
    z = load(name)
    dataset, B, K, mode = MG.FULL[name]
    ocfg = N.Cfg(dataset, batch_size=B, n_coms=K)
    P0 = MG.perturbed_params(ocfg)
    feed = S.make_feed(ocfg, np.random.default_rng(int(z['feed_seed'])), MG.omode_of(mode))
    assert MG.feed_checksum(feed) == int(z['feed_crc']), 'feed generator drifted from the one that made the fixture'
    assert float(z['margin']) >= MG.MARGIN and float(z['margin_relu']) >= MG.MARGIN_RELU
    optim.reset_optimizers()
    lib.delete_all_params()
    tr = Trainer(Config(dataset, batch_size=B, n_coms=K, mode=mode), device=gpu, graph=False, inject_noise=True)
    tr.load_params(P0)
    tr.set_feed(feed)
    for which in ('gen', 'disc'):
        out = tr.model.forward(tr.feed, which)
        ref = float(z[which + '/cost'])
        c = float(out[which + '_cost'].detach())
        assert abs(c - ref) <= 1e-5 * max(1.0, abs(ref)), (which, c, ref)
        df, dr = (out['disc_fake'], out['disc_real']) if not K else (out['disc_fake'][1], out['disc_real'][1])
        for key, t in (('disc_fake', df), ('disc_real', dr)) + ((('hyper_fake', out['disc_fake'][0]), ('hyper_real', out['disc_real'][0])) if K else ()):
            r = z['%s/%s' % (which, key)]
            assert np.abs(t.detach().cpu().numpy() - r).max() <= 2e-5 * max(1.0, np.abs(r).max()), (which, key)
        opt = out[which + '_train_op'].optimizer
        names = [p.param_name for p in opt.params]
        # (through the optimizer: a weight that two passes of the step reach comes back as a pair of contributions)
        grads = [(g[0] + g[1]) if isinstance(g, tuple) else g for g in opt.compute_gradients(out[which + '_cost'])]
        # Gradient tolerance at this size: 1e-3 of the tensor's max |g| per entry (3e-4 on its L2 norm); 2e-3 for the double backward
        # of the gradient penalty and with the mixture prior (its Gumbel-softmax divides 128-term squared distances by TEMP = 0.1
        # before exponentiating them).  Costs and logits stay at 1e-5 / 2e-5.  The nets hold ~7e6 ReLU / LeakyReLU units per pass;
        # a handful of them sit within fp32 rounding of their kink in ANY evaluation order, and each flip is a sparse O(1e-4..1e-3)
        # perturbation of some gradient tensor: two float32 evaluations that differ only in summation order (e.g. two tile shapes
        # of the same kernel, or PyTorch-CPU float32 vs float64: up to 1.4e-3 on unscreened feeds) do not agree tighter than that.
        # The fixture feeds are screened so that the Linear-layer activations are clear of their kinks and a float32 CPU
        # evaluation reproduces float64 to 3e-5; at the small sizes of tests/test_step_gpu.py the tolerance is 1e-4.
        loose = (mode == 'wali-gp' and which == 'disc') or K
        # (the one-flip allowance only where BatchNorm sits INSIDE the critic: the gan/gmgan_inference_mnist fixtures)
        _check_grads(z, which, names, grads, 2e-3 if loose else 1e-3, allow_flip=dataset == 'mnist')
    optim.reset_optimizers()
    lib.delete_all_params()


@pytest.mark.parametrize('name', ['full_cifar_ali', 'full_face_ali', 'full_cifar_gmgan_k30', 'full_cifar_wali_gp'])
def test_full_size_timed_configuration_vs_fixture(gpu, name):
    """The configuration bench.py TIMES, against the float64 fixture (round-3 review: the full-size fixtures ran graph=False on the default
    launch plan, the graph path was only compared with itself): batch 64, minibatches read in place from the device ring, generator
    step + critic step(s) captured as ONE HIP graph with the Generator / Extractor passes forked onto two streams and their conv
    launches planned for 128 workgroups (models.launch_hint / functional.target_workgroups), gradients packed and the Adam update
    applied by the pack launch.  The noise launch is replaced by the fixture's draws (inject_noise) and the learning rates are zero, so
    that every step of the replayed graph is evaluated at the fixture's weights: costs, critic logits, and every parameter's gradient as
    the pack launch left it in the optimizer's flat bucket."""
    import torch
    import make_golden_full as MG
    from oracle import nets as N, step as S
    from graphical_gan_amd import tflib as lib, optim
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models import Config
    z = load(name)
    dataset, B, K, mode = MG.FULL[name]
    ocfg = N.Cfg(dataset, batch_size=B, n_coms=K)
    P0 = MG.perturbed_params(ocfg)
    feed = S.make_feed(ocfg, np.random.default_rng(int(z['feed_seed'])), MG.omode_of(mode))
    assert MG.feed_checksum(feed) == int(z['feed_crc'])
    optim.reset_optimizers()
    lib.delete_all_params()
    cfg = Config(dataset, batch_size=B, n_coms=K, mode=mode)
    tr = Trainer(cfg, device=gpu, graph=True, inject_noise=True)
    tr.keep_outputs = True
    tr.load_params(P0)
    feeds = iter([feed] * 64)
    tr.iteration(0, feeds)                   # eager first calls: parameters' flat homes and the optimizers come into being
    tr.iteration(1, feeds)
    from graphical_gan_amd.optim import _optimizers
    for o in _optimizers.values():
        o.lr = 0.0                           # (baked into the captured update launches: the weights stay the fixture's)
    tr.load_params(P0)
    tr.set_feed(feed)
    x = torch.as_tensor(np.ascontiguousarray(feed['real_x_int'])).to(gpu)
    tr.use_ring([x] * 4)                     # every ring slot holds the fixture's minibatch
    res = None
    for it in (2, 3, 4):                     # capture (with its warm-up steps), then replays
        res = tr.iteration(it, None)
    assert getattr(tr, '_iter_graph', None) is not None and tr._iter_graph['kinds'] == ('gen',) + ('disc',) * cfg.critic_iters
    assert tr.model.fork_nets                # (the two-stream nets pass and its 128-workgroup plans were on while the graph was built)
    torch.cuda.synchronize()
    for which in ('gen', 'disc'):
        ref = float(z[which + '/cost'])
        c = float(res[which + '_cost'])
        assert abs(c - ref) <= 1e-5 * max(1.0, abs(ref)), (which, c, ref)
        out = tr.last_out[which]
        df, dr = (out['disc_fake'], out['disc_real']) if not K else (out['disc_fake'][1], out['disc_real'][1])
        for key, t in (('disc_fake', df), ('disc_real', dr)):
            r = z['%s/%s' % (which, key)]
            assert np.abs(t.detach().cpu().numpy() - r).max() <= 2e-5 * max(1.0, np.abs(r).max()), (which, key)
        opt = next(o for k, o in _optimizers.items() if k[0] == which)
        names = [p.param_name for p in opt.params]
        grads = [opt.g[o:o + n].view(p.shape) for p, (o, n) in zip(opt.params, opt.slots)]
        loose = (mode == 'wali-gp' and which == 'disc') or K
        _check_grads(z, which, names, grads, 2e-3 if loose else 1e-3)
    # the weights did not move (lr 0) and match the fixture's
    got = tr.get_params()
    for n_, v in P0.items():
        if not (n_.endswith('.moving_mean') or n_.endswith('.moving_variance')):
            assert np.array_equal(got[n_].reshape(-1), np.asarray(v, np.float32).reshape(-1)), n_
    optim.reset_optimizers()
    lib.delete_all_params()


def test_full_size_ssgan_first_step_vs_fixture(gpu):
    """ssgan_inference_moving_mnist.py at BASELINE configs[4]: 32 sequences x 16 frames of 64x64"""
    import torch
    import make_golden_full as MG
    from oracle import ssgan as O
    from graphical_gan_amd import tflib as lib, optim
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models_ssgan import SSConfig, StateSpaceGAN
    name = 'full_ssgan_b32_t16'
    z = load(name)
    ocfg = O.Cfg(**MG.SSGAN[name])
    P0 = MG.ssgan_params(ocfg)
    feed = O.make_feed(ocfg, np.random.default_rng(int(z['feed_seed'])))
    assert MG.feed_checksum(feed) == int(z['feed_crc'])
    optim.reset_optimizers()
    lib.delete_all_params()
    cfg = SSConfig(**MG.SSGAN[name])
    tr = Trainer(cfg, device=gpu, graph=False, inject_noise=True, model=StateSpaceGAN(cfg))
    tr.load_params(P0)
    tr.set_feed(feed)
    fx = tr.model.forward_nets(tr.feed)['fake_x'].detach().cpu().numpy().astype(np.float64).reshape(-1)
    ref = z['fake_x_digest']
    assert np.abs(fx[MG.sample_index('fake_x', fx.size)] - ref[2:]).max() <= 2e-5 and abs(np.linalg.norm(fx) - ref[0]) <= 1e-5 * ref[0]
    for which in ('gen', 'disc'):
        out = tr.model.forward(tr.feed, which)
        refc = float(z[which + '/cost'])
        c = float(out[which + '_cost'].detach())
        assert abs(c - refc) <= 1e-5 * max(1.0, abs(refc)), (which, c, refc)
        opt = out[which + '_train_op'].optimizer
        names = [p.param_name for p in opt.params]
        grads = torch.autograd.grad(out[which + '_cost'], opt.params, allow_unused=True)
        _check_grads(z, which, names, grads, 1e-3)       # (as above; filter gradients here are fp32 sums over up to 5e5 pixels)
    optim.reset_optimizers()
    lib.delete_all_params()
