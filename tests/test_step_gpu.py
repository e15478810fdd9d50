"""-m gpu: the full training iteration (all nets, objective, backward, TF-Adam; incl. the wali-gp double
backward) on the HIP path vs the float64 oracle, same injected weights / minibatches / noise.

Tolerances: costs rel <= 1e-5 on the first session.run, <= 1e-3 after a few Adam steps
(Adam moves every weight by ~lr regardless of |g|, so fp32 sign flips of near-zero gradients are amplified);
logits / gradients rel <= 1e-4 (GP second-order <= 1e-3 of the largest entry).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fresh():
    from graphical_gan_amd import tflib as lib
    from graphical_gan_amd import optim
    optim.reset_optimizers()
    lib.delete_all_params()


def _mk(dataset, B, K, mode, dim, dl, fuse, graph, gpu, z_samples=None, bn=None):
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.engine import Trainer
    from oracle import nets as N
    agg = mode in ('vegan-kl', 'vegan-ikl', 'vegan-jsd')
    ocfg = N.Cfg(dataset, batch_size=B, n_coms=K, dim=dim, dim_latent=dl, latent_critic=mode.startswith('vegan') and not agg, learn_std=agg,
                 z_samples=z_samples or 100, bn=bn)
    P0 = N.init_params(ocfg, seed=0)
    rng = np.random.default_rng(7)
    for k in P0:   # make biases / BN params non-trivial so their gradients paths are exercised
        if P0[k].ndim <= 2 and ('Biases' in k or k.endswith('.b') or 'offset' in k):
            P0[k] = (0.1 * rng.standard_normal(P0[k].shape)).astype(np.float32)
        if k.endswith('.scale'):
            P0[k] = (1 + 0.1 * rng.standard_normal(P0[k].shape)).astype(np.float32)
    _fresh()
    cfg = Config(dataset, batch_size=B, n_coms=K, mode=mode, dim=dim, dim_latent=dl, fuse=fuse, bn=bn)
    if z_samples:
        cfg.z_samples = z_samples
    tr = Trainer(cfg, device=gpu, graph=graph, inject_noise=True)
    tr.load_params(P0)
    return ocfg, P0, cfg, tr


UPDATE_REL_TOL = 0.02      # deviation of the post-step weights from the oracle's, relative to the norm of the oracle's total update

CASES = [  # dataset, B, K, mode, dim, dim_latent
    ('cifar10', 8, 0, 'ali', 8, 16),
    ('cifar10', 8, 5, 'local_ep', 8, 16),
    ('cifar10', 8, 0, 'wali-gp', 8, 16),
    ('cifar10', 8, 0, 'wali', 8, 16),          # RMSProp + critic weight clipping, CRITIC_ITERS 5
    ('cifar10', 8, 0, 'alice', 8, 16),         # + l2(real_x, G(q_z)) + l2(p_z, E(fake_x))
    ('cifar10', 8, 0, 'alice-x', 8, 16),
    ('cifar10', 8, 5, 'local_epce', 8, 16),    # gmgan + l2(real_x, G(q_z))
    ('mnist', 6, 4, 'local_ep', 8, 16),
    ('mnist', 6, 0, 'ali', 8, 16),             # gan_inference_mnist.py: the critic with BatchNorm + Discriminator.2 / zx2 Linear layers
    ('mnist', 6, 0, 'wali-gp', 8, 16),         # ... differentiated twice: second derivative of an NCHW BatchNorm backward
    ('face', 4, 6, 'local_ep', 4, 16),
    ('svhn', 8, 5, 'local_ep', 8, 16),         # the CIFAR nets without BatchNorm (g(m)gan_inference_svhn.py)
    ('cifar10', 8, 0, 'vegan', 8, 16),         # latent MLP critic with BatchNorm + Gaussian noise layers, + l2(real_x, G(q_z))
    ('svhn', 8, 0, 'vegan-wgan-gp', 8, 16),    # latent critic differentiated twice (no BatchNorm)
    ('cifar10', 8, 0, 'vegan-wgan-gp', 8, 16), # ... with BatchNorm: second derivative of the BatchNorm backward
    ('cifar10', 64, 0, 'ali', None, 128),      # BASELINE config 2 at full size
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: '-'.join(str(x) for x in c))
@pytest.mark.parametrize('fuse', [True, False], ids=['fused', 'unfused'])
def test_first_step_costs_and_grads(gpu, case, fuse):
    """One forward/backward: costs, critic logits and every parameter gradient vs the oracle."""
    import torch
    from oracle import step as S, tape as tp, nets as N
    dataset, B, K, mode, dim, dl = case
    if dim is None and not fuse:
        pytest.skip('full-size case runs fused only (oracle time)')
    ocfg, P0, cfg, tr = _mk(dataset, B, K, mode, dim, dl, fuse, False, gpu)
    omode = mode if mode in ('wali', 'wali-gp', 'alice', 'alice-z', 'alice-x', 'local_epce', 'vegan', 'vegan-wgan-gp') else 'ali'
    feed = S.make_feed(ocfg, np.random.default_rng(11), omode)
    Pt = {k: tp.T(v.astype(np.float64)) for k, v in P0.items()}
    tp.KINK_LOG = kink_log = []
    try:
        oout = S.forward(ocfg, Pt, feed, omode)
    finally:
        tp.KINK_LOG = None
    tr.set_feed(feed)
    out = tr.model.forward(tr.feed)
    for which in ('gen', 'disc'):
        oc = float(oout[which + '_cost'].v)
        c = float(out[which + '_cost'].detach())
        assert abs(c - oc) <= 1e-5 * max(1.0, abs(oc)), (which, c, oc)
        opt = out[which + '_train_op'].optimizer
        names = [p.param_name for p in opt.params]
        grads = torch.autograd.grad(out[which + '_cost'], opt.params, allow_unused=True, retain_graph=True)
        ogs = tp.grad(oout[which + '_cost'], [Pt[n] for n in names])
        tol = 1e-3 if mode in ('wali-gp', 'vegan-wgan-gp') and which == 'disc' else 1e-4
        gmax = max(np.abs(og.v).max() for og in ogs if og is not None)   # scale for mathematically-zero grads
        for n, g, og in zip(names, grads, ogs):
            if og is None:
                assert g is None or float(g.abs().max()) == 0.0, n
                continue
            ref = og.v
            err = np.abs(g.cpu().numpy().reshape(ref.shape) - ref)
            scale = max(np.abs(ref).max(), 1e-2 * gmax)
            if err.max() <= tol * scale:
                continue
            # LeakyReLU/ReLU kinks: a pre-activation within fp32 rounding of 0 takes the other branch in fp32 than in
            # the fp64 oracle.  In a late critic layer that re-weights ONE sample's whole backward signal, i.e. a rank-1
            # perturbation of every weight gradient; in the Generator's first layer (ReLU on a Linear output, 64 terms per
            # weight-gradient entry) it changes one column of that gradient.  The allowance below is granted only when
            # the oracle's own forward pass PROVES that such near-kinks exist in the Linear-layer activations and that they
            # are a handful (_kink_samples); the committed full-size fixtures pick feeds that avoid the situation
            # altogether and are checked without any allowance (tests/test_golden_full_gpu.py).
            kinks = _kink_samples(kink_log)
            assert 1 <= len(kinks) <= 8, (which, n, err.max(), scale, 'deviation without a provable (and rare) near-kink', kinks)
            l2 = np.linalg.norm(err) / (np.linalg.norm(ref) + 1e-30)
            assert np.median(err) <= tol * scale and l2 <= 2e-3, (which, n, err.max(), np.median(err), l2, scale)


def _kink_samples(log, margin=1e-5):
    """(layer call, row) pairs whose Linear-layer ReLU / LeakyReLU inputs come within `margin` (relative to the row's rms) of zero
    in the float64 oracle forward pass (oracle.tape.KINK_LOG): the places where an fp32 evaluation can legitimately take the other
    branch for a unit that carries a macroscopic share of a weight-gradient entry"""
    rows = []
    for i, x in enumerate(log):
        rms = np.sqrt((x ** 2).mean(1, keepdims=True)) + 1e-30
        rows += [(i, int(r)) for r in np.nonzero((np.abs(x) / rms).min(1) < margin)[0]]
    return rows


@pytest.mark.parametrize("case", CASES[:15], ids=lambda c: '-'.join(str(x) for x in c))
@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hipgraph'])
def test_trajectory(gpu, case, graph):
    """3 iterations of the loop (iteration 0 = critic only), scripted minibatches + noise: cost sequence and
    post-step weights vs the oracle; eager and HIP-graph replay must agree with it equally."""
    from oracle import step as S
    dataset, B, K, mode, dim, dl = case
    ocfg, P0, cfg, tr = _mk(dataset, B, K, mode, dim, dl, True, graph, gpu)
    omode = mode if mode in ('wali', 'wali-gp', 'alice', 'alice-z', 'alice-x', 'local_epce', 'vegan', 'vegan-wgan-gp') else 'ali'
    otr = S.Trainer(ocfg, P0, omode, np.float64)
    n_it = 4 if graph else 3        # graph path: call 1 eager, call 2 captures+replays, ...
    n_feeds = n_it * (1 + otr.critic_iters)
    feeds = [S.make_feed(ocfg, np.random.default_rng(100 + i), omode) for i in range(n_feeds)]
    fo, fp = iter(feeds), iter(feeds)
    for it in range(n_it):
        ro = otr.iteration(it, fo)
        rp = tr.iteration(it, fp)
        for k in ro:
            a, b = float(rp[k]), ro[k]
            assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (it, k, a, b)
    P = tr.get_params()
    lr = cfg.lr
    steps = n_it * (1 + otr.critic_iters)
    worst = 0.0
    for n, ref in otr.P.items():
        if cfg.bn and (n.endswith('.Biases') or n == 'Generator.Input.b') and not n.startswith('Discriminator') \
                and n not in ('Extractor.1.Biases', 'Generator.5.Biases'):
            continue   # bias feeding BatchNorm: true gradient is 0, Adam random-walks on rounding noise (fp64 too)
        if mode in ('wali-gp', 'vegan-wgan-gp') and n == 'Discriminator.Output.b':
            continue   # Wasserstein critic cost: the output bias cancels exactly, same random walk
        if cfg.bn and cfg.latent_critic and n in ('Discriminator.Input.b', 'Discriminator.2.b', 'Discriminator.3.b', 'Discriminator.4.b'):
            continue   # latent critic: every hidden Linear feeds a BatchNorm
        if cfg.bn and getattr(cfg, 'critic_deep', False) and n in ('Discriminator.2.Biases', 'Discriminator.3.Biases'):
            continue   # gan_inference_mnist.py: the critic's conv 2 / 3 feed a BatchNorm
        if n.endswith(('moving_mean', 'moving_variance')):
            assert np.array_equal(P[n].reshape(ref.shape), ref.astype(np.float32)), n      # never written
            continue
        d = np.abs(P[n].reshape(ref.shape) - ref)
        assert d.max() <= 2.5 * lr * steps, (n, d.max())
        assert (d > 2e-5).mean() <= 0.02, (n, (d > 2e-5).mean())
        # relative to what the steps did to this tensor: the deviation from the oracle's weights against the oracle's own
        # total update (Adam moves every entry by ~lr per step; an entry whose gradient is rounding noise may take the other sign)
        # (floor: a float32 weight cannot follow an update below its own resolution -- RMSProp's lr*g steps on small gradients --
        #  and a tensor whose true gradient is zero only random-walks on rounding noise, in the oracle as well)
        upd = np.linalg.norm((ref - P0[n].astype(np.float64)).ravel())
        floor = 4 * 6e-8 * np.linalg.norm(ref.ravel()) * np.sqrt(steps)
        if upd > 0.05 * lr * np.sqrt(ref.size):
            rel = max(np.linalg.norm(d.ravel()) - floor, 0.0) / upd
            worst = max(worst, rel)
            assert rel <= UPDATE_REL_TOL, (n, rel, upd, floor)
    print('trajectory %s graph=%s: worst ||P - P_oracle|| / ||oracle update|| = %.4f' % ('-'.join(str(x) for x in case), graph, worst))


@pytest.mark.parametrize('mode', ['ali', 'wali-gp'])
def test_split_graph_path_matches_single_graph(gpu, monkeypatch, mode):
    """The data-parallel step runs as [nets graph] (pending critic Adam) [critic + backward + pack graph] -> all-reduce ->
    [Adam graph].  Exercise that exact code path on one GPU (1-rank no-op collective) and require bit-identical weights vs
    the single-graph path (wali-gp: five critic steps per generator step, the deferred Adam lands under the next critic step)."""
    import torch
    from oracle import step as S
    res = []
    for force in (False, True):
        if force:
            monkeypatch.setenv('GGAN_FORCE_SPLIT_GRAPH', '1')
        else:
            monkeypatch.delenv('GGAN_FORCE_SPLIT_GRAPH', raising=False)
        ocfg, P0, cfg, tr = _mk('cifar10', 8, 0, mode, 8, 16, True, True, gpu)
        assert tr.split_graph == force
        feeds = iter([S.make_feed(ocfg, np.random.default_rng(300 + i), mode) for i in range(40)])
        for it in range(5):
            tr.iteration(it, feeds)
        torch.cuda.synchronize()
        if force:      # the generator step ran as [nets][critic + Generator bucket][Extractor bucket][Adam]
            assert tr._graphs['gen']['g0'] is not None and tr._graphs['gen']['g1b'] is not None
            assert tr._graphs['disc']['g0'] is not None and tr._graphs['disc']['g1b'] is None
        res.append(tr.get_params())
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_fused_conv_backward_entry_points(gpu):
    """ggan_conv2d_bwd_{data,filter}_act: gy*act'(y) applied on staging + bias gradient out of the filter kernel."""
    import torch
    from graphical_gan_amd import functional as F
    from oracle import ops as O
    rng = np.random.default_rng(21)
    for (N, Ci, H, Co) in [(64, 64, 16, 128), (8, 3, 32, 64), (5, 128, 7, 256)]:
        geom = F.conv_geom(N, Ci, H, H, Co, 5, 2)
        Ho = geom[5]
        x = rng.standard_normal((N, Ci, H, H)).astype(np.float32)
        w = (rng.standard_normal((5, 5, Ci, Co)) / np.sqrt(25 * Ci)).astype(np.float32)
        b = rng.standard_normal(Co).astype(np.float32)
        gy = rng.standard_normal((N, Co, Ho, Ho)).astype(np.float32)
        xd = torch.as_tensor(x, device=gpu).requires_grad_(True)
        wd = torch.as_tensor(w, device=gpu).requires_grad_(True)
        bd = torch.as_tensor(b, device=gpu).requires_grad_(True)
        y = F.ConvFwd.apply(xd, wd, bd, geom, F.ACT_LRELU, 0.2)
        gx, gw, gb = torch.autograd.grad(y, [xd, wd, bd], grad_outputs=torch.as_tensor(gy, device=gpu))
        pre = O.conv2d(x.astype(np.float64), w.astype(np.float64), 2) + b.reshape(1, -1, 1, 1)
        g2 = gy.astype(np.float64) * np.where(pre > 0, 1.0, 0.2)
        assert _rel_(gx.cpu().numpy(), O.conv2d_bwd_data(g2, w.astype(np.float64), (H, H), 2)) < 3e-5
        assert _rel_(gw.cpu().numpy(), O.conv2d_bwd_filter(x.astype(np.float64), g2, 5, 2)) < 3e-5
        assert _rel_(gb.cpu().numpy(), g2.sum(axis=(0, 2, 3))) < 3e-5


def _rel_(a, ref):
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30))


def test_checkpoint_save_restore_resumes_bit_exact(gpu, tmp_path):
    """train 3 iterations, save; (a) train 2 more; (b) fresh registry + restore + the same 2 iterations -> identical weights
    (parameters, both Adam moments and step counters travel through the .npz keyed by registry names)."""
    import torch
    from graphical_gan_amd import checkpoint
    from oracle import step as S
    ocfg, P0, cfg, tr = _mk('cifar10', 8, 0, 'ali', 8, 16, True, False, gpu)
    feeds = [S.make_feed(ocfg, np.random.default_rng(500 + i), 'ali') for i in range(12)]
    it_feeds = iter(feeds)
    for it in range(3):
        tr.iteration(it, it_feeds)
    path = str(tmp_path / 'ck.npz')
    keys = checkpoint.save(path, tr)
    assert 'Generator.Input.W' in keys and 'adam/gen/step' in keys and 'adam/disc/Discriminator.1.Filters/m' in keys
    used = 5                                             # it 0: 1 feed, it 1-2: 2 feeds each
    rest = feeds[used:]
    a = iter(rest)
    for it in range(3, 5):
        tr.iteration(it, a)
    PA = tr.get_params()
    ocfg, P0, cfg, tr2 = _mk('cifar10', 8, 0, 'ali', 8, 16, True, False, gpu)     # fresh registry, initial weights
    checkpoint.restore(path, tr2)                        # optimizers do not exist yet: Adam state is parked
    b = iter(rest)
    for it in range(3, 5):
        tr2.iteration(it, b)
    PB = tr2.get_params()
    for k in PA:
        assert np.array_equal(PA[k], PB[k]), k


def test_device_prefetcher_feeds_trainer(gpu):
    """pinned double-buffered H->D staging (graphical_gan_amd/data.py): same minibatches, same order, same training result as
    feeding device-resident tensors."""
    import torch
    from graphical_gan_amd.data import DevicePrefetcher
    rng = np.random.default_rng(0)
    host = [rng.integers(0, 256, size=(8, 3072)).astype(np.uint8) for _ in range(5)]
    pf = DevicePrefetcher(lambda: iter(host), gpu, depth=3, dtypes=[np.int32])
    for i in range(12):                                  # wraps around the 5-batch "epoch" and the 3-slot ring
        t = next(pf)
        assert t.dtype == torch.int32 and np.array_equal(t.cpu().numpy(), host[i % 5].astype(np.int32))
    res = []
    for use_pf in (False, True):
        ocfg, P0, cfg, tr = _mk('cifar10', 8, 0, 'ali', 8, 16, True, False, gpu)
        tr.inject_noise = False
        torch.manual_seed(77)
        if use_pf:
            src = DevicePrefetcher(lambda: iter(host), gpu, dtypes=[np.int32])
        else:
            src = iter([torch.as_tensor(h.astype(np.int32)).to(gpu) for h in host] * 3)
        for it in range(3):
            tr.iteration(it, src)
        res.append(tr.get_params())
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_driver_loop_and_counterpart_script(gpu, tmp_path):
    """graphical_gan_amd/run.py (the scripts' train loop: plot log, checkpoint, sample grid) on a small config, and one of the
    counterpart scripts end to end for 3 iterations (synthetic minibatches: no dataset on the box)."""
    import os, subprocess, sys
    from graphical_gan_amd import run
    from graphical_gan_amd.models import Config
    _fresh()
    cfg = Config('cifar10', batch_size=8, mode='ali', dim=8, dim_latent=16)
    S = dict(DATASET='cifar10', BATCH_SIZE=8, ITERS=4, SAVE_EVERY=2, LOG_EVERY=2, OUT_DIR=str(tmp_path), DATA_DIR='/nonexistent')
    with pytest.raises(FileNotFoundError):          # a missing dataset raises, as the reference scripts do ...
        run.train(S, cfg)
    _fresh()
    S['SYNTHETIC'] = True                            # ... the synthetic ring is an explicit opt-in
    run.train(S, cfg)
    assert 'data source: synthetic' in open(str(tmp_path / 'logfile.txt')).read()
    names = sorted(os.listdir(str(tmp_path)))
    assert 'params_2.npz' in names and 'params_4.npz' in names and 'samples_4.png' in names and 'logfile.txt' in names
    z = np.load(str(tmp_path / 'params_4.npz'))
    assert 'Discriminator.zx1.W' in z.files and int(z['adam/disc/step'][0]) == 4 and int(z['adam/gen/step'][0]) == 3
    assert str(z['meta/data_source']) == 'synthetic'
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'gmgan_inference_cifar10.py'), '3'], capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, GGAN_DATA_DIR='/nonexistent', GGAN_SYNTHETIC='1'))
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'iter 2' in r.stdout and 'disc cost' in r.stdout


def test_driver_loop_on_loader_data(gpu, tmp_path, monkeypatch):
    """real-data plumbing: a (fake) mnist.pkl.gz and CIFAR batch files on disk -> py3 loaders -> pinned ring + copy kernel ->
    Trainer, for the MNIST, CIFAR and moving-MNIST drivers."""
    import gzip, os, pickle
    from graphical_gan_amd import run
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.models_ssgan import SSConfig, StateSpaceGAN
    rng = np.random.default_rng(0)
    mk = lambda n: (rng.random((n, 784), dtype=np.float32), rng.integers(0, 10, size=n))
    with gzip.open(str(tmp_path / 'mnist.pkl.gz'), 'wb') as f:
        pickle.dump((mk(64), mk(16), mk(16)), f)
    monkeypatch.setenv('GGAN_MNIST', str(tmp_path / 'mnist.pkl.gz'))
    for i in list(range(1, 6)) + ['t']:
        name = 'test_batch' if i == 't' else 'data_batch_%d' % i
        with open(str(tmp_path / name), 'wb') as f:
            pickle.dump({'data': rng.integers(0, 256, size=(16, 3072)).astype(np.uint8), 'labels': list(range(16))}, f)
    _fresh()
    tr = run.train(dict(DATASET='mnist', BATCH_SIZE=8, ITERS=5, LOG_EVERY=2), Config('mnist', batch_size=8, dim=8, dim_latent=16))
    assert tr.feed['real_x'].shape == (8, 784) and float(tr.feed['real_x'].max()) <= 1.0
    _fresh()
    tr = run.train(dict(DATASET='cifar10', BATCH_SIZE=8, ITERS=5, LOG_EVERY=2, DATA_DIR=str(tmp_path)),
                   Config('cifar10', batch_size=8, dim=8, dim_latent=16))
    assert tr.feed['real_x_int'].dtype.is_floating_point is False and int(tr.feed['real_x_int'].max()) <= 255
    # (int32 image data: from the third iteration on the loader feeds the device ring one iteration ahead, one graph per iteration)
    assert getattr(tr, '_feeder', None) is not None and tr._iter_graph is not None and int(tr.feed['ring'][0].max()) <= 255
    _fresh()
    cfg = SSConfig(batch_size=4, length=3, dim=4, dim_op=16, dim_g=8, dim_l=4)
    tr = run.train(dict(DATASET='moving_mnist', BATCH_SIZE=4, LEN=3, N_C=10, ITERS=4, LOG_EVERY=2), cfg, model=StateSpaceGAN(cfg))
    x = tr.feed['real_x_unit'].cpu().numpy().reshape(4, 3, 64, 64)
    assert x.max() <= 1.0 and (x.sum(axis=(2, 3)) > 0).all()             # every frame carries its digit
    assert np.allclose(tr.feed['real_y'].cpu().numpy().sum(1), 1.0)


@pytest.mark.parametrize('mode', ['ali', 'wali-gp'])
def test_dp_two_ranks_on_one_gpu(gpu, tmp_path, mode):
    """End-to-end data-parallel control flow with REAL cross-rank collectives: two processes (gloo on device tensors) share the
    one GPU, same data and seeds on both, so the averaged gradient equals each rank's own ((g+g)*0.5 is exact) and the result
    must be bit-identical to a single process running the cut-graph path.  Catches rank-asymmetric control flow (hangs),
    ordering of the asynchronous sub-bucket exchanges against the graphs, and the deferred critic Adam."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, 'tests', '_dp_worker.py')
    env = dict(os.environ, GGAN_FORCE_SPLIT_GRAPH='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    one = str(tmp_path / 'one.npz')
    r = subprocess.run([sys.executable, worker, one, mode], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    two = str(tmp_path / 'two.npz')
    port = 29600 + (os.getpid() % 300)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), worker, two, mode], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    a, b = np.load(one), np.load(two)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k


def test_sync_batchnorm_two_ranks_match_one_process(gpu, tmp_path):
    """with cross-replica BatchNorm statistics, 2 processes x B/2 reproduce 1 process x B (same global batch cut
    in halves, gloo collectives on device tensors of the one GPU) -- weights after 3 iterations within fp32 tolerance."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, 'tests', '_dp_worker.py')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    one = str(tmp_path / 'one.npz')
    r = subprocess.run([sys.executable, worker, one, 'ali:syncbn'], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    two = str(tmp_path / 'two.npz')
    port = 29900 + (os.getpid() % 90)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), worker, two, 'ali:syncbn'], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    a, b = np.load(one), np.load(two)
    for k in a.files:
        # a bias that feeds a BatchNorm has an exactly-zero true gradient: what it receives is rounding noise, which Adam turns
        # into +-lr steps (g / (|g| + eps)), so those entries random-walk by up to 2 * lr * steps whatever the arithmetic order
        slack = 2 * 2e-4 * 3 if k.endswith(('.Biases', '.b')) else 0.0
        d = np.abs(a[k] - b[k]).max()
        assert d <= 2e-3 * max(np.abs(a[k]).max(), 1e-3) + slack, (k, d)


def test_trajectory_unfused_epilogues(gpu):
    """the same 3-iteration trajectory with every pointwise op as its own kernel (Config(fuse=False)): the batched critic with
    the pruned data-gradient (grad_rows) must not depend on the fused epilogues."""
    from oracle import step as S
    ocfg, P0, cfg, tr = _mk('cifar10', 8, 0, 'ali', 8, 16, False, False, gpu)
    otr = S.Trainer(ocfg, P0, 'ali', np.float64)
    feeds = [S.make_feed(ocfg, np.random.default_rng(100 + i), 'ali') for i in range(6)]
    fo, fp = iter(feeds), iter(feeds)
    for it in range(3):
        ro, rp = otr.iteration(it, fo), tr.iteration(it, fp)
        for k in ro:
            assert abs(float(rp[k]) - ro[k]) <= 2e-3 * max(1.0, abs(ro[k])), (it, k, float(rp[k]), ro[k])
    P = tr.get_params()
    for n in ('Discriminator.2.Filters', 'Generator.3.Filters', 'Extractor.Output.W', 'Discriminator.zx1.W'):
        d = np.abs(P[n].reshape(otr.P[n].shape) - otr.P[n])
        assert d.max() <= 2.5 * cfg.lr * 6 and (d > 2e-5).mean() <= 0.02, (n, d.max())


def test_bench_contract_line_single_and_two_ranks(gpu):
    """bench.py end to end: the one-line JSON contract at N=1 (with roofline and cpu_baseline objects), and the N=2 launch the
    driver uses (python -m torch.distributed.run ... bench.py --gpus 2) rehearsed with two gloo ranks sharing the one GPU."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env['GGAN_BENCH_CPU_BUDGET_S'] = '2'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '6', '--warmup', '2', '--variant-steps', '4'],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    # the driver keeps the tail of stdout: the LAST line must parse on its own and stay well under 4 KB
    last = r.stdout.rstrip('\n').splitlines()[-1]
    assert len(last) < 4096, len(last)
    d = json.loads(last)
    assert len([l for l in r.stdout.splitlines() if l.startswith('{')]) == 1
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 6 and d['warmup'] == 2 and d['dtype'] == 'f32' and d['value'] > 0
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_eager', 'frac_in_graph', 'algorithmic_bytes')) <= set(d['roofline'])
    assert abs(d['roofline']['frac'] - d['roofline']['achieved'] / d['roofline']['peak']) < 1e-3
    assert d['roofline']['algorithmic_bytes'] > 0
    assert set(('value', 'unit', 'cores', 'kind', 'sample')) <= set(d['cpu_baseline']) and d['cpu_baseline']['value'] > 0
    assert 'workload' in d['config'] and 'model' not in d['config']
    # every BASELINE configuration rides along as a SHORT record; the long form is the full record next to it
    assert 'G+D+GP' in d['metric'] and 'MODE=wali-gp' in d['config']['workload'] and d['config']['minibatches_per_step'] == 6
    assert d['g_d_step']['value'] > d['value']           # (the script's default MODE='ali': one critic step, no penalty)
    assert [v['key'] for v in d['variants']] == ['ali', 'gmgan-cifar10-K30', 'gmgan-cifar10-K10', 'gan-face', 'ssgan-moving-mnist',
                                                 'ssgan-moving-mnist-3dcnn']
    for v in d['variants']:
        assert v['value'] > 0 and v['ms_per_step'] > 0 and v['frac'] > 0 and v['cpu'] > 0 and v['finite'], v['key']
    full = json.load(open(os.path.join(root, d['full_record'])))
    assert full['value'] == d['value'] and len(full['variants']) == 6
    for v in full['variants']:
        assert v['algorithmic_gflop_per_step'] > 0 and v['roofline']['frac'] > 0 and v['cpu_baseline']['value'] > 0, v['key']
    assert 'G+D step' in full['variants'][0]['metric'] and 'N_COMS=10' in full['variants'][2]['config']['workload']
    # the roofline object is one record: the kernel-level figures are sums / launch-mix averages of its per-grid rows
    rf = full['roofline']
    mix = rf['launch_mix']
    n = sum(r['launches_per_step'] for r in mix)
    assert n > 0 and abs(sum(r['launches_per_step'] * r['flop_per_launch'] for r in mix) / n - rf['flop_per_launch']) <= 1e-6 * rf['flop_per_launch']
    assert abs(sum(r['launches_per_step'] * r['algorithmic_bytes'] for r in mix) / n - rf['algorithmic_bytes']) <= 2 + 1e-6 * rf['algorithmic_bytes']
    assert abs(sum(r['launches_per_step'] * r['avg_us'] for r in mix) / n - rf['avg_launch_us_eager']) <= 0.02 * rf['avg_launch_us_eager']
    # one row = one problem shape; the headline figure is the in-graph one whenever the committed trace covers the launched shapes
    assert len(set(r['grid'] for r in mix)) == len(mix)           # (one row per grid; shapes that share a grid are listed inside their row)
    for r in mix:
        if 'shapes' in r:
            assert abs(sum(q['launches_per_step'] for q in r['shapes']) - r['launches_per_step']) < 1e-9
    assert rf['frac_basis'] in ('in_graph', 'eager_bracket')
    if rf['frac_basis'] == 'in_graph':
        assert abs(rf['frac'] - rf['frac_in_graph']) < 1e-3
        assert abs(sum(r['launches_per_step'] * r['avg_us_in_graph'] for r in mix) / n - rf['avg_launch_us']) <= 0.02 * rf['avg_launch_us']
        assert abs(rf['frac'] - rf['flop_per_launch'] / (rf['avg_launch_us'] * 1e-6) / 1e12 / rf['peak']) < 2e-3
    else:
        assert abs(rf['frac'] - rf['frac_eager']) < 1e-3
    port = 29700 + (os.getpid() % 200)
    env2 = dict(env, GGAN_DIST_BACKEND='gloo')
    # exactly as the driver invokes it: no launcher around it -- bench.py starts its own ranks (round-3 review: this form used to exit)
    env2 = {k: v for k, v in env2.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
                        '--variants', 'gmgan-cifar10-K10', '--variant-steps', '4'],
                       capture_output=True, text=True, timeout=900, env=env2, cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and len(lines[0]) < 4096         # rank 0 only
    d2 = json.loads(r.stdout.rstrip('\n').splitlines()[-1])
    assert d2['n_gpus'] == 2 and d2['scaling'] == 'weak' and d2['config']['global_batch'] == 128 and d2['value'] > 0
    assert d2['config']['parallelism'] == 'dp2' and d2['data_parallel']['ranks'] == 2 and d2['data_parallel']['launched_by'] == 'bench.py itself'
    assert 'diag_env' not in d2
    assert len(d2['variants']) == 1 and d2['variants'][0]['key'] == 'gmgan-cifar10-K10' and d2['variants'][0]['value'] > 0
    # a rank that dies in the first attempt the way a failed captured collective kills it (abort): the supervisors stop that
    # attempt on every rank and the second one -- exchanges issued by the host between cut graphs -- delivers the line
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port + 3), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2',
                        '--no-variants', '--no-cpu-baseline', '--no-kernel-profile', '--repeats', '0'],
                       capture_output=True, text=True, timeout=900, env=dict(env2, GGAN_BENCH_ABORT_TEST='1:1'), cwd=root)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2 and 'retrying with host-issued exchanges' in r.stderr


def test_bench_line_stays_last_when_rccl_writes_to_stdout(gpu):
    """librccl prints its version banner to STDOUT through C stdio (block-buffered on a pipe: it would surface at exit, behind the
    JSON line, and the driver reads the LAST line).  One rank over RCCL (the forced-exchange rehearsal) initialises the library in the
    bench process: the contract line must still be the last line, and the only JSON one."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GGAN_FORCE_ALLREDUCE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(29900 + (os.getpid() % 90)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '6', '--warmup', '2', '--no-variants',
                        '--no-cpu-baseline', '--no-kernel-profile', '--repeats', '0'],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.rstrip('\n').splitlines()
    d = json.loads(lines[-1])
    assert d['n_gpus'] == 1 and d['value'] > 0 and len([l for l in lines if l.startswith('{')]) == 1


def test_rccl_exchange_inside_the_step_graph_one_rank_rehearsal(gpu):
    """The driver's N > 1 launch puts the gradient exchange (RCCL all-reduce) INSIDE the captured graphs.  One rank over RCCL is
    enough to rehearse that on a single-GPU box (GGAN_FORCE_ALLREDUCE): the exchange goes through the directly bound communicator
    (graphical_gan_amd/rccl.py: no process-group watchdog polls a captured stream, nothing to wait out before a capture), is
    captured, replayed, and the costs stay finite, for the headline workload (one graph per iteration) and one variant."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GGAN_FORCE_ALLREDUCE='1')
    port = 29400 + (os.getpid() % 200)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '6', '--warmup', '2',
                        '--variants', 'ali', '--variant-steps', '3', '--no-cpu-baseline', '--repeats', '0'],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['value'] > 0 and d['config']['finite_costs'] and 'one graph per iteration' in d['config']['minibatch_feed']
    assert d['variants'][0]['value'] > 0 and d['variants'][0]['finite']


def test_in_graph_exchange_two_buckets_one_rank_bit_identical(gpu):
    """tools/dp_one_rank_check.py: six one-graph iterations with the gradient exchange inside the graph (one rank over RCCL) --
    critic step in two buckets (generator step in one: the default), both in two (GGAN_GEN_TWO_BUCKETS=1), everything in one
    (GGAN_ONE_BUCKET=1), and no exchange at all -- end in bit-identical weights (the bucket boundaries only decide WHEN a gradient range
    is summed over the replicas).  And the ISSUE ORDER of the captured iteration (optim.EXCHANGE_LOG), which is what decides the overlap:
    a bucket's all-reduce is handed to the communicator's stream before the part of the backward pass it is meant to run under is issued,
    the update is issued after the waits for every bucket of its step (round-3 review: assert the structure, not only the sums)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums, logs = [], []
    cases = (dict(GGAN_FORCE_ALLREDUCE='1'), dict(GGAN_FORCE_ALLREDUCE='1', GGAN_GEN_TWO_BUCKETS='1'),
             dict(GGAN_FORCE_ALLREDUCE='1', GGAN_ONE_BUCKET='1'), dict())
    for i, extra in enumerate(cases):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **extra)
        r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                            '127.0.0.1', '--master-port', str(29300 + (os.getpid() % 200) + i), os.path.join(root, 'tools', 'dp_one_rank_check.py'),
                            'ali', '0'], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        line = [l for l in r.stdout.splitlines() if l.startswith('CHECK')]
        assert r.returncode == 0 and line, (r.stdout[-500:], r.stderr[-2000:])
        assert ('dp_graph=True' in line[0]) == bool(extra) and 'one_graph=True' in line[0], line[0]
        sums.append(line[0].split()[-1])
        logs.append([tuple(e) for e in json.loads([l for l in r.stdout.splitlines() if l.startswith('XLOG')][0][5:])])
    assert sums[0] == sums[1] == sums[2] == sums[3], sums

    def steps_of(log):
        """the captured iteration = the last two steps of the log (generator step, critic step), each ending with its update"""
        ends = [i for i, e in enumerate(log) if e[0] == 'update']
        assert len(ends) >= 2
        a = ends[-3] + 1 if len(ends) >= 3 else 0
        return log[a:ends[-2] + 1], log[ends[-2] + 1:ends[-1] + 1]

    def kinds(step):
        return [e[0] if e[0] != 'exchange' else ('exchange_async' if e[3] else 'exchange') for e in step]
    one = ['pack', 'exchange', 'update']
    two = lambda what: ['pack', 'exchange_async', 'backward', 'pack', 'exchange_async', 'wait', 'wait', 'update']
    # default: generator step one bucket, critic step two (tail first: it is on the wire while the conv stack's backward pass is issued)
    g, d = steps_of(logs[0])
    assert kinds(g) == one and kinds(d) == two('critic'), (g, d)
    assert d[2] == ('backward', 'critic conv stack')
    (_, lo_a, hi_a, _), (_, lo_b, hi_b, _) = d[1], d[4]
    assert lo_b == 0 and hi_b == lo_a and hi_a > lo_a > 0               # tail bucket [off, total) first, then the conv stack's [0, off)
    assert d[0][1] > 0 and d[3][1] == 0 and d[3][2] == d[0][1]          # parameters [k, n) packed before [0, k)
    # both steps in two buckets: the Generator's bucket goes out before the Extractor's backward pass is issued
    g, d = steps_of(logs[1])
    assert kinds(g) == two('gen') and g[2] == ('backward', 'Extractor') and kinds(d) == two('critic'), (g, d)
    assert g[1][1] == 0 and g[4][1] == g[1][2] and g[4][2] > g[4][1]
    # one bucket everywhere / no exchange at all
    g, d = steps_of(logs[2])
    assert kinds(g) == one and kinds(d) == one
    g, d = steps_of(logs[3])
    assert kinds(g) == ['pack', 'update'] and kinds(d) == ['pack', 'update']


def test_direct_rccl_communicator_eager_and_captured(gpu):
    """graphical_gan_amd/rccl.py at one rank: all-reduce and all-gather issued eagerly and from a replayed HIP graph, on the
    communicator's stream and on a caller's stream, right after an eager torch.distributed collective -- the situation in which a
    captured process-group collective used to kill the process (no pauses anywhere here)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        from graphical_gan_amd import rccl
        comm = rccl.get(dev)
        t = torch.arange(1 << 16, dtype=torch.float32, device=dev)
        dist.all_reduce(t.clone())                                   # eager process-group work just before the captures
        ref = t.clone()
        comm.all_reduce_(t); torch.cuda.synchronize(); assert torch.equal(t, ref)
        s = torch.cuda.Stream(device=dev); s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                u = t * 2.0
                w = comm.all_reduce_(u, async_op=True)              # parallel branch on the communicator's stream
                v = t + 1.0
                w.wait()
                out = torch.empty((1, 16), dtype=torch.float32, device=dev)
                comm.all_gather(out, u[:16].contiguous())            # on the capturing stream itself
                z = u + v
            for _ in range(3):
                g.replay()
        torch.cuda.synchronize()
        assert torch.equal(z, 3 * ref + 1) and torch.equal(out[0], 2 * ref[:16])
        import time; time.sleep(1.5)                                 # (a watchdog poll interval: nothing dies afterwards either)
        print('RCCL_DIRECT_OK')
        dist.destroy_process_group()
    ''' % root)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29250 + os.getpid() % 200))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and 'RCCL_DIRECT_OK' in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_cross_replica_batchnorm_inside_the_step_graph_one_rank_rehearsal(gpu):
    """Trainer(sync_bn=True) with the directly bound communicator: the statistics exchange (ncclAllGather) is an enqueue on the
    step's stream, so the steps run from HIP graphs (one graph per iteration) -- same weights, bit for bit, as the eager steps."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = []
    for i, extra in enumerate((dict(), dict(CHECK_EAGER='1'))):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GGAN_FORCE_ALLREDUCE='1', CHECK_SYNC_BN='1', **extra)
        r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                            '127.0.0.1', '--master-port', str(29100 + (os.getpid() % 200) + i), os.path.join(root, 'tools', 'dp_one_rank_check.py'),
                            'ali', '0'], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        line = [l for l in r.stdout.splitlines() if l.startswith('CHECK')]
        assert r.returncode == 0 and line, (r.stdout[-500:], r.stderr[-2000:])
        assert 'sync_bn=True' in line[0] and ('one_graph=True' in line[0]) == (not extra), line[0]
        sums.append(line[0].split()[-1])
    assert sums[0] == sums[1], sums


@pytest.mark.parametrize('mode,K', [('ali', 0), ('local_ep', 30)])
def test_full_size_training_is_bitwise_reproducible(gpu, mode, K):
    """BASELINE-size step (batch 64, HIP-graph replay, two-stream nets pass for ali, on-device noise): two runs from the same
    seeds end in bit-identical weights -- no atomics, no order-dependent reductions anywhere on the path, and the parallel graph
    branches only reorder independent kernels."""
    import torch
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.engine import Trainer
    finals = []
    for run in range(2):
        _fresh()
        np.random.seed(0)
        cfg = Config('cifar10', batch_size=64, n_coms=K, mode=mode)
        tr = Trainer(cfg, device=gpu, graph=True, seed=4321)
        ring = tr.model.synthetic_ring(gpu, n=4, seed=99)
        batches = iter(ring * 8)
        for it in range(6):
            res = tr.iteration(it, batches)
        tr.flush()
        torch.cuda.synchronize()
        assert all(np.isfinite(float(v)) for v in res.values())
        finals.append({k: v.copy() for k, v in tr.get_params().items()})
    for k in finals[0]:
        assert np.array_equal(finals[0][k], finals[1][k]), k


def test_vegan_mmd_mode_generator_only_steps(gpu):
    """MODE vegan-mmd (gan_inference_cifar10.py:327-329, CRITIC_ITERS = 0): no critic -- cost, every Generator / Extractor gradient
    and a 4-iteration trajectory (generator steps only, from iteration 1) against the oracle."""
    import torch
    from oracle import step as S, tape as tp
    ocfg, P0, cfg, tr = _mk('cifar10', 8, 0, 'vegan-mmd', 8, 16, True, True, gpu)
    assert cfg.critic_iters == 0
    feed = S.make_feed(ocfg, np.random.default_rng(11), 'vegan-mmd')
    Pt = {k: tp.T(v.astype(np.float64)) for k, v in P0.items()}
    oout = S.forward(ocfg, Pt, feed, 'vegan-mmd')
    tr.set_feed(feed)
    out = tr.model.forward(tr.feed, 'gen')
    assert out['disc_cost'] is None
    oc, c = float(oout['gen_cost'].v), float(out['gen_cost'].detach())
    assert abs(c - oc) <= 1e-5 * max(1.0, abs(oc)), (c, oc)
    opt = out['gen_train_op'].optimizer
    names = [p.param_name for p in opt.params]
    grads = torch.autograd.grad(out['gen_cost'], opt.params, allow_unused=True)
    ogs = tp.grad(oout['gen_cost'], [Pt[n] for n in names])
    gmax = max(np.abs(og.v).max() for og in ogs if og is not None)
    for n, g, og in zip(names, grads, ogs):
        if og is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        err = np.abs(g.cpu().numpy().reshape(og.v.shape) - og.v).max()
        assert err <= 1e-4 * max(np.abs(og.v).max(), 1e-2 * gmax), (n, err)
    # (a fresh Trainer: the autograd graph built above on the default stream must not be alive while the step graph is captured)
    del out, grads, opt
    ocfg, P0, cfg, tr = _mk('cifar10', 8, 0, 'vegan-mmd', 8, 16, True, True, gpu)
    otr = S.Trainer(ocfg, P0, 'vegan-mmd', np.float64)
    feeds = [S.make_feed(ocfg, np.random.default_rng(200 + i), 'vegan-mmd') for i in range(6)]
    fo, fp = iter(feeds), iter(feeds)
    for it in range(4):
        ro, rp = otr.iteration(it, fo), tr.iteration(it, fp)
        assert set(ro) == set(rp) == ({'gen_cost'} if it > 0 else set())
        for k in ro:
            assert abs(float(rp[k]) - ro[k]) <= 2e-3 * max(1.0, abs(ro[k])), (it, k, float(rp[k]), ro[k])


@pytest.mark.parametrize('mode,z_samples,bn', [('vegan-kl', None, False), ('vegan-ikl', 13, False), ('vegan-jsd', None, False), ('vegan-jsd', 7, True)])
def test_vegan_aggregated_divergence_modes(gpu, mode, z_samples, bn):
    """MODE vegan-kl / vegan-ikl / vegan-jsd (gan_inference_cifar10.py:331-341, tflib/objs/kl_aggregated.py; CRITIC_ITERS = 0, stochastic
    encoder TYPE_Q = 'learn_std'): cost, the divergence alone, every Generator / Extractor gradient and a 4-iteration trajectory
    (generator steps only) against the oracle, same injected draws."""
    import torch
    from oracle import step as S, tape as tp
    mk = lambda: _mk('cifar10', 8, 0, mode, 8, 8, True, True, gpu, z_samples=z_samples, bn=bn)
    ocfg, P0, cfg, tr = mk()
    assert cfg.critic_iters == 0 and cfg.learn_std and 'Extractor.Std.W' in P0
    feed = S.make_feed(ocfg, np.random.default_rng(11), mode)
    Pt = {k: tp.T(v.astype(np.float64)) for k, v in P0.items()}
    oout = S.forward(ocfg, Pt, feed, mode)
    tr.set_feed(feed)
    out = tr.model.forward(tr.feed, 'gen')
    assert out['disc_cost'] is None
    for k in ('q_z', 'q_z_mean', 'q_z_std'):
        assert np.abs(out[k].detach().cpu().numpy() - oout[k].v).max() <= 1e-5 * max(1.0, np.abs(oout[k].v).max()), k
    oc, c = float(oout['gen_cost'].v), float(out['gen_cost'].detach())
    od, d = float(oout['divergence'].v), c - float(out['rec_penalty'].detach())
    assert abs(c - oc) <= 1e-5 * max(1.0, abs(oc)) and abs(d - od) <= 2e-5 * max(1.0, abs(od)), (c, oc, d, od)
    opt = out['gen_train_op'].optimizer
    names = [p.param_name for p in opt.params]
    assert 'Extractor.Std.W' in names and not any(n.startswith('Discriminator') for n in names)
    grads = torch.autograd.grad(out['gen_cost'], opt.params, allow_unused=True)
    ogs = tp.grad(oout['gen_cost'], [Pt[n] for n in names])
    gmax = max(np.abs(og.v).max() for og in ogs if og is not None)
    for n, g, og in zip(names, grads, ogs):
        if og is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        err = np.abs(g.cpu().numpy().reshape(og.v.shape) - og.v).max()
        assert err <= 1e-4 * max(np.abs(og.v).max(), 1e-2 * gmax), (n, err)
    del out, grads, opt
    ocfg, P0, cfg, tr = mk()
    otr = S.Trainer(ocfg, P0, mode, np.float64)
    feeds = [S.make_feed(ocfg, np.random.default_rng(300 + i), mode) for i in range(6)]
    fo, fp = iter(feeds), iter(feeds)
    for it in range(4):
        ro, rp = otr.iteration(it, fo), tr.iteration(it, fp)
        assert set(ro) == set(rp) == ({'gen_cost'} if it > 0 else set())
        for k in ro:
            assert abs(float(rp[k]) - ro[k]) <= 2e-3 * max(1.0, abs(ro[k])), (it, k, float(rp[k]), ro[k])
    P = tr.get_params()
    for n, v in otr.P.items():
        if n.startswith('Discriminator'):
            continue
        assert np.abs(P[n].astype(np.float64).reshape(v.shape) - v).max() <= 2e-3 * max(1.0, np.abs(v).max()), n


@pytest.mark.parametrize('mode,crit', [('ali', 1), ('wali-gp', 5)])
def test_ring_feed_one_graph_per_iteration_matches_staging_buffer(gpu, mode, crit):
    """Trainer.use_ring: the steps read their minibatch in place from the device-resident ring (slot = the optimizers' step
    counts) and a whole iteration is ONE graph replay.  Same seeds, same minibatch order => bit-identical weights to the path
    that copies every minibatch into the staging buffer and replays one graph per step."""
    import torch
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.engine import Trainer
    finals = []
    for use_ring in (False, True, 'host'):
        _fresh()
        np.random.seed(0)
        cfg = Config('cifar10', batch_size=16, n_coms=0, mode=mode, dim=16, dim_latent=32)
        assert cfg.critic_iters == crit
        tr = Trainer(cfg, device=gpu, graph=True, seed=4321)
        ring = tr.model.synthetic_ring(gpu, n=5, seed=99)
        batches = iter(ring * 40)
        taken = 0
        for it in range(2):                       # eager: parameters + optimizers
            tr.iteration(it, batches)
            taken += (1 if it > 0 else 0) + crit
        k = taken % len(ring)
        if use_ring == 'host':
            # Trainer.use_host_ring: the same minibatches from HOST memory, copied into the ring's slots one iteration ahead
            host = [b.cpu().numpy() for b in ring[k:] + ring[:k]]
            tr.use_host_ring(lambda: iter(host))
        elif use_ring:
            tr.use_ring(ring[k:] + ring[:k])      # the ring continues where the iterator stands
        for it in range(2, 7):
            res = tr.iteration(it, batches)
        if use_ring:
            assert tr._iter_graph is not None and not tr._graphs, 'one graph per iteration expected'
            # (round 5: with CRITIC_ITERS > 1 the nets passes of critic steps 2.. run ahead of time on a stream of their own inside that
            #  graph -- per-step feeds, ring slots from a snapshot of the critic's step count, noise launches in step order)
            assert (getattr(tr, '_ahead', None) is not None) == (crit > 1)
        tr.flush()
        torch.cuda.synchronize()
        assert all(np.isfinite(float(v)) for v in res.values())
        finals.append(({k: v.copy() for k, v in tr.get_params().items()}, {k: float(v) for k, v in res.items()}))
    assert finals[0][1] == finals[1][1] == finals[2][1]
    for k in finals[0][0]:
        assert np.array_equal(finals[0][0][k], finals[1][0][k]) and np.array_equal(finals[0][0][k], finals[2][0][k]), k


@pytest.mark.parametrize('mode,crit', [('ali', 1), ('wali-gp', 5)])
def test_update_riding_in_the_pack_launch_is_bit_identical(gpu, monkeypatch, mode, crit):
    """ggan_pack_adam (gradient pack + Adam update in one launch, single-replica steps) against ggan_pack_parts2 followed by
    ggan_adam_step_counted: same sums, same update arithmetic, the step counter advanced once per step either way => bit-identical
    weights, Adam moments and step counts after eager and graph-replayed steps (wali-gp: two gradient contributions per critic
    parameter, five critic steps per iteration)."""
    import torch
    from graphical_gan_amd import optim
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.engine import Trainer
    finals = []
    for fused in (False, True):
        if fused:
            monkeypatch.delenv('GGAN_NO_PACK_ADAM', raising=False)
        else:
            monkeypatch.setenv('GGAN_NO_PACK_ADAM', '1')
        _fresh()
        np.random.seed(0)
        cfg = Config('cifar10', batch_size=16, n_coms=0, mode=mode, dim=16, dim_latent=32)
        tr = Trainer(cfg, device=gpu, graph=True, seed=4321)
        batches = iter(tr.model.synthetic_ring(gpu, n=5, seed=99) * 40)
        for it in range(6):
            res = tr.iteration(it, batches)
        tr.flush()
        torch.cuda.synchronize()
        opts = tr._optimizers()
        assert all(o.can_fuse_update() == fused for o in opts if type(o) is optim.AdamOptimizer)
        assert all(int(o._arrive.sum()) == 0 for o in opts if o._arrive is not None), 'arrival counters must be left at zero'
        finals.append(({k: v.copy() for k, v in tr.get_params().items()}, {k: float(v) for k, v in res.items()},
                       [(o.m.cpu().numpy(), o.v.cpu().numpy(), int(o.step)) for o in opts]))
    assert finals[0][1] == finals[1][1]
    for k in finals[0][0]:
        assert np.array_equal(finals[0][0][k], finals[1][0][k]), k
    for (m0, v0, s0), (m1, v1, s1) in zip(finals[0][2], finals[1][2]):
        assert s0 == s1 and s0 > 0 and np.array_equal(m0, m1) and np.array_equal(v0, v1)


@pytest.mark.parametrize('dataset,mode', [('cifar10', 'ali'), ('face', 'ali'), ('cifar10', 'local_ep')])
def test_cost_launch_carrying_the_head_backward_is_bit_identical(gpu, monkeypatch, dataset, mode):
    """ggan_bce_head_bwd (BCE cost + unit-seed gradients + the head kernel of the critic tail's backward in one launch, where the
    cost's logits are one critic head's output) against ggan_bce_logits_multi_fwd_grad followed by the head kernel of
    ggan_critic_head_bwd: the same arithmetic in the same order => bit-identical costs and weights after eager and graph-replayed
    steps.  The mixture scripts' cost takes the logits of two heads (joint critic, mixture critic): both head kernels ride along."""
    import torch
    from graphical_gan_amd import functional as F, _lib
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.engine import Trainer
    finals, fused_launches = [], []
    monkeypatch.setenv('GGAN_NO_HEAD_HINT', '1')      # (MODE ali otherwise takes the hinted head: the test below)
    for fused in (False, True):
        if fused:
            monkeypatch.delenv('GGAN_NO_HEAD_BCE', raising=False)
        else:
            monkeypatch.setenv('GGAN_NO_HEAD_BCE', '1')
        _fresh()
        np.random.seed(0)
        cfg = Config(dataset, batch_size=16, n_coms=10 if mode == 'local_ep' else 0, mode=mode, dim=16, dim_latent=32)
        tr = Trainer(cfg, device=gpu, graph=True, seed=4321)
        batches = iter(tr.model.synthetic_ring(gpu, n=5, seed=99) * 40)
        lib_, calls = _lib.load(), []
        entry = lib_.ggan_bce_heads_bwd

        def counted(*args):
            calls.append(int(args[7]))             # number of heads carried
            return entry(*args)
        monkeypatch.setattr(lib_, 'ggan_bce_heads_bwd', counted)
        for it in range(6):
            res = tr.iteration(it, batches)
        monkeypatch.setattr(lib_, 'ggan_bce_heads_bwd', entry)
        fused_launches.append(max(calls) if calls else 0)
        tr.flush()
        torch.cuda.synchronize()
        assert not F.HEAD_LOGITS or all(r['h']() is None for r in F.HEAD_LOGITS.values()), 'no record may keep a step alive'
        finals.append(({k: v.copy() for k, v in tr.get_params().items()}, {k: float(v) for k, v in res.items()}))
    assert fused_launches == [0, 1 if mode == 'ali' else 2]      # (the mixture scripts: joint critic + mixture critic in one cost)
    assert finals[0][1] == finals[1][1]
    for k in finals[0][0]:
        assert np.array_equal(finals[0][0][k], finals[1][0][k]), k


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('dataset,mode', [('cifar10', 'ali'), ('face', 'ali'), ('cifar10', 'wali-gp'), ('cifar10', 'local_ep')])
def test_head_that_knows_its_cost_is_bit_identical(gpu, monkeypatch, dataset, mode, graph):
    """MODE ali / wali-gp on the batched critic: the step's cost is known before the critic runs (functional.head_bce_hint, set by
    models.forward for engine.Trainer's steps; sigmoid cross-entropy terms, or the Wasserstein means with the gradient penalty as a
    one-element term that only enters the cost's value), so the head's forward tail leaves g = d cost / d logits and gh behind
    (ggan_critic_head_fwd_bce) and the products' launch of its backward carries the cost, d_wout and d_bout
    (ggan_critic_head_bwd_tail) -- one launch less per step than tail + cost/head launch + products.  Same expressions in the same
    order => bit-identical costs and weights, eager and graph-replayed."""
    import torch
    from graphical_gan_amd import functional as F, _lib
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.engine import Trainer
    finals, counts = [], []
    for hinted in (False, True):
        if hinted:
            monkeypatch.delenv('GGAN_NO_HEAD_HINT', raising=False)
        else:
            monkeypatch.setenv('GGAN_NO_HEAD_HINT', '1')
        _fresh()
        np.random.seed(0)
        cfg = Config(dataset, batch_size=16, n_coms=10 if mode == 'local_ep' else 0, mode=mode, dim=16, dim_latent=32)
        tr = Trainer(cfg, device=gpu, graph=graph, seed=4321)
        batches = iter(tr.model.synthetic_ring(gpu, n=5, seed=99) * 40)
        lib_, calls = _lib.load(), {'fwd': 0, 'bwd': 0, 'old': 0}
        e_fwd, e_bwd, e_old = lib_.ggan_critic_head_fwd_bce, lib_.ggan_critic_head_bwd_tail, lib_.ggan_bce_heads_bwd

        def c_fwd(*a):
            calls['fwd'] += 1
            return e_fwd(*a)

        def c_bwd(*a):
            calls['bwd'] += 1
            return e_bwd(*a)

        def c_old(*a):
            calls['old'] += 1
            return e_old(*a)
        monkeypatch.setattr(lib_, 'ggan_critic_head_fwd_bce', c_fwd)
        monkeypatch.setattr(lib_, 'ggan_critic_head_bwd_tail', c_bwd)
        monkeypatch.setattr(lib_, 'ggan_bce_heads_bwd', c_old)
        costs = []
        for it in range(6):
            res = tr.iteration(it, batches)
            if not graph:
                costs.append({k: float(v) for k, v in res.items()})
        monkeypatch.setattr(lib_, 'ggan_critic_head_fwd_bce', e_fwd)
        monkeypatch.setattr(lib_, 'ggan_critic_head_bwd_tail', e_bwd)
        monkeypatch.setattr(lib_, 'ggan_bce_heads_bwd', e_old)
        tr.flush()
        torch.cuda.synchronize()
        counts.append(dict(calls))
        finals.append(({k: v.copy() for k, v in tr.get_params().items()}, {k: float(v) for k, v in res.items()}, costs))
    assert counts[0]['fwd'] == 0 and counts[0]['bwd'] == 0 and (counts[0]['old'] > 0 or mode == 'wali-gp')     # (Wasserstein costs: ggan_mean_multi_fwd_grad)
    assert counts[1]['fwd'] > 0 and counts[1]['bwd'] == counts[1]['fwd'] and counts[1]['old'] == 0
    assert finals[0][1] == finals[1][1] and finals[0][2] == finals[1][2]
    assert all(np.isfinite(v) for v in finals[1][1].values())
    for k in finals[0][0]:
        assert np.array_equal(finals[0][0][k], finals[1][0][k]), k


@pytest.mark.gpu
@pytest.mark.parametrize('graph', [False, True])
def test_penalty_value_joining_the_cost_late_is_bit_identical(gpu, monkeypatch, graph):
    """wali-gp critic steps (round 6): the critic cost's VALUE needs the gradient penalty, its backward pass does not.  The head that writes the
    cost with its backward launch leaves the one-element penalty term out, the main stream is not joined with the penalty stream in front
    of the cost, and the Trainer adds the term behind the backward pass (functional.LATE_EXT / add_late_terms: one ggan_axpby launch on the
    penalty's stream per critic step).  The sum is the same two terms in the same order => bit-identical costs and weights against
    GGAN_NO_LATE_PENALTY=1, eager and graph-replayed; and the late launches are really issued."""
    import torch
    from graphical_gan_amd import functional as F, _lib
    from graphical_gan_amd.models import Config
    from graphical_gan_amd.engine import Trainer
    finals, lates = [], []
    for late in (False, True):
        if late:
            monkeypatch.delenv('GGAN_NO_LATE_PENALTY', raising=False)
        else:
            monkeypatch.setenv('GGAN_NO_LATE_PENALTY', '1')
        _fresh()
        np.random.seed(0)
        cfg = Config('cifar10', batch_size=16, n_coms=0, mode='wali-gp', dim=16, dim_latent=32)
        tr = Trainer(cfg, device=gpu, graph=graph, seed=4321)
        batches = iter(tr.model.synthetic_ring(gpu, n=5, seed=99) * 60)
        n_late = [0]
        entry = F.add_late_terms

        def counted():
            evs = entry()
            n_late[0] += len(evs)
            return evs
        monkeypatch.setattr(F, 'add_late_terms', counted)
        costs = []
        for it in range(4):
            res = tr.iteration(it, batches)
            if not graph:
                costs.append({k: float(v) for k, v in res.items()})
        monkeypatch.setattr(F, 'add_late_terms', entry)
        tr.flush()
        torch.cuda.synchronize()
        assert not F.pending_costs() and not F._LATE_TERMS
        lates.append(n_late[0])
        finals.append(({k: v.copy() for k, v in tr.get_params().items()}, {k: float(v) for k, v in res.items()}, costs))
    # (eager steps do not fork the penalty pass onto a stream of its own: nothing is late there, and the two runs must still agree)
    assert lates[0] == 0 and (lates[1] > 0) == bool(graph), lates
    assert finals[0][1] == finals[1][1] and finals[0][2] == finals[1][2]
    assert all(np.isfinite(v) for v in finals[1][1].values())
    for k in finals[0][0]:
        assert np.array_equal(finals[0][0][k], finals[1][0][k]), k
