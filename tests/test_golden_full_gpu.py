"""-m gpu: the HIP path at BASELINE sizes against the committed full-size fixtures (tests/golden/full_*.npz: float64 first-step
costs, critic logits and per-parameter gradient digests of every BASELINE configuration, generated offline by
tests/golden/make_golden_full.py).  Nothing of the oracle's arithmetic runs here: only its deterministic initial-weight and
feed generators, checked against the checksums in the fixture.  The fixtures' feeds keep every LeakyReLU input of the critics'
MLP layers clear of its kink (margin stored in the fixture); the conv-layer units that float32 rounding can put on the other side of
their kink are IDENTIFIED (kink tables, below) and the gradients gated at 5e-5 against the float64 evaluation with exactly those units
on the GPU's branch -- for every fixture (the image scripts' and the state-space script's), eager and timed."""
import os
import sys

import numpy as np
import pytest

from _golden import load

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))


OBSERVED = {}        # (fixture, configuration, step) -> (worst entry error / scale, worst L2 error / norm), filled by _check_grads

# ---- kink flips (round 6) ---------------------------------------------------------------------------------------------------------
# A fixture holds, for every image-shaped ReLU / LeakyReLU activation of the step, the 48 units nearest their kink (flat positions,
# float64 pre-activations, the layer's rms).  The GPU's outputs at those positions say on which side of zero ITS float32 evaluation
# put each of them; a unit on the other side than float64 is a "flip".  A flip is accepted only if float32 rounding can explain it
# (|float64 pre-activation| <= KINK_BOUND x the layer's rms), and the table must reach beyond that bound (else a flip could hide
# outside it).  The step's gradients are then gated at TOL_KINK against the float64 evaluation with EXACTLY those units forced onto
# the GPU's branch (fixture: <step>/flip<i>/...; the plain digests when there is no flip) -- a bound, not a regression pin.
KINK_BOUND = 2e-6        # (worst flipped unit observed on MI355X, all fixtures, eager and timed: 6.7e-7 of its layer's rms)
TOL_KINK = 5e-5
FLIP_REPORT = os.environ.get('GGAN_FLIP_REPORT')       # discovery mode: append every observed flip set to this JSON-lines file


def _gpu_flips(z, which, taps, B, last, table=None):
    """taps: [(layer name, output tensor)] of this step in call order; last: take the LAST call of a layer (a critic step of an iteration
    graph: the step whose gradients the bucket holds) instead of the first; table: the fixture prefix the kink table sits under (default:
    the step's own; the state-space fixture has ONE table, 'fwd', for both steps, with the row count of every activation stored).
    -> sorted [[kink key, flat position], ...], worst |pre| / rms"""
    table = table or which
    by = {}
    for name, t in taps:
        by.setdefault((name, int(t.shape[0])), []).append(t)
    pick = (lambda v: v[-1]) if last else (lambda v: v[0])
    flips, worst = [], 0.0
    keys = sorted(set(k.split('/')[2] for k in z.files if k.startswith(table + '/kink/') and k.endswith('/idx')))
    assert keys, 'fixture without a kink table'
    for key in keys:
        layer, _, tag = key.partition('@')
        idx = z['%s/kink/%s/idx' % (table, key)].astype(np.int64)
        val = z['%s/kink/%s/val' % (table, key)]
        rms = float(z['%s/kink/%s/rms' % (table, key)])
        rows = int(z['%s/kink/%s/rows' % (table, key)]) if ('%s/kink/%s/rows' % (table, key)) in z.files else B
        if tag in ('fake', 'real') and (layer, 2 * rows) in by:       # the critic evaluated once on [fake; real]
            t = pick(by[(layer, 2 * rows)])
            off = 0 if tag == 'fake' else t.numel() // 2
        elif tag in ('fake', 'real'):                                 # (BatchNorm inside the critic: two calls, fake then real)
            calls = by.get((layer, rows), [])
            assert len(calls) >= 2, ('no critic taps for', key, sorted(by))
            t, off = (calls[-2:] if last else calls[:2])[0 if tag == 'fake' else 1], 0
        else:
            assert (layer, rows) in by, ('no tap for', key, sorted(by))
            t, off = pick(by[(layer, rows)]), 0
        y = t.reshape(-1)[(idx + off).tolist()].cpu().numpy()
        gpu_pos, ref_pos = y > 0, val > 0
        assert np.abs(val).max() >= KINK_BOUND * rms, (key, 'kink table does not reach the rounding bound', np.abs(val).max() / rms)
        for i in np.where(gpu_pos != ref_pos)[0]:
            r = abs(val[i]) / rms
            assert r <= KINK_BOUND, (which, key, int(idx[i]), 'sign differs from float64 at a unit %.3g rms from its kink: not a rounding flip' % r)
            worst = max(worst, r)
            flips.append([key, int(idx[i])])
    return sorted(flips), worst


def _grad_reference(z, name, config, which, flips, worst, table=None):
    """which digests the step's gradients are gated against: -> (prefix in the fixture, tolerance)"""
    import json
    sets = json.loads(str(z[(table or which) + '/flipsets']))
    if FLIP_REPORT:
        with open(FLIP_REPORT, 'a') as f:
            f.write(json.dumps(dict(fixture=name, config=config, step=which, flips=flips, worst_rms=worst, known=(not flips) or flips in sets)) + '\n')
    if not flips:
        return which, TOL_KINK
    if flips in sets:
        return '%s/flip%d' % (which, sets.index(flips)), TOL_KINK
    if FLIP_REPORT:                      # discovery: the set is not in the fixture yet -- fall back to the observed-error table
        return which, None
    raise AssertionError('%s/%s/%s: the GPU flipped %s, a set the fixture has no float64 variant for (known: %s): run the test with '
                         'GGAN_FLIP_REPORT=<file>, merge it into tests/golden/full_flips.json (tools/merge_flips.py) and regenerate the fixture'
                         % (name, config, which, flips, sets))


def _check_grads(z, which, names, grads, tol, tag=None):
    """per tensor: the 64 fixture entries within tol * scale and the L2 norm within tol/3 (scale = max |g| of the tensor, floored at
    1e-2 of the largest gradient of the step for the mathematically-zero ones, e.g. a bias that feeds a BatchNorm).  Every failure
    message carries the observed error; the worst observed ratios of a passing check go to OBSERVED[tag] (and to stdout with
    GGAN_TEST_REPORT=1: that is where the table OBS below comes from)."""
    import make_golden_full as MG
    refs = {n: z['%s/g/%s' % (which, n)] for n in names if '%s/g/%s' % (which, n) in z.files}
    gmax = max(r[1] for r in refs.values())
    worst_e, worst_l, n_all, n_over = 0.0, 0.0, 0, [0, 0, 0]
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
        assert err <= tol * scale, (which, n, 'entries: observed %.3g of the scale, tolerance %.3g' % (err / scale, tol))
        l2den = max(ref[0], 1e-2 * gmax * np.sqrt(f.size))
        l2 = abs(np.linalg.norm(f) - ref[0])
        assert l2 <= tol / 3 * l2den, (which, n, 'l2: observed %.3g relative, tolerance %.3g' % (l2 / l2den, tol / 3))
        worst_e, worst_l = max(worst_e, err / scale), max(worst_l, l2 / l2den)
        n_all += e.size
        for k_, th in enumerate((1e-5, 5e-5, 2e-4)):
            n_over[k_] += int((e > th * scale).sum())
    if tag is not None:
        OBSERVED[tag + (which,)] = (worst_e, worst_l)
        if os.environ.get('GGAN_TEST_REPORT'):
            print('OBSERVED %s entries %.3g l2 %.3g (tolerance %.3g); of %d sampled entries %d / %d / %d above 1e-5 / 5e-5 / 2e-4 of the scale'
                  % ('/'.join(tag + (which,)), worst_e, worst_l, tol, n_all, n_over[0], n_over[1], n_over[2]))


# Round 5 gated every fixture at 3 x the error the same implementation had shown (up to 2e-3): the errors were bimodal -- 1e-6 .. 5e-5 for a
# step whose ~7e6 ReLU / LeakyReLU units all land on the float64 side of their kinks, 1e-5 .. 1.6e-3 for a step with a flipped conv-layer
# unit -- and the explanation was asserted, not shown.  Round 6 shows it: every (fixture, configuration, step) with an error above 5e-5
# has 1-7 flipped units, each within 7e-7 of its layer's rms of zero, and against the float64 evaluation with exactly those units forced
# the worst error of ANY fixture is 2.9e-5 (profiles/r06_notes.md; full_cifar_gmgan_k30: 1.6e-3 -> 2.9e-5 eager, no flip and 1.4e-5
# timed; the state-space fixture: 3.2e-4 -> 8.8e-7 with its 7 flips forced).  The observed-error table is what a run in discovery mode
# (GGAN_FLIP_REPORT, a flip set the fixture does not hold yet) falls back to:
CEIL = {}
OBS = {'full_ssgan_b32_t16': (3.2e-4, 2.4e-5)}


def tol_of(name, which):
    return min(CEIL.get(name, 1e-3), max(2e-5, 3.0 * OBS[name][0 if which == 'gen' else 1]))


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
        lib.TAPS[0] = []
        out = tr.model.forward(tr.feed, which)
        taps, lib.TAPS[0] = [(n_, t_) for _, n_, t_ in lib.TAPS[0]], None
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
        # (the gradient gate: TOL_KINK = 5e-5 of the tensor's max |g| per entry, a third of it on its L2 norm, against the float64 digests of
        #  the flip set this run shows -- see the kink section at the top; costs and logits stay at 1e-5 / 2e-5)
        flips, worst = _gpu_flips(z, which, taps, B, last=False)
        prefix, tol = _grad_reference(z, name, 'eager', which, flips, worst)
        _check_grads(z, prefix, names, grads, tol if tol is not None else 2e-3, tag=(name, 'eager'))
    optim.reset_optimizers()
    lib.delete_all_params()


@pytest.mark.parametrize('name', ['full_cifar_ali', 'full_face_ali', 'full_cifar_gmgan_k30', 'full_cifar_wali_gp', 'full_cifar_gmgan_k10',
                                  'full_face_gmgan_k100', 'full_mnist_ali', 'full_mnist_gmgan'])
def test_full_size_timed_configuration_vs_fixture(gpu, name):
    """The configuration bench.py TIMES, against the float64 fixture (round-3 review: the full-size fixtures ran graph=False on the default
    launch plan, the graph path was only compared with itself; round-4 review: every fixture, not four of them): the steps captured as
    HIP graphs with the Generator / Extractor passes forked onto two streams and their conv launches planned for 128 workgroups
    (models.launch_hint / functional.target_workgroups), gradients packed and the Adam update applied by the pack launch; for the int32
    image scripts the minibatches are read in place from the device ring and generator step + critic step(s) are ONE graph, the MNIST
    scripts (float minibatches) go through the staging buffer and one graph per step -- exactly what bench.py runs for each.  The
    noise launch is replaced by the fixture's draws (inject_noise) and the learning rates are zero, so that every step of the replayed
    graph is evaluated at the fixture's weights: costs, critic logits, and every parameter's gradient as the pack launch left it in
    the optimizer's flat bucket."""
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
    tr.flush()
    tr._graphs, tr._iter_graph = {}, None    # (the critic step's second call captured its graph with the learning rate it had then)
    ring = dataset != 'mnist'
    if ring:
        x = torch.as_tensor(np.ascontiguousarray(feed['real_x_int'])).to(gpu)
        tr.use_ring([x] * 4)                 # every ring slot holds the fixture's minibatch
    res = None
    lib.TAPS[0] = []
    for it in (2, 3, 4):                     # capture (with its warm-up steps), then replays
        res = tr.iteration(it, None if ring else feeds)
    all_taps, lib.TAPS[0] = lib.TAPS[0], None     # (the captured graphs' activations: static graph memory, rewritten by every replay)
    if os.environ.get('GGAN_EXPECT_DP_GRAPH'):           # (test_full_size_dp_schedule_vs_fixture: this process runs the data-parallel schedule)
        assert tr.dp_graph and tr.comm is not None
        if cfg.critic_iters > 1:
            assert getattr(tr, '_ahead', None) is not None and len(tr._ahead['feeds']) == cfg.critic_iters - 1
    if ring:
        assert getattr(tr, '_iter_graph', None) is not None and tr._iter_graph['kinds'] == ('gen',) + ('disc',) * cfg.critic_iters
        assert tr.model.fork_nets            # (the two-stream nets pass and its 128-workgroup plans were on while the graph was built)
    else:
        assert set(tr._graphs) == {'gen', 'disc'}        # (one captured graph per step kind, replayed by the last iterations)
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
        # a critic step's gradients in the bucket are those of the iteration's LAST critic step: its critic pass, and the nets pass
        # issued for it (ahead of time, one step earlier) -- the last call of each layer among the critic steps' scopes
        taps = [(n_, t_) for sc, n_, t_ in all_taps if sc is not None and sc.startswith(which)]
        flips, worst = _gpu_flips(z, which, taps, B, last=which == 'disc')
        prefix, tol = _grad_reference(z, name, 'timed', which, flips, worst)
        _check_grads(z, prefix, names, grads, tol if tol is not None else 2e-3, tag=(name, 'timed'))
    # the weights did not move (lr 0) and match the fixture's
    got = tr.get_params()
    for n_, v in P0.items():
        if not (n_.endswith('.moving_mean') or n_.endswith('.moving_variance')):
            assert np.array_equal(got[n_].reshape(-1), np.asarray(v, np.float32).reshape(-1)), n_
    optim.reset_optimizers()
    lib.delete_all_params()


def test_full_size_dp_schedule_vs_fixture(gpu):
    """The DATA-PARALLEL schedule at full size against the float64 fixtures (round-5 review: the schedule a replica of an N > 1 run executes
    was never the one a full-size fixture ran): the timed-configuration test above for BASELINE configs[2] (gmgan_inference_cifar10, K = 10)
    and for the headline (wali-gp: the critic steps' nets passes ahead of time INSIDE the data-parallel iteration graph) in a process
    with a one-rank RCCL communicator and GGAN_FORCE_ALLREDUCE=1 -- gradient exchange captured inside the iteration graph on the
    communicator's stream, pack -> all-reduce -> Adam instead of the update riding in the pack launch.  Same tolerances."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GGAN_FORCE_ALLREDUCE='1', GGAN_TEST_DP_ONE_RANK='1', GGAN_EXPECT_DP_GRAPH='1')
    env = {k: v for k, v in env.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_golden_full_gpu.py'), '-q', '-x', '-m', 'gpu', '-k',
                        'test_full_size_timed_configuration_vs_fixture and (full_cifar_gmgan_k10 or full_cifar_wali_gp)'],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert r.returncode == 0 and '2 passed' in r.stdout, (r.stdout[-3000:], r.stderr[-1500:])


@pytest.mark.parametrize('timed', [False, True], ids=['eager', 'timed'])
def test_full_size_ssgan_first_step_vs_fixture(gpu, timed):
    """ssgan_inference_moving_mnist.py at BASELINE configs[4]: 32 sequences x 16 frames of 64x64.  `timed`: the configuration bench.py
    times for the state-space scripts -- one captured HIP graph per step kind, minibatch through the staging buffer (an 8 MB device
    copy per 9 ms step; the int32 ring of the image scripts does not exist for float frames + labels), pack + Adam in one launch --
    with the fixture's noise and zero learning rates; gradients as the pack launch left them in the flat bucket."""
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
    tr = Trainer(cfg, device=gpu, graph=timed, inject_noise=True, model=StateSpaceGAN(cfg))
    tr.load_params(P0)
    tr.set_feed(feed)
    fx = tr.model.forward_nets(tr.feed)['fake_x'].detach().cpu().numpy().astype(np.float64).reshape(-1)
    ref = z['fake_x_digest']
    assert np.abs(fx[MG.sample_index('fake_x', fx.size)] - ref[2:]).max() <= 2e-5 and abs(np.linalg.norm(fx) - ref[0]) <= 1e-5 * ref[0]
    if timed:
        from graphical_gan_amd.optim import _optimizers
        feeds = iter([feed] * 64)
        tr.iteration(0, feeds)
        tr.iteration(1, feeds)
        for o in _optimizers.values():
            o.lr = 0.0
        tr.load_params(P0)
        tr.flush()
        tr._graphs, tr._iter_graph = {}, None
        res = None
        lib.TAPS[0] = []
        for it in (2, 3, 4):
            res = tr.iteration(it, feeds)
        all_taps, lib.TAPS[0] = lib.TAPS[0], None
        assert set(tr._graphs) == {'gen', 'disc'}
        torch.cuda.synchronize()
        for which in ('gen', 'disc'):
            refc = float(z[which + '/cost'])
            c = float(res[which + '_cost'])
            assert abs(c - refc) <= 1e-5 * max(1.0, abs(refc)), (which, c, refc)
            opt = next(o for k, o in _optimizers.items() if k[0] == which)
            names = [p.param_name for p in opt.params]
            grads = [opt.g[o:o + n].view(p.shape) for p, (o, n) in zip(opt.params, opt.slots)]
            taps = [(n_, t_) for sc, n_, t_ in all_taps if sc is not None and sc.startswith(which)]
            flips, worst = _gpu_flips(z, which, taps, cfg.B, last=False, table='fwd')
            prefix, tol = _grad_reference(z, name, 'timed', which, flips, worst, table='fwd')
            _check_grads(z, prefix, names, grads, tol if tol is not None else tol_of(name, which), tag=(name, 'timed'))
        optim.reset_optimizers()
        lib.delete_all_params()
        return
    for which in ('gen', 'disc'):
        lib.TAPS[0] = []
        out = tr.model.forward(tr.feed, which)
        taps, lib.TAPS[0] = [(n_, t_) for _, n_, t_ in lib.TAPS[0]], None
        refc = float(z[which + '/cost'])
        c = float(out[which + '_cost'].detach())
        assert abs(c - refc) <= 1e-5 * max(1.0, abs(refc)), (which, c, refc)
        opt = out[which + '_train_op'].optimizer
        names = [p.param_name for p in opt.params]
        grads = torch.autograd.grad(out[which + '_cost'], opt.params, allow_unused=True)
        flips, worst = _gpu_flips(z, which, taps, cfg.B, last=False, table='fwd')
        prefix, tol = _grad_reference(z, name, 'eager', which, flips, worst, table='fwd')
        _check_grads(z, prefix, names, grads, tol if tol is not None else tol_of(name, which), tag=(name, 'eager'))       # (filter gradients here are fp32 sums over up to 5e5 pixels)
    optim.reset_optimizers()
    lib.delete_all_params()
