"""not-gpu: host-side logic of the tflib mirror -- parameter registry semantics (tflib/__init__.py:9-47),
TF padding arithmetic, flat-buffer layout, the loop's step order."""
import numpy as np
import pytest


@pytest.fixture()
def lib():
    import graphical_gan_amd  # noqa: F401
    import tflib
    tflib.delete_all_params()
    tflib.delete_param_aliases()
    tflib.set_device('cpu')
    yield tflib
    tflib.delete_all_params()
    tflib.delete_param_aliases()
    tflib._device[0] = None


def test_tflib_alias_resolves_reference_import_paths():
    import graphical_gan_amd
    import tflib
    import tflib.ops.linear, tflib.ops.conv2d, tflib.ops.deconv2d, tflib.ops.batchnorm, tflib.objs.gan_inference, tflib.plot
    assert tflib is graphical_gan_amd.tflib
    for fn in ('Linear',):
        assert hasattr(tflib.ops.linear, fn)
    assert all(hasattr(tflib.objs.gan_inference, f) for f in ('ali', 'local_ep', 'local_ep_dynamic', 'weighted_local_epce', 'wali_gp'))
    assert all(hasattr(tflib.plot, f) for f in ('plot', 'tick', 'flush'))


def test_param_create_once_share_after(lib):
    a = lib.param('Generator.Input.W', np.ones((2, 3), 'float32'))
    b = lib.param('Generator.Input.W', np.zeros((2, 3), 'float32'))     # second init value is ignored
    assert a is b and float(a.detach().sum()) == 6.0 and a.requires_grad
    mv = lib.param('Generator.BN2.moving_mean', np.zeros(4, 'float32'), trainable=False)
    assert not mv.requires_grad
    lib.param('Extractor.1.Filters', np.zeros((5, 5, 3, 4), 'float32'))
    lib.param('Discriminator.HyperInput.W', np.zeros((3, 4), 'float32'))
    assert len(lib.params_with_name('Generator')) == 2                   # substring match incl. moving stats
    assert len(lib.params_with_name('Discriminator')) == 1
    assert lib.params_with_name('Hyper')[0] is lib.param('Discriminator.HyperInput.W', None)
    lib.delete_all_params()
    assert lib.params_with_name('Generator') == []


def test_param_aliases(lib):
    a = lib.param('A.W', np.ones(2, 'float32'))
    b = lib.param('B.W', np.zeros(2, 'float32'))
    lib.alias_params({a: b})
    assert lib.param('A.W', None) is b
    lib.delete_param_aliases()
    assert lib.param('A.W', None) is a


def test_initialisers_consume_rng_like_the_reference(lib):
    """conv2d.py:75-88 draws filter values on EVERY call, even when the param exists."""
    import torch
    from graphical_gan_amd import _lib
    np.random.seed(0)
    s0 = np.random.get_state()[1][:4].copy()
    x = torch.zeros(1, 3, 8, 8)
    for _ in range(2):
        with pytest.raises(_lib.GganError):       # op refuses CPU tensors, after creating/looking up the params
            lib.ops.conv2d.Conv2D('Extractor.1', 3, 4, 5, x, stride=2)
    w = lib.param('Extractor.1.Filters', None)
    assert tuple(w.shape) == (5, 5, 3, 4)
    bound = np.sqrt(4. / (3 * 25 + 4 * 25 / 4.)) * np.sqrt(3)
    assert float(w.abs().max()) <= bound + 1e-6
    np.random.seed(0)
    np.random.uniform(size=(5, 5, 3, 4)); np.random.uniform(size=(5, 5, 3, 4))
    expect = np.random.get_state()[1][:4]
    np.random.seed(0)
    with pytest.raises(_lib.GganError):
        lib.ops.conv2d.Conv2D('Extractor.1', 3, 4, 5, x, stride=2)
    with pytest.raises(_lib.GganError):
        lib.ops.conv2d.Conv2D('Extractor.1', 3, 4, 5, x, stride=2)
    assert np.array_equal(np.random.get_state()[1][:4], expect)
    # once a Trainer has declared the graphs built (session.run time in the reference), calls that find their parameters draw
    # nothing; new parameters still do, and delete_all_params() starts a new build phase
    lib.end_build_phase()
    before = np.random.get_state()[1][:4].copy()
    with pytest.raises(_lib.GganError):
        lib.ops.conv2d.Conv2D('Extractor.1', 3, 4, 5, x, stride=2)
    assert np.array_equal(np.random.get_state()[1][:4], before)
    with pytest.raises(_lib.GganError):
        lib.ops.conv2d.Conv2D('Extractor.9', 3, 4, 5, x, stride=2)
    assert not np.array_equal(np.random.get_state()[1][:4], before)
    lib.delete_all_params()
    assert lib.initial_values_needed('anything')
    assert lib.ops.deconv2d.Deconv2D.__defaults__ is not None
    with pytest.raises(Exception, match='Unsupported configuration'):
        lib.ops.deconv2d.Deconv2D('G.2', 4, 2, 5, x, mask_type=('a', 1))
    with pytest.raises(Exception, match='Invalid initialization!'):
        lib.ops.linear.Linear('L', 3, 4, torch.zeros(2, 3), initialization=('bogus', 1))


def test_same_padding_table():
    """k=5,s=2: 64/32/28/16/14/8 -> pad (1,2); 7 -> (2,2)."""
    from graphical_gan_amd import functional as F
    from oracle import ops as O
    for size in (64, 32, 28, 16, 14, 8):
        assert F.same_geometry(size, 5, 2) == (size // 2, 1)
        assert O.conv_geometry(size, 5, 2) == (size // 2, 1, 2)
    assert F.same_geometry(7, 5, 2) == (4, 2) and O.conv_geometry(7, 5, 2) == (4, 2, 2)
    assert F.same_geometry(12, 5, 1, 'VALID') == (8, 0)
    g = F.conv_geom(64, 3, 32, 32, 64, 5, 2)
    assert g == (64, 3, 32, 32, 64, 16, 16, 5, 2, 1, 1)


def test_flat_layout_is_aligned():
    from graphical_gan_amd.optim import layout_slots
    slots, total = layout_slots([5, 64, 100, 1])
    assert slots == [(0, 5), (64, 64), (128, 100), (256, 1)] and total == 320
    assert all(o % 64 == 0 for o, _ in slots)


def test_loop_step_order():
    """gmgan_inference_cifar10.py:480-494: iteration 0 = critic steps only; then 1 gen + CRITIC_ITERS critic."""
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models import Config

    class Fake(Trainer):
        def __init__(self, cfg):
            self.cfg, self.log = cfg, []

        def set_batch(self, b):
            self.log.append(('load', b))

        def step(self, which):
            self.log.append(which)
            return 0.0
    for mode, ci in (('ali', 1), ('wali-gp', 5)):
        t = Fake(Config('cifar10', mode=mode))
        assert t.cfg.critic_iters == ci
        t.iteration(0, iter(range(100)))
        assert [x for x in t.log if isinstance(x, str)] == ['disc'] * ci
        t.log = []
        t.iteration(1, iter(range(100)))
        assert [x for x in t.log if isinstance(x, str)] == ['gen'] + ['disc'] * ci
        assert [x[1] for x in t.log if not isinstance(x, str)] == list(range(1 + ci))    # fresh minibatch per run
        # ring mode (Trainer.use_ring): the same step order, nothing copied between the steps; without graphs (or with several
        # ranks) the iteration stays a sequence of steps
        t.feed, t.graph_enabled, t.log = {'ring': object()}, False, []
        t.iteration(2, iter(range(100)))
        assert t.log == ['gen'] + ['disc'] * ci


def test_plot_shim(tmp_path, capsys):
    import graphical_gan_amd  # noqa: F401
    import tflib.plot as plot
    plot.plot('time', 1.0); plot.tick(); plot.plot('time', 3.0)
    logfile = tmp_path / 'log.txt'
    plot.flush(str(tmp_path), str(logfile))
    assert 'time\t2.0' in capsys.readouterr().out and 'time\t2.0' in logfile.read_text()


def test_save_images_grid_and_png(tmp_path):
    import struct, zlib
    from graphical_gan_amd.tflib import save_images as SI
    X = np.linspace(0, 1, 12 * 3 * 4 * 4, dtype=np.float32).reshape(12, 3, 4, 4)
    img = SI.large_image(X)
    assert img.shape == (3 * 4, 4 * 4, 3) and img.dtype == np.uint8          # 12 samples -> 3 x 4 grid
    assert np.array_equal(img[:4, 4:8], (255.99 * X[1]).astype('uint8').transpose(1, 2, 0))
    assert SI.large_image(np.zeros((6, 49), np.float32)).shape == (2 * 7, 3 * 7)
    p = str(tmp_path / 'g.png')
    SI.save_images(X, p)
    raw = open(p, 'rb').read()
    assert raw[:8] == b'\x89PNG\r\n\x1a\n'
    w, h, depth, ctype = struct.unpack('>IIBB', raw[16:26])
    assert (w, h, depth, ctype) == (16, 12, 8, 2)
    n = struct.unpack('>I', raw[33:37])[0]
    rows = zlib.decompress(raw[41:41 + n])
    assert len(rows) == 12 * (1 + 16 * 3) and rows[1:49] == img[0].tobytes()


def test_api_surface_matches_reference_signatures():
    """positional order, keyword names and defaults of the reference's tflib entry points (extensions only
    appended as trailing keyword arguments)."""
    import inspect
    import graphical_gan_amd  # noqa: F401  (registers the `tflib` alias)
    import tflib as lib
    import tflib.ops.linear, tflib.ops.conv2d, tflib.ops.deconv2d, tflib.ops.batchnorm, tflib.objs.gan_inference, tflib.plot  # noqa
    import tflib.utils.distance, tflib.mnist, tflib.cifar10, tflib.celebA, tflib.simple_moving_mnist, tflib.save_images  # noqa
    J = lib.objs.gan_inference
    expect = {  # function: reference parameter list, in order (name or (name, default))
        lib.ops.linear.Linear: ['name', 'input_dim', 'output_dim', 'inputs', ('biases', True), ('initialization', None),
                                ('weightnorm', None), ('gain', 1.)],
        lib.ops.conv2d.Conv2D: ['name', 'input_dim', 'output_dim', 'filter_size', 'inputs', ('he_init', True), ('mask_type', None),
                                ('stride', 1), ('weightnorm', None), ('biases', True), ('gain', 1.), ('padding', 'SAME')],
        lib.ops.deconv2d.Deconv2D: ['name', 'input_dim', 'output_dim', 'filter_size', 'inputs', ('he_init', True),
                                    ('weightnorm', None), ('biases', True), ('gain', 1.), ('mask_type', None), ('stride', 2),
                                    ('padding', 'SAME')],
        lib.ops.batchnorm.Batchnorm: ['name', 'axes', 'inputs', ('is_training', None), ('stats_iter', None),
                                      ('update_moving_stats', True), ('fused', True)],
        J.local_ep: ['disc_fake_list', 'disc_real_list', 'gen_params', 'disc_params', ('lr', 2e-4), ('beta1', 0.5), ('beta2', .999),
                     ('s_f', None)],
        J.ali: ['disc_fake', 'disc_real', 'gen_params', 'disc_params', ('lr', 2e-4), ('beta1', 0.5), ('beta2', 0.999), ('s_f', None)],
        J.local_ep_dynamic: ['disc_fake_zz', 'disc_real_zz', 'disc_fake_xz', 'disc_real_xz', 'gen_params', 'disc_params', ('lr', 2e-4),
                             ('beta1', 0.5), ('beta2', .999), ('rec_penalty', None)],
        J.weighted_local_epce: ['disc_fake_list', 'disc_real_list', 'ratio_list', 'gen_params', 'disc_params', ('lr', 2e-4),
                                ('beta1', 0.5), ('rec_penalty', None)],
        J.wali_gp: ['disc_fake', 'disc_real', 'gradient_penalty', 'gen_params', 'disc_params', ('lr', 1e-4)],
        J.wali: ['disc_fake', 'disc_real', 'gen_params', 'disc_params', ('lr', 5e-5)],
        J.local_epce: ['disc_fake_list', 'disc_real_list', 'rec_penalty', 'gen_params', 'disc_params', ('lr', 2e-4), ('beta1', 0.5),
                       ('s_f', None)],
        J.alice: ['disc_fake', 'disc_real', 'rec_penalty', 'gen_params', 'disc_params', ('lr', 2e-4), ('beta1', 0.5), ('s_f', None)],
        J.vegan: ['disc_fake', 'disc_real', 'rec_penalty', 'gen_params', 'disc_params', 'lamb', ('lr', 2e-4), ('beta1', .5),
                  ('s_f', None)],
        J.vegan_wgan_gp: ['disc_fake', 'disc_real', 'rec_penalty', 'gradient_penalty', 'gen_params', 'disc_params', 'lamb',
                          ('lr', 2e-4), ('beta1', .5)],
        lib.utils.distance.distance: ['x', 'y', 'd_type'],
        lib.plot.plot: ['name', 'value'], lib.plot.tick: [], lib.plot.flush: ['outf', 'logfile'],
        lib.params_with_name: ['name'], lib.alias_params: ['replace_dict'], lib.print_model_settings: ['locals_'],
        lib.print_model_settings_to_file: ['locals_', 'logfile'], lib.print_model_settings_dict: ['settings'],
        lib.mnist.load: ['batch_size', 'test_batch_size', ('n_labelled', None)],
        lib.cifar10.load: ['batch_size', 'data_dir'],
        lib.celebA.load: ['batch_size', 'data_dir', ('num_dev', 5000)],
        lib.simple_moving_mnist.load_video: ['seq_length', 'batch_size', ('cla', None)],
        lib.save_images.save_images: ['X', 'save_path', ('size', None)],
    }
    for fn, ref in expect.items():
        ps = list(inspect.signature(fn).parameters.values())
        assert len(ps) >= len(ref), fn
        for p, r in zip(ps, ref):
            name, has_default, default = (r, False, None) if isinstance(r, str) else (r[0], True, r[1])
            assert p.name == name, (fn.__name__, p.name, name)
            assert (p.default is not inspect.Parameter.empty) == has_default, (fn.__name__, name)
            if has_default:
                assert p.default == default, (fn.__name__, name, p.default, default)
        for p in ps[len(ref):]:                                   # extensions must be optional
            assert p.default is not inspect.Parameter.empty, (fn.__name__, p.name)
    assert str(inspect.signature(lib.param)) == '(name, *args, **kwargs)'
    for name in ('enable_default_weightnorm', 'disable_default_weightnorm', 'set_weights_stdev', 'unset_weights_stdev'):
        assert callable(getattr(lib.ops.linear, name))


def test_second_leaf_shares_storage_and_keeps_gradients_apart(lib):
    """tflib.second_leaf: a second pass of a step sees the same weight VALUES through a second autograd leaf, so the tape returns
    two gradient contributions per weight instead of adding them (the optimizer sums them where it packs the bucket); frozen
    parameters and non-trainable entries are not affected; delete_all_params drops the leaves."""
    import torch
    w = lib.param('Discriminator.1.W', np.arange(6, dtype='float32').reshape(2, 3))
    mv = lib.param('Discriminator.BN.moving_mean', np.zeros(3, 'float32'), trainable=False)
    x = torch.ones(4, 2)
    y1 = (x @ lib.param('Discriminator.1.W', None)).sum()
    with lib.second_leaf():
        w2 = lib.param('Discriminator.1.W', None)
        assert lib.param('Discriminator.BN.moving_mean', None) is mv
        y2 = (2.0 * x @ w2).sum()
        with lib.frozen('Discriminator'):
            assert not lib.param('Discriminator.1.W', None).requires_grad
    assert w2 is not w and w2.is_leaf and w2.requires_grad and w2.data_ptr() == w.data_ptr()
    assert lib.param('Discriminator.1.W', None) is w                       # outside the context: the registry parameter again
    assert lib.second_leaf_for(w) is w2
    g1, g2 = torch.autograd.grad(y1 + y2, [w, w2])
    assert torch.equal(g1, torch.full((2, 3), 4.0)) and torch.equal(g2, torch.full((2, 3), 8.0))
    with torch.no_grad():
        w.mul_(2.0)
    assert torch.equal(w2.detach(), w.detach())                            # same storage: values follow
    with lib.second_leaf():
        assert lib.param('Discriminator.1.W', None) is w2                  # cached while the storage is unchanged
    w.data = w.data.clone()                                                # (what an optimizer does when it re-homes a parameter)
    with lib.second_leaf():
        w3 = lib.param('Discriminator.1.W', None)
    assert w3 is not w2 and w3.data_ptr() == w.data_ptr()
    lib.delete_all_params()
    assert lib.second_leaf_for(w) is None


def test_optimizer_returns_contribution_pairs(lib):
    """optim.AdamOptimizer.compute_gradients: a parameter that a second-leaf pass reached comes back as (gradient, second
    gradient); others as a plain tensor or None"""
    import torch
    from graphical_gan_amd import optim

    class Opt(optim.AdamOptimizer):          # (CPU: no flat device buffers, only the gradient bookkeeping under test)
        def __init__(self, params):
            self.params, self._one = list(params), None
    a = lib.param('Discriminator.a', np.ones(3, 'float32'))
    b = lib.param('Discriminator.b', np.ones(3, 'float32'))
    c = lib.param('Discriminator.c', np.ones(3, 'float32'))
    cost = (a * 2).sum() + (b * 3).sum()
    with lib.second_leaf():
        cost = cost + (lib.param('Discriminator.a', None) * 5).sum()
    g = Opt([a, b, c]).compute_gradients(cost)
    assert isinstance(g[0], tuple) and float(g[0][0][0]) == 2.0 and float(g[0][1][0]) == 5.0
    assert torch.is_tensor(g[1]) and float(g[1][0]) == 3.0 and g[2] is None


def test_fastdiv24_is_exact_for_every_tile_divisor():
    """csrc/conv.h make_fastdiv24 / fdiv24: floor(n / d) as one 24-bit multiply and a shift, used for the per-thread staging
    descriptors of the correlation kernels (n <= XE_MAX * 256 = 6144).  The same arithmetic restated with numpy integers: for every
    divisor a tile can produce, the chosen (mul, shift) keeps both factors below 2^24, the product below 2^32, and is exact for
    every numerator."""
    nmax = 24 * 256
    n = np.arange(nmax + 1, dtype=np.uint64)
    for d in list(range(1, 1025)) + [1408, 2047, 2048, 3071, 4096, 6143, 6144]:
        if d <= 1:
            mul, shift = 1, 0
        else:
            mul = shift = None
            for s in range(31, 0, -1):
                m = ((1 << s) + d - 1) // d
                if m >= (1 << 24) or m * nmax >= (1 << 32):
                    continue
                err = m * d - (1 << s)
                assert nmax * err < (1 << s), (d, s)          # make_fastdiv24 returns false here: never for these ranges
                mul, shift = m, s
                break
            assert mul is not None, d
        assert mul < (1 << 24) and mul * nmax < (1 << 32)
        q = (n * np.uint64(mul)) >> np.uint64(shift)
        assert np.array_equal(q, n // np.uint64(d)), d


def test_reference_hyperparameter_blocks_and_their_mode_rules():
    """run.reference_block: the UPPERCASE blocks of the ten reference scripts as data, with the MODE-dependent constants derived as
    the scripts derive them (gan_inference_cifar10.py:39-79, gmgan_inference_cifar10.py:39-87, gmgan_inference_face.py:35-54,
    ssgan_inference_moving_mnist.py:27-55, ssgan_inference_chairs.py:28-57), and run.config: the model configuration they describe."""
    import glob
    import os
    from graphical_gan_amd import run
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(root, 'scripts', '*.py')))
    assert len(names) == 10
    for n in names:
        S = run.reference_block(n)
        cfg = run.config(S)
        assert cfg.B == S['BATCH_SIZE'] and S['OUTPUT_DIM'] == cfg.output_dim and cfg.critic_iters == S['CRITIC_ITERS'], n
    S = run.reference_block('gan_inference_cifar10.py')
    assert (S['MODE'], S['TYPE_Q'], S['TYPE_P'], S['STD'], S['CRITIC_ITERS'], S['BATCH_SIZE'], S['LAMBDA'], S['LR'], S['BETA1'], S['ITERS'],
            S['DIM'], S['OUTPUT_DIM'], S['BN_FLAG'], S['DIM_LATENT'], S['N_VIS'], S['DR_RATE']) == (
        'ali', 'no_std', 'no_std', .1, 1, 64, 1., 2e-4, .5, 200000, 64, 3072, True, 128, 128, .2)
    assert 'DISTANCE_X' not in S and 'Z_SAMPLES' not in S and 'N_COMS' not in S
    S = run.reference_block('/some/where/gan_inference_cifar10.py', MODE='vegan')        # :52-59, :72-74
    assert (S['DISTANCE_X'], S['CRITIC_ITERS'], S['BN_FLAG'], S['DIM_LATENT']) == ('l2', 5, False, 8)
    cfg = run.config(S)
    assert (cfg.bn, cfg.dim_latent, cfg.critic_iters, cfg.latent_critic) == (False, 8, 5, True)
    S = run.reference_block('gan_inference_cifar10', MODE='vegan-kl')
    assert (S['TYPE_Q'], S['TYPE_P'], S['Z_SAMPLES'], S['CRITIC_ITERS'], S['DIM_LATENT']) == ('learn_std', 'no_std', 100, 0, 8)
    assert run.config(S).learn_std
    S = run.reference_block('gan_inference_mnist', MODE='vegan-mmd')
    assert (S['CRITIC_ITERS'], S['BN_FLAG'], S['DIM_LATENT'], S['BATCH_SIZE'], S['OUTPUT_DIM']) == (0, True, 128, 50, 784)
    S = run.reference_block('gmgan_inference_svhn')
    assert (S['N_COMS'], S['N_VIS'], S['MODE_K'], S['TEMP_INIT'], S['TEMP'], S['BN_FLAG'], S['DR_RATE']) == (50, 500, 'CONCRETE', .1, .1, False, .2)
    S = run.reference_block('gmgan_inference_face', N_COMS=10)
    assert (S['N_COMS'], S['N_VIS'], S['DIM_G'], S['DIM_D'], S['BATCH_SIZE'], S['BETA2'], S['DECAY']) == (10, 100, 32, 32, 128, .999, False)
    cfg = run.config(S)
    assert (cfg.K, cfg.dim, cfg.B, cfg.temp) == (10, 32, 128, .1)
    with pytest.raises(NotImplementedError):
        run.config(run.reference_block('gmgan_inference_cifar10', MODE_K='REINFORCE'))
    with pytest.raises(NotImplementedError):
        run.config(run.reference_block('gan_inference_cifar10', MODE='vae'))
    S = run.reference_block('ssgan_inference_chairs')
    assert (S['LEN'], S['OUTPUT_SHAPE'], S['OUTPUT_DIM'], S['OP_DYN_MODE'], S['N_C'], S['ITERS'], S['N_VIS'], S['DIM_LATENT_T'], S['BN_FLAG_G']) == (
        31, [3, 64, 64], 12288, 'res_w', 0, 40000, 50, 8, False)
    cfg = run.config(run.reference_block('ssgan_inference_moving_mnist', MODE='ali', ALI_MODE='3dcnn'))
    assert (cfg.seq_critic, cfg.ali_mode, cfg.LEN, cfg.lamb, cfg.lr) == (True, '3dcnn', 16, 0.1, 1e-4)


def test_bench_static_tables_are_keyed_by_workload_not_by_role():
    """bench.py measures the other BASELINE configurations in a process of their own, where each is the 'headline' of its process: the static
    per-(kernel, grid) figures (profiles/pmc_traffic.json) must still come from the table of THAT workload (round 6: they came from the
    G+D+GP step's table for every one of them)."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    head = dict(key='headline', dataset='cifar10', mode='wali-gp', ssgan_mode='local_ep', n_coms=None, batch_size=None)
    assert bench._static_key(head) == 'headline'
    for v in bench.VARIANTS:
        as_child = dict(head, dataset=v['dataset'], mode=v.get('mode', 'wali-gp'), ssgan_mode=v.get('ssgan_mode', 'local_ep'),
                        n_coms=v.get('n_coms'))
        assert bench._static_key(as_child) == v['key'], v
        assert bench._static_key(dict(v)) == v['key']                       # (in-process: its own key)
    assert bench._static_key(dict(head, batch_size=16)) not in [v['key'] for v in bench.VARIANTS] + ['headline']
    import json
    tab = json.load(open(os.path.join(root, 'profiles', 'pmc_traffic.json')))
    for v in bench.VARIANTS:
        assert v['key'] in tab or v['key'] == 'gmgan-cifar10-K30', v['key']      # (K = 30 reads the K = 10 table: same launch shapes)
    assert 'headline' in tab and tab.get('_build') and tab.get('_tag')
