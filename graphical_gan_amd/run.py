"""The driver loop shared by the counterpart scripts in scripts/ (same file names and UPPERCASE hyper-parameter blocks as the
reference's gan_inference_* / gmgan_inference_* / ssgan_inference_* scripts; their train loop is
gmgan_inference_cifar10.py:470-549: alternate one generator step and CRITIC_ITERS critic steps on fresh minibatches, log the
costs through tflib.plot, dump sample grids and a checkpoint every so often)."""
import os
import time

import numpy as np
import torch

from . import checkpoint
from . import tflib as lib
from .data import DevicePrefetcher
from .engine import Trainer


def _batches(S, model, device):
    """real data through the py3 loaders.  A missing dataset raises, as the reference scripts do; the synthetic ring of the
    same shapes is an explicit opt-in (S['SYNTHETIC'] = True or GGAN_SYNTHETIC=1: smoke runs, benchmarks)."""
    ds = S['DATASET']
    synthetic_ok = bool(S.get('SYNTHETIC')) or os.environ.get('GGAN_SYNTHETIC', '') not in ('', '0')
    try:
        if synthetic_ok and S.get('SYNTHETIC') == 'force':
            raise FileNotFoundError('synthetic data requested')
        if ds == 'mnist':
            train, _, _ = lib.mnist.load(S['BATCH_SIZE'], S['BATCH_SIZE'])
            return DevicePrefetcher(train, device, pick=[0]), 'mnist.pkl.gz'
        if ds == 'cifar10':
            train, _ = lib.cifar10.load(S['BATCH_SIZE'], S.get('DATA_DIR', ''))
            return DevicePrefetcher(train, device, pick=[0], dtypes=[np.int32]), S.get('DATA_DIR')
        if ds == 'svhn':
            train, _ = lib.svhn.load(S['BATCH_SIZE'], S.get('DATA_DIR', ''))
            return DevicePrefetcher(train, device, pick=[0], dtypes=[np.int32]), S.get('DATA_DIR')
        if ds == 'face':
            train, _ = lib.celebA.load(S['BATCH_SIZE'], S.get('DATA_DIR', ''))
            return DevicePrefetcher(train, device, dtypes=[np.int32]), S.get('DATA_DIR')
        if ds == 'moving_mnist':
            train, _ = lib.simple_moving_mnist.load_video(S['LEN'], S['BATCH_SIZE'])

            def with_onehot():
                for x, y in train():
                    oh = np.zeros((len(y), S['N_C']), np.float32)
                    oh[np.arange(len(y)), y] = 1
                    yield x, oh
            return DevicePrefetcher(with_onehot, device), 'mnist.pkl.gz'
        if ds == 'chairs':
            train, _ = lib.chairs.load(S['LEN'], S['BATCH_SIZE'], 64, S.get('DATA_DIR', ''))
            return DevicePrefetcher(train, device), S.get('DATA_DIR')
    except FileNotFoundError as e:
        if not synthetic_ok:
            raise FileNotFoundError("%s -- dataset %r not found (DATA_DIR=%r); set S['SYNTHETIC'] = True or GGAN_SYNTHETIC=1 to "
                                    "train on synthetic minibatches instead" % (e, ds, S.get('DATA_DIR', '')))
        print('[run] %s -> synthetic minibatches (opt-in)' % e)
    ring = model.synthetic_ring(device, n=8)

    def forever():
        i = 0
        while True:
            yield ring[i % len(ring)]
            i += 1
    return forever(), 'synthetic'


# ---- the reference scripts' UPPERCASE hyper-parameter blocks, as data ------------------------------------------------------------
# per script: what differs between the image scripts (gan_inference_cifar10.py:39-79, gan_inference_svhn.py:32-72,
# gan_inference_mnist.py:31-70, gan_inference_face.py:33-50, gmgan_inference_cifar10.py:39-87, gmgan_inference_svhn.py:33-81,
# gmgan_inference_mnist.py:32-80, gmgan_inference_face.py:35-54); the MODE-dependent constants are derived below as the scripts do
_IMAGE_SCRIPTS = {
    'gan_inference_cifar10': dict(DATASET='cifar10', MODE='ali', BATCH_SIZE=64, ITERS=200000, DIM=64, OUTPUT_DIM=3072, BN_FLAG=True, DR_RATE=.2),
    'gan_inference_svhn': dict(DATASET='svhn', MODE='ali', BATCH_SIZE=64, ITERS=200000, DIM=64, OUTPUT_DIM=3072, BN_FLAG=False, DR_RATE=.2),
    'gan_inference_mnist': dict(DATASET='mnist', MODE='ali', BATCH_SIZE=50, ITERS=200000, DIM=64, OUTPUT_DIM=784, BN_FLAG=True),
    'gan_inference_face': dict(DATASET='face', MODE='ali', BATCH_SIZE=128, ITERS=100000, DIM_G=32, DIM_D=32, OUTPUT_DIM=12288,
                               BN_FLAG=False, DECAY=False, BETA2=.999),
    'gmgan_inference_cifar10': dict(DATASET='cifar10', MODE='local_ep', BATCH_SIZE=64, ITERS=200000, DIM=64, OUTPUT_DIM=3072, BN_FLAG=True,
                                    N_COMS=30, DR_RATE=.2),
    'gmgan_inference_svhn': dict(DATASET='svhn', MODE='local_ep', BATCH_SIZE=64, ITERS=200000, DIM=64, OUTPUT_DIM=3072, BN_FLAG=False,
                                 N_COMS=50, DR_RATE=.2),
    'gmgan_inference_mnist': dict(DATASET='mnist', MODE='local_ep', BATCH_SIZE=50, ITERS=200000, DIM=64, OUTPUT_DIM=784, BN_FLAG=True,
                                  N_COMS=30),
    'gmgan_inference_face': dict(DATASET='face', MODE='local_ep', BATCH_SIZE=128, ITERS=100000, DIM_G=32, DIM_D=32, OUTPUT_DIM=12288,
                                 BN_FLAG=False, DECAY=False, BETA2=.999, N_COMS=100),
}
# ssgan_inference_moving_mnist.py:27-55 / ssgan_inference_chairs.py:28-57
_SEQUENCE_SCRIPTS = {
    'ssgan_inference_moving_mnist': dict(DATASET='moving_mnist', MODE='local_ep', POS_MODE='naive_mean_field', ALI_MODE='concat_x',
                                         OP_DYN_MODE='res', BN_FLAG=False, DIM_LATENT_G=128, DIM_LATENT_L=8, DIM=32, DIM_OP=256, LEN=16,
                                         OUTPUT_SHAPE=[1, 64, 64], N_C=10, LAMBDA=0.1, LR=1e-4, BATCH_SIZE=50, BETA1=.5, BETA2=.999,
                                         ITERS=100000, CRITIC_ITERS=1),
    'ssgan_inference_chairs': dict(DATASET='chairs', MODE='local_ep', POS_MODE='naive_mean_field', ALI_MODE='concat_x', OP_COM_MODE='concat',
                                   OP_DYN_MODE='res_w', BN_FLAG=False, BN_FLAG_OP=False, DIM_LATENT_G=128, DIM_LATENT_L=8, DIM=32,
                                   DIM_OP=256, LEN=31, OUTPUT_SHAPE=[3, 64, 64], N_C=0, LAMBDA=0.1, LR=1e-4, BATCH_SIZE=50, BETA1=.5,
                                   BETA2=.999, ITERS=40000, CRITIC_ITERS=1),
}
_NO_CRITIC = ('vegan-mmd', 'vegan-kl', 'vegan-ikl', 'vegan-jsd', 'vae')
_RECON = ('alice', 'alice-z', 'alice-x', 'vegan', 'vegan-wgan-gp', 'vegan-kl', 'vegan-ikl', 'vegan-jsd', 'vegan-mmd', 'local_epce')


def reference_block(script, **overrides):
    """-> dict: the UPPERCASE hyper-parameter block the reference script of that name builds for its MODE (`script`: a file name or
    path, e.g. __file__), with `overrides` (MODE, ITERS, N_COMS, BATCH_SIZE, ...) applied BEFORE the MODE-dependent constants are
    derived, as editing the script's top would."""
    name = os.path.splitext(os.path.basename(script))[0]
    if name in _SEQUENCE_SCRIPTS:
        S = dict(_SEQUENCE_SCRIPTS[name])
        S.update(overrides)
        S.setdefault('BN_FLAG_G', S['BN_FLAG']); S.setdefault('BN_FLAG_E', S['BN_FLAG']); S.setdefault('BN_FLAG_D', S['BN_FLAG'])
        S.setdefault('DIM_LATENT_T', S['DIM_LATENT_L'])
        S.setdefault('OUTPUT_DIM', int(np.prod(S['OUTPUT_SHAPE'])))
        S.setdefault('N_VIS', S['BATCH_SIZE'])
        return S
    S = dict(_IMAGE_SCRIPTS[name])
    S.update(overrides)
    mode, bn = S['MODE'], S.pop('BN_FLAG')
    if mode in ('vegan-kl', 'vegan-ikl', 'vegan-jsd'):          # gan_inference_cifar10.py:40-49
        S.setdefault('TYPE_Q', 'learn_std'); S.setdefault('TYPE_P', 'no_std'); S.setdefault('Z_SAMPLES', 100)
    elif mode == 'vae':
        S.setdefault('TYPE_Q', 'learn_std'); S.setdefault('TYPE_P', 'learn_std')
    else:
        S.setdefault('TYPE_Q', 'no_std'); S.setdefault('TYPE_P', 'no_std')
    S.setdefault('STD', .1)
    if mode in _RECON:
        S.setdefault('DISTANCE_X', 'l2')
    S.setdefault('CRITIC_ITERS', 0 if mode in _NO_CRITIC else (5 if mode in ('vegan', 'vegan-wgan-gp', 'wali', 'wali-gp') else 1))   # :54-59
    S.setdefault('LAMBDA', 1.)
    # (the wali objectives build their own optimizers: RMSProp 5e-5 / Adam 1e-4, tflib/objs/gan_inference.py:4,28)
    S.setdefault('LR', {'wali-gp': 1e-4, 'wali': 5e-5}.get(mode, 2e-4))
    S.setdefault('BETA1', .9 if mode == 'vae' else .5)
    thin = mode in ('vegan', 'vegan-wgan-gp', 'vegan-kl', 'vegan-jsd', 'vegan-ikl')      # :72-77
    S.setdefault('BN_FLAG', False if thin else bn)
    S.setdefault('DIM_LATENT', 8 if thin else 128)
    if 'N_COMS' in S:
        S.setdefault('N_VIS', S['N_COMS'] * 10)
        S.setdefault('MODE_K', 'CONCRETE')
        if S['MODE_K'] == 'REINFORCE':
            S.setdefault('CONTROL_VARIATE', .0)
        elif S['MODE_K'] in ('CONCRETE', 'STRAIGHT_THROUGHT_CONCRETE'):
            S.setdefault('TEMP_INIT', .1)
            S.setdefault('TEMP', S['TEMP_INIT'])
    else:
        S.setdefault('N_VIS', S['BATCH_SIZE'] * 2)
    return S


def config(S):
    """the model configuration a settings block describes (Config / SSConfig); constants this implementation fixes are checked,
    not silently ignored"""
    if S['DATASET'] in ('moving_mnist', 'chairs'):
        from .models_ssgan import SSConfig
        assert not S.get('BN_FLAG') and S.get('DIM_LATENT_T', S['DIM_LATENT_L']) == S['DIM_LATENT_L'] and S.get('BETA1', .5) == .5
        return SSConfig(batch_size=S['BATCH_SIZE'], length=S['LEN'], dim=S['DIM'], dim_op=S['DIM_OP'], dim_g=S['DIM_LATENT_G'],
                        dim_l=S['DIM_LATENT_L'], n_c=S['N_C'], pos_mode=S['POS_MODE'], op_dyn_mode=S['OP_DYN_MODE'], lr=S['LR'],
                        channels=S['OUTPUT_SHAPE'][0], dataset=S['DATASET'], mode=S['MODE'], lamb=S['LAMBDA'], ali_mode=S['ALI_MODE'])
    from .models import Config
    if S.get('MODE_K', 'CONCRETE') != 'CONCRETE':
        raise NotImplementedError('MODE_K = %r: only the CONCRETE relaxation is built (DESIGN.md 0)' % S['MODE_K'])
    if S['MODE'] == 'vae':
        raise NotImplementedError('MODE vae: the reference Generator returns no decoder statistics (DESIGN.md 8)')
    assert S.get('DISTANCE_X', 'l2') == 'l2' and S.get('LAMBDA', 1.) == 1. and S.get('BETA1', .5) == .5 and S.get('Z_SAMPLES', 100) == 100, S
    assert S.get('DIM_G', S.get('DIM')) == S.get('DIM_D', S.get('DIM')), 'one model width'
    return Config(S['DATASET'], batch_size=S['BATCH_SIZE'], n_coms=S.get('N_COMS', 0), mode=S['MODE'], dim=S.get('DIM', S.get('DIM_G')),
                  dim_latent=S['DIM_LATENT'], bn=S['BN_FLAG'], temp=S.get('TEMP', 0.1), lr=S['LR'])


def train(S, cfg, model=None, out_dir=None):
    """S: dict of the script's UPPERCASE settings (needs DATASET, BATCH_SIZE, ITERS; optional SAVE_EVERY, LOG_EVERY, SEED)."""
    lib.print_model_settings_dict(S)
    device = lib.get_device()
    np.random.seed(S.get('SEED', 0))
    torch.manual_seed(S.get('SEED', 0))
    tr = Trainer(cfg, device=device, graph=S.get('HIP_GRAPH', True), model=model, sync_bn=S.get('SYNC_BN', False))
    batches, source = _batches(S, tr.model, device)
    print('[run] data: %s' % source)
    out_dir = out_dir or S.get('OUT_DIR')
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'logfile.txt'), 'a') as f:
            f.write('data source: %s\n' % source)
    # `time` = seconds per iteration since the start, on the device clock (HIP events on the Trainer's stream)
    timed = device.type == 'cuda'
    t0 = time.time()
    if timed:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(device))
    for it in range(S['ITERS']):
        if (it == 2 and isinstance(batches, DevicePrefetcher) and S.get('RING_FEED', True) and tr.graph_enabled
                and isinstance(tr.feed, dict) and 'real_x_int' in tr.feed and tr.world == 1):
            # int32 image data from a loader: from here on the host minibatches go into a device ring one iteration ahead and an
            # iteration is one graph replay (Trainer.use_host_ring); the loader's host iterator is continued where it stands
            tr.use_host_ring(batches)
        res = tr.iteration(it, batches)
        if it % S.get('LOG_EVERY', 100) == 0 or it == S['ITERS'] - 1:
            for k, v in res.items():
                lib.plot.plot(k.replace('_', ' '), float(v))
            if timed:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(device))
                ev.synchronize()
                lib.plot.plot('time', ev0.elapsed_time(ev) * 1e-3 / (it + 1))
            else:
                lib.plot.plot('time', (time.time() - t0) / (it + 1))
            lib.plot.flush(out_dir, os.path.join(out_dir, 'logfile.txt') if out_dir else None)
        lib.plot.tick()
        if out_dir and S.get('SAVE_EVERY') and (it + 1) % S['SAVE_EVERY'] == 0:
            checkpoint.save(os.path.join(out_dir, 'params_%d.npz' % (it + 1)), tr, data_source=source)
            with torch.no_grad():
                nets = tr.model.forward_nets(tr.feed)
                # (the critic-free code-space modes never build Generator(p_z) in a step: samples are drawn here)
                fx = nets['fake_x'] if 'fake_x' in nets else tr.model.Generator(nets['p_z'])
            fx = fx.detach().float().cpu().numpy()
            side = getattr(cfg, 'S', 64)
            fx = fx.reshape(-1, getattr(cfg, 'C', 1), side, side)[:64]
            lo = 0.0 if getattr(cfg, 'out_act', 'tanh') == 'sigmoid' else -1.0
            lib.save_images.save_images(np.clip((fx - lo) / (1.0 - lo), 0, 1), os.path.join(out_dir, 'samples_%d.png' % (it + 1)))
    tr.flush()
    torch.cuda.synchronize()
    return tr
