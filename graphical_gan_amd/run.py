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
    t0 = time.time()
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
