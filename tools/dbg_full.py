"""debug: per-tensor deviation of the HIP path from a full-size fixture (python tools/dbg_full.py full_mnist_ali)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch
import make_golden_full as MG
from _golden import load
from oracle import nets as N, step as S
from graphical_gan_amd import tflib as lib, optim
from graphical_gan_amd.engine import Trainer
from graphical_gan_amd.models import Config
name = sys.argv[1]
z = load(name)
dataset, B, K, mode = MG.FULL[name]
ocfg = N.Cfg(dataset, batch_size=B, n_coms=K)
P0 = MG.perturbed_params(ocfg)
feed = S.make_feed(ocfg, np.random.default_rng(int(z['feed_seed'])), MG.omode_of(mode))
tr = Trainer(Config(dataset, batch_size=B, n_coms=K, mode=mode, fuse=not os.environ.get('NOFUSE')), device='cuda:0', graph=False, inject_noise=True)
tr.load_params(P0)
tr.set_feed(feed)
for which in ('gen', 'disc'):
    out = tr.model.forward(tr.feed, which)
    print(which, 'cost', float(out[which + '_cost']), float(z[which + '/cost']))
    opt = out[which + '_train_op'].optimizer
    names = [p.param_name for p in opt.params]
    grads = [(g[0] + g[1]) if isinstance(g, tuple) else g for g in opt.compute_gradients(out[which + '_cost'])]
    refs = {n: z['%s/g/%s' % (which, n)] for n in names if '%s/g/%s' % (which, n) in z.files}
    gmax = max(r[1] for r in refs.values())
    for n, g in zip(names, grads):
        if n not in refs:
            continue
        ref = refs[n]
        f = g.detach().cpu().numpy().astype(np.float64).reshape(-1)
        scale = max(ref[1], 1e-2 * gmax)
        idx = MG.sample_index(n, f.size)
        e = np.abs(f[idx] - ref[2:])
        print('  %-34s err/scale %.2e  l2rel %.2e  nbad %d/%d  scale %.2e' % (n, e.max() / scale, abs(np.linalg.norm(f) - ref[0]) / max(ref[0], 1e-30), (e > 1e-3 * scale).sum(), len(e), scale))
