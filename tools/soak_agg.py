"""Long run of a critic-free code-space mode with periodic statistics of the stochastic encoder (finite costs, range of std / mean).
usage: python tools/soak_agg.py vegan-jsd 20000 [every] [script]   ('script': the scripts' DIM_LATENT = 8, BN_FLAG = False for these modes)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from graphical_gan_amd.engine import Trainer  # noqa: E402
from graphical_gan_amd.models import Config  # noqa: E402

mode, steps = sys.argv[1], int(sys.argv[2])
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
np.random.seed(0)
dev = torch.device('cuda', 0)
script = len(sys.argv) > 4 and sys.argv[4] == 'script'
cfg = Config('cifar10', batch_size=64, mode=mode, **(dict(dim_latent=8, bn=False) if script else {}))
tr = Trainer(cfg, device=dev, graph=True, seed=1234)
ring = tr.model.synthetic_ring(dev, n=8, seed=1234)


def batches():
    i = 0
    while True:
        yield ring[i % len(ring)]
        i += 1


bi = batches()
first_bad = None
for it in range(steps):
    r = tr.iteration(it, bi)
    if it % every == 0 or it == steps - 1:
        tr.flush()
        with torch.no_grad():
            nets = tr.model.forward_nets(tr.feed)
        sd, mu = nets['q_z_std'], nets['q_z_mean']
        c = float(r['gen_cost']) if 'gen_cost' in r else float('nan')
        print('it %6d cost %.6g  std [%.3g, %.3g]  |mean| max %.3g  finite %s' % (it, c, float(sd.min()), float(sd.max()), float(mu.abs().max()),
                                                                               bool(torch.isfinite(sd).all() and torch.isfinite(mu).all())), flush=True)
        if not np.isfinite(c) and it > 0 and first_bad is None:
            first_bad = it
            break
print('first non-finite cost at', first_bad)
