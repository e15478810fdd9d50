"""Paired nets pass (engine._iteration_pair) against one nets pass per step: same flattened step sequence, weights after it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphical_gan_amd import tflib as lib, optim
from graphical_gan_amd.engine import Trainer
from graphical_gan_amd.models import Config

dev = torch.device('cuda:0')
dataset = sys.argv[1] if len(sys.argv) > 1 else 'cifar10'
mode = sys.argv[2] if len(sys.argv) > 2 else 'ali'
K = 30 if mode == 'local_ep' else 0
NIT = int(sys.argv[3]) if len(sys.argv) > 3 else 6


def run(pair):
    optim.reset_optimizers(); lib.delete_all_params()
    np.random.seed(0); torch.manual_seed(1234)
    cfg = Config(dataset, batch_size=64, n_coms=K, mode=mode)
    tr = Trainer(cfg, device=dev, graph=True, seed=1234, pair_nets=pair)
    ring = tr.model.synthetic_ring(dev, n=8, seed=1234)
    bi = iter(ring * 100)
    tr.iteration(0, bi); tr.iteration(1, bi)
    tr.use_ring(ring)
    costs = []
    for it in range(2, 2 + NIT):
        r = tr.iteration(it, None)
        costs.append({k: float(v) for k, v in r.items()})
    if not pair:
        tr.step('gen')          # the paired grouping ends one generator step later in the sequence
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for it in range(n):
        tr.iteration(100 + it, None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    return None, costs, ms


def weights(pair):
    optim.reset_optimizers(); lib.delete_all_params()
    np.random.seed(0); torch.manual_seed(1234)
    cfg = Config(dataset, batch_size=64, n_coms=K, mode=mode)
    tr = Trainer(cfg, device=dev, graph=True, seed=1234, pair_nets=pair)
    ring = tr.model.synthetic_ring(dev, n=8, seed=1234)
    bi = iter(ring * 100)
    tr.iteration(0, bi); tr.iteration(1, bi)
    P1 = tr.get_params()
    tr.use_ring(ring)
    costs = []
    for it in range(2, 2 + NIT):
        r = tr.iteration(it, None)
        costs.append({k: float(v) for k, v in r.items()})
    if not pair:
        tr.step('gen')
    torch.cuda.synchronize()
    return P1, tr.get_params(), costs


P1a, Pa, ca = weights(False)
P1b, Pb, cb = weights(True)
print('costs plain', ca[:3]); print('costs pair ', cb[:3])
worst = 0
for n in sorted(Pa):
    if n.endswith('.moving_mean') or n.endswith('.moving_variance'):
        continue
    upd = np.linalg.norm(Pa[n] - P1a[n])
    d = np.linalg.norm(Pa[n] - Pb[n])
    rel = d / max(upd, 1e-12)
    worst = max(worst, rel) if not (n.endswith('.b') and upd < 1e-6) else worst
    if rel > 1e-3:
        print('%-32s |update| %.3e  |pair - plain| %.3e  rel %.2e' % (n, upd, d, rel))
print('worst relative difference of the weights (to the size of the update):', worst)
_, _, ms_a = run(False)
_, _, ms_b = run(True)
print('ms/iteration plain %.4f  paired %.4f' % (ms_a, ms_b))
