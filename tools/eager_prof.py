import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from graphical_gan_amd.engine import Trainer, synthetic_ring
from graphical_gan_amd.models import Config
dev = torch.device('cuda:0')
cfg = Config('cifar10', batch_size=64, mode='ali')
np.random.seed(0)
tr = Trainer(cfg, device=dev, graph=False)
ring = synthetic_ring(cfg, dev, n=4)
bi = iter(ring * 1000)
for it in range(5):
    tr.iteration(it, bi)
torch.cuda.synchronize()
t0 = time.time()
for it in range(5, 25):
    tr.iteration(it, bi)
torch.cuda.synchronize()
print('eager ms/iter', (time.time() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for it in range(25, 35):
    tr.iteration(it, bi)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
