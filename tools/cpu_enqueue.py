#!/usr/bin/env python
"""Host-side enqueue time per iteration vs GPU time (is the launch path CPU-bound?).  env GGAN_FORCE_SPLIT_GRAPH=1 for the DP path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphical_gan_amd.engine import Trainer, synthetic_ring
from graphical_gan_amd.models import Config
dev = torch.device('cuda:0')
cfg = Config('cifar10', batch_size=64, mode='ali')
np.random.seed(0)
tr = Trainer(cfg, device=dev, graph=True)
ring = synthetic_ring(cfg, dev, n=4)
bi = iter(ring * 1000)
for it in range(8):
    tr.iteration(it, bi)
tr.flush(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for it in range(100):
        tr.iteration(10 + it, bi)
    t1 = time.perf_counter()
    tr.flush(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('split=%s enqueue %.3f ms/it, total %.3f ms/it' % (tr.split_graph, (t1 - t0) * 10, (t2 - t0) * 10))
