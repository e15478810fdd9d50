#!/usr/bin/env python
"""List the torch (aten) operators that launch kernels inside one training iteration -- the glue outside libggan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from graphical_gan_amd.engine import Trainer, synthetic_ring
from graphical_gan_amd.models import Config

dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'ali'
cfg = Config('cifar10', batch_size=64, mode=mode, n_coms=30 if mode.startswith('local_ep') else 0)
np.random.seed(0)
tr = Trainer(cfg, device=dev, graph=False)
ring = synthetic_ring(cfg, dev, n=4)
bi = iter(ring * 100)
for it in range(3):
    tr.iteration(it, bi)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.iteration(3, bi)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith('aten::') and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
for e in rows:
    print('%-40s calls=%3d  device_us=%8.1f' % (e.key, e.count, e.device_time_total))
