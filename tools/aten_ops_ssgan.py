#!/usr/bin/env python
"""List the torch (aten) operators that launch kernels inside one state-space-GAN iteration (the glue outside libggan)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from graphical_gan_amd.engine import Trainer
from graphical_gan_amd.models_ssgan import SSConfig, StateSpaceGAN

dev = torch.device('cuda:0')
cfg = SSConfig(batch_size=32)
np.random.seed(0)
model = StateSpaceGAN(cfg)
tr = Trainer(cfg, device=dev, graph=False, model=model)
ring = model.synthetic_ring(dev, n=4)
bi = iter(ring * 100)
for it in range(3):
    tr.iteration(it, bi)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.iteration(3, bi)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith('aten::') and e.device_time_total > 0]
rows.sort(key=lambda e: -e.count)
tot = 0
for e in rows:
    print('%-40s calls=%3d  device_us=%8.1f' % (e.key, e.count, e.device_time_total))
    tot += e.count
print('total aten kernel-launching calls', tot)
