"""host-core probe for the CPU baseline: affinity, and the torch-CPU step at a few thread counts (bounded)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nets as N, step as S, torch_cpu
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads default', torch.get_num_threads())
try:
    print(open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('no cgroup cpu.max', e)
cfg = N.Cfg('cifar10', batch_size=64)
rng = np.random.default_rng(0)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    ts = torch_cpu.Step(cfg, N.init_params(cfg, 0), torch.float32)
    def one():
        for which in ('gen', 'disc'):
            f = S.make_feed(cfg, rng); ts.step(which, S.real_x_from_feed(cfg, f, np.float32), f['p_z_noise'])
    t = time.time(); one(); w = time.time() - t
    n, t = 0, time.time()
    while n < 1 or (time.time() - t < 4 and n < 50):
        one(); n += 1
    print('threads %3d: warm-up %.2fs, %.1f img/s' % (nt, w, 64 * n / (time.time() - t)), flush=True)
    if 64 * n / (time.time() - t) < 5:
        break
