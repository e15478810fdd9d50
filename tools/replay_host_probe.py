#!/usr/bin/env python
"""How long the HOST spends in one iteration-graph replay call, and whether it waits for the previous replay (ring mode, no
synchronisation in the loop).  usage: python tools/replay_host_probe.py [mode] [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models import Config
    mode = sys.argv[1] if len(sys.argv) > 1 else 'wali-gp'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dev = torch.device('cuda:0')
    np.random.seed(0)
    cfg = Config('cifar10', batch_size=64, n_coms=0, mode=mode)
    tr = Trainer(cfg, device=dev, graph=True, seed=1234)
    ring = tr.model.synthetic_ring(dev, n=8, seed=1234)
    bi = iter(ring * 1000)
    tr.iteration(0, bi); tr.iteration(1, bi)
    tr.use_ring(ring)
    for it in range(2, 8):
        tr.iteration(it, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = []
    for it in range(8, 8 + n):
        a = time.perf_counter()
        tr.iteration(it, None)
        host.append((time.perf_counter() - a) * 1e3)
    t_issue = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) * 1e3
    print('%s: host ms per replay call: %s' % (mode, ' '.join('%.2f' % h for h in host)))
    print('issued %d replays in %.2f ms; all done after %.2f ms (%.3f ms per iteration)' % (n, t_issue, t_all, t_all / n))


if __name__ == '__main__':
    main()
