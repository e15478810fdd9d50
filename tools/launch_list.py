#!/usr/bin/env python
"""One eager iteration with GGAN_TRACE_LAUNCHES=1: every launch with its grid, block, dynamic LDS and flops (stderr).
usage: python tools/launch_list.py [dataset] [mode] [batch] [n_coms]     (default: cifar10 ali 64 0 = the headline)"""
import os
import sys

os.environ['GGAN_TRACE_LAUNCHES'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models import Config
    dataset = sys.argv[1] if len(sys.argv) > 1 else 'cifar10'
    mode = sys.argv[2] if len(sys.argv) > 2 else 'ali'
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    n_coms = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    dev = torch.device('cuda:0')
    np.random.seed(0)
    cfg = Config(dataset, batch_size=batch, n_coms=n_coms, mode=mode)
    tr = Trainer(cfg, device=dev, graph=False, seed=1234)
    ring = tr.model.synthetic_ring(dev, n=4, seed=1234)

    def batches():
        i = 0
        while True:
            yield ring[i % len(ring)]
            i += 1
    bi = batches()
    sys.stderr.write('==== iteration 0 (critic step only)\n')
    tr.iteration(0, bi)
    torch.cuda.synchronize()
    sys.stderr.write('==== iteration 1 (generator step, then critic step)\n')
    tr.iteration(1, bi)
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
