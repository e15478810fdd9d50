"""stand-alone timing of the fused mixture-critic chain (ggan_mlp_chain_*) against the composed layers: python tools/bench_mlp.py [M]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphical_gan_amd import functional as F          # noqa: E402
from graphical_gan_amd import tflib as lib              # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = 'cuda:0'
rng = np.random.default_rng(0)
z = torch.tensor(rng.standard_normal((M, 128)), dtype=torch.float32, device=dev)
k = torch.tensor(rng.random((M, 30)), dtype=torch.float32, device=dev)
names = ['T.HyperInput', 'T.Hyper2', 'T.Hyper3']


def run(chain, mode):
    zz, kk = z.clone().requires_grad_(mode != 'fwd'), k.clone().requires_grad_(mode != 'fwd')
    ctxm = lib.frozen('T.') if mode == 'data' else lib.frozen()
    with ctxm:
        if chain:
            lg = lib.ops.linear.MlpLReLUChain(names, 158, 512, 'T.HyperOutput', (zz, kk))
        else:
            h = lib.ops.linear.Linear(names[0], 158, 512, (zz, kk), activation=F.ACT_LRELU)
            h = lib.ops.linear.Linear(names[1], 512, 512, h, activation=F.ACT_LRELU)
            lg = lib.ops.linear.LinearLReLULinear(names[2], 512, 512, 'T.HyperOutput', h)
    if mode == 'fwd':
        return
    ps = [p for p in lib.params_with_name('T.') if p.requires_grad] if mode == 'all' else []
    torch.autograd.grad(lg, [zz, kk] + ps, grad_outputs=torch.ones_like(lg))


st = F.shared_stream(dev, 'capture')
for mode in ('fwd', 'data', 'all'):
    for chain in (False, True):
        with torch.cuda.stream(st):
            for _ in range(5):
                run(chain, mode)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                run(chain, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        print('%-5s %-8s %.1f us per pass' % (mode, 'chain' if chain else 'composed', (time.perf_counter() - t0) / 200 * 1e6))
