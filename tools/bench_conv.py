#!/usr/bin/env python
"""Micro-benchmark of the conv-family kernels at the hot-path shapes (HIP events, L2-warm steady state).
usage: python tools/bench_conv.py [--sk 1,2,4,8] [--ops fwd,dgrad,wgrad] [--B 64]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphical_gan_amd import functional as F, _lib

SHAPES = {  # name: (Ci, H, Co)
    'A 3->64 @32': (3, 32, 64),
    'B 64->128 @16': (64, 16, 128),
    'C 128->256 @8': (128, 8, 256),
    'F1 3->32 @64': (3, 64, 32),
    'F2 32->64 @32': (32, 32, 64),
}


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sk', default='0')
    ap.add_argument('--ops', default='fwd,dgrad,wgrad')
    ap.add_argument('--B', type=int, default=64)
    ap.add_argument('--shapes', default='A,B,C')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    _lib.load()
    for name, (ci, h, co) in SHAPES.items():
        if name.split()[0] not in args.shapes.split(','):
            continue
        N = args.B
        geom = F.conv_geom(N, ci, h, h, co, 5, 2)
        x = torch.randn(N, ci, h, h, device=dev)
        w = torch.randn(5, 5, ci, co, device=dev) * 0.05
        gy = torch.randn(N, co, geom[5], geom[6], device=dev)
        b = torch.randn(co, device=dev)
        fl = 2.0 * N * co * geom[5] * geom[6] * ci * 25
        for op in args.ops.split(','):
            for sk in args.sk.split(','):
                env = {'fwd': 'GGAN_FWD_SK', 'dgrad': 'GGAN_DGRAD_SK', 'wgrad': 'GGAN_WGRAD_SK'}[op]
                if sk != '0':
                    os.environ[env] = sk
                else:
                    os.environ.pop(env, None)
                if op == 'fwd':
                    fn = lambda: F.ConvFwd.apply(x, w, b, geom, 1, 0.2)
                elif op == 'dgrad':
                    fn = lambda: F.ConvDgrad.apply(gy, w, None, geom, 0, 0.0)
                else:
                    fn = lambda: F.ConvWgrad.apply(x, gy, geom)
                us = timeit(fn)
                L = _lib.load()
                torch.cuda.synchronize(); L.ggan_prof_reset(); L.ggan_prof_enable(1)
                for _ in range(20):
                    fn()
                torch.cuda.synchronize(); L.ggan_prof_enable(0)
                recs = _lib.prof_report(); L.ggan_prof_reset()
                kt = sum(r['total_ms'] for r in recs) * 1e3 / 20
                det = ' '.join('%s=%.1f' % (r['name'].replace('conv_', ''), r['total_ms'] * 1e3 / r['launches']) for r in recs)
                print('%-16s %-6s sk=%-3s wall %7.1f us | kernels %7.1f us %6.1f TF | %s' % (name, op, sk, us, kt, fl / kt / 1e6, det), flush=True)


if __name__ == '__main__':
    main()
