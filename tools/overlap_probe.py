#!/usr/bin/env python
"""Do the filter-gradient and the data-gradient kernel of one layer gain from running on two streams?  Captures REPS pairs in a
HIP graph, once on one stream and once forked, and prints the time per pair.  usage: python tools/overlap_probe.py [B|C] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphical_gan_amd import functional as F, _lib

dev = torch.device('cuda:0')
_lib.load()
shape = sys.argv[1] if len(sys.argv) > 1 else 'B'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ci, h, co = {'B': (64, 16, 128), 'C': (128, 8, 256)}[shape]
geom = F.conv_geom(N, ci, h, h, co, 5, 2)
x = torch.randn(N, ci, h, h, device=dev)
w = torch.randn(5, 5, ci, co, device=dev) * .05
gy = torch.randn(N, co, geom[5], geom[6], device=dev)
REPS = 20
side = torch.cuda.Stream()


def pair(fork):
    if not fork:
        F.ConvWgrad.apply(x, gy, geom)
        F.ConvDgrad.apply(gy, w, None, geom, 0, 0.0)
        return
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        F.ConvWgrad.apply(x, gy, geom)
    F.ConvDgrad.apply(gy, w, None, geom, 0, 0.0)
    cur.wait_stream(side)


s = torch.cuda.Stream()
for fork in (False, True, False, True):
    with torch.cuda.stream(s):
        for _ in range(3):
            pair(fork)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                pair(fork)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print('%s N=%d fork=%d: %.1f us per (wgrad + dgrad) pair' % (shape, N, fork, a.elapsed_time(b) / 10 / REPS * 1e3))
