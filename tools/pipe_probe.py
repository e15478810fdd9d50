#!/usr/bin/env python
"""Timing probe: does the NEXT generator step's Extractor/Generator pass hide under the critic step's critic pass + update when
the two are replayed as graphs on two streams?  (Numerics are not meaningful here: the probe shares feed buffers.)"""
import os, sys, time
os.environ['GGAN_FORCE_SPLIT_GRAPH'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
ge.build()
from graphical_gan_amd.engine import Trainer
from graphical_gan_amd.models import Config
mode = sys.argv[1] if len(sys.argv) > 1 else 'ali'
dev = torch.device('cuda:0')
np.random.seed(0)
cfg = Config('cifar10', batch_size=64, mode=mode)
tr = Trainer(cfg, device=dev, graph=True)
ring = tr.model.synthetic_ring(dev, n=8)
def batches():
    i = 0
    while True:
        yield ring[i % 8]; i += 1
bi = batches()
for it in range(8):
    tr.iteration(it, bi)
tr.flush(); torch.cuda.synchronize()
G, D = tr._graphs['gen'], tr._graphs['disc']
print({k: (v is not None) for k, v in G.items() if k.startswith('g')})
S, T = torch.cuda.current_stream(), torch.cuda.Stream()
def seq():
    for r in (G, D):
        r['g0'].replay(); r['g1'].replay()
        if r['g1b'] is not None: r['g1b'].replay()
        r['g2'].replay()
def pipe():
    # gen: nets already in flight on T
    S.wait_stream(T)
    G['g1'].replay()
    if G['g1b'] is not None: G['g1b'].replay()
    G['g2'].replay()
    D['g0'].replay()
    T.wait_stream(S)            # (after the generator step's update; conservative: also after the critic step's nets pass)
    with torch.cuda.stream(T):
        G['g0'].replay()
    D['g1'].replay(); D['g2'].replay()
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.cuda.stream(T):
    G['g0'].replay()
print('sequential  %.4f ms/iteration' % timeit(seq))
print('pipelined   %.4f ms/iteration' % timeit(pipe))
print('sequential  %.4f ms/iteration' % timeit(seq))
