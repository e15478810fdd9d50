"""Do HIP events recorded INSIDE a captured graph (event-record nodes) report elapsed times after a replay?  (bench.py: per-kernel
times of the graph-replayed step, live)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphical_gan_amd import functional as F, _lib
dev = torch.device('cuda:0')
L = _lib.load()
N, ci, h, co = 64, 64, 16, 128
geom = F.conv_geom(N, ci, h, h, co, 5, 2)
x = torch.randn(N, ci, h, h, device=dev); w = torch.randn(5, 5, ci, co, device=dev) * .05
gy = torch.randn(N, co, 8, 8, device=dev)
def body():
    y = F.ConvFwd.apply(x, w, None, geom, 0, 0.0)
    g = F.ConvDgrad.apply(gy, w, None, geom, 0, 0.0)
    return y, g
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): body()
torch.cuda.synchronize()
L.ggan_prof_reset(); L.ggan_prof_enable(1)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    out = body()
L.ggan_prof_enable(0)
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    try:
        print(i, [(r['name'], r['launches'], round(r['total_ms'] * 1e3, 2)) for r in _lib.prof_report()])
    except Exception as e:
        print('report failed', e)
