#!/usr/bin/env python
"""Time ggan_gemm on the Linear-layer shapes of the hot path (graph replay of 20 back-to-back launches, so the number
includes the per-kernel floor a step graph pays).  usage: python tools/bench_gemm.py [sk ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from graphical_gan_amd import functional as F

dev = torch.device('cuda:0')
SHAPES = [  # (ta, tb, M, N, K) as issued by fwd / dX / dW of the critic + generator/extractor Linears
    (0, 0, 64, 512, 4608), (0, 0, 128, 512, 4608), (0, 0, 64, 128, 4096), (0, 0, 64, 512, 512), (0, 0, 128, 512, 128),
    (0, 0, 64, 4096, 128), (0, 0, 128, 512, 512), (0, 0, 128, 512, 158), (0, 1, 128, 512, 512), (0, 1, 128, 158, 512), (0, 1, 128, 128, 512), (0, 1, 64, 4608, 512), (0, 1, 128, 4608, 512), (0, 1, 64, 512, 512), (0, 1, 64, 4096, 128),
    (1, 0, 4608, 512, 128), (1, 0, 512, 512, 64), (1, 0, 4096, 128, 64), (1, 0, 128, 4096, 64),
]
sks = sys.argv[1:] or ['auto']
for (ta, tb, M, N, K) in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    res = []
    for sk in sks:
        if sk == 'auto':
            os.environ.pop('GGAN_GEMM_SK', None)
        else:
            os.environ['GGAN_GEMM_SK'] = sk
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.no_grad():
                for _ in range(3):
                    F.Gemm.apply(a, b, None, bool(ta), bool(tb), 0, 0.0)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                with torch.no_grad():
                    for _ in range(20):
                        out = F.Gemm.apply(a, b, None, bool(ta), bool(tb), 0, 0.0)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                g.replay()
            e1.record(); torch.cuda.synchronize()
        res.append('%s=%.1f' % (sk, e0.elapsed_time(e1) * 1e3 / 200))
    print('ta%d tb%d M%-5d N%-5d K%-5d  us/launch(+reduce): %s' % (ta, tb, M, N, K, '  '.join(res)))

# fused Linear backward (activation derivative applied on the operand load) vs act_bwd + plain GEMM
import ctypes as C
from graphical_gan_amd import _lib
L = _lib.load()
def timeit(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 200
for (M, N, K) in [(64, 512, 4608), (128, 512, 4608), (64, 512, 512), (128, 512, 128)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev); y = torch.randn(M, N, device=dev); g = torch.randn(M, N, device=dev)
    dx = torch.empty(M, K, device=dev); dw = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev); g2 = torch.empty_like(g)
    ws = F.workspace(dev) if False else None
    def P(t): return C.c_void_p(t.data_ptr())
    def st(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def fused_dx():
        wsb = F.workspace(dev); L.ggan_linear_bwd_data_act(M, N, K, P(g), P(y), 1, 0.2, P(w), P(dx), P(wsb), wsb.numel(), st())
    def fused_dw():
        wsb = F.workspace(dev); L.ggan_linear_bwd_weight_act(M, N, K, P(x), P(g), P(y), 1, 0.2, P(dw), P(db), P(wsb), wsb.numel(), st())
    def plain_dx():
        wsb = F.workspace(dev); L.ggan_gemm(0, 1, M, K, N, P(g2), P(w), None, P(dx), 0, 0.0, P(wsb), wsb.numel(), st())
    def plain_dw():
        wsb = F.workspace(dev); L.ggan_gemm_colsum(1, K, N, M, P(x), P(g2), P(dw), P(db), P(wsb), wsb.numel(), st())
    def actb():
        L.ggan_act_bwd(P(g), P(y), P(g2), g.numel(), 1, 0.2, st())
    print('linear bwd M%d N%d K%d: fused dx %.1f dw %.1f | plain dx %.1f dw %.1f act_bwd %.1f' % (
        M, N, K, timeit(fused_dx), timeit(fused_dw), timeit(plain_dx), timeit(plain_dw), timeit(actb)))
