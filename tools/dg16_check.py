#!/usr/bin/env python
"""Dev check of conv_dg16.hip: the 64-pixel x 16/32-channel data gradient against the older kernels (GGAN_DG16=0) and, for the small
cases, the float64 oracle; then timings of both.  usage (GPU box): python tools/dg16_check.py [--time]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graphical_gan_amd import functional as F, _lib

dev = torch.device('cuda:0')
CASES = [  # (N, Ci, H, Co, oracle?)
    (64, 64, 16, 128, False), (128, 64, 16, 128, False), (64, 128, 8, 256, False), (128, 128, 8, 256, False),
    (6, 32, 32, 64, True), (3, 16, 64, 32, True), (5, 32, 8, 16, True), (2, 16, 8, 32, True), (64, 32, 32, 64, False),
    (64, 256, 8, 256, False),
]


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def run(case, kq, masked, act_epi):
    N, Ci, H, Co, use_oracle = case
    g = torch.Generator(device='cpu').manual_seed(sum(case[:4]) + kq)
    geom = F.conv_geom(N, Ci, H, H, Co, 5, 2, 'SAME')
    Ho = geom[5]
    gy = torch.randn(N, Co, Ho, Ho, generator=g).to(dev)
    w = (torch.randn(5, 5, Ci, Co, generator=g) / (25 * Ci) ** .5).to(dev)
    b = torch.randn(Ci, generator=g).to(dev) if act_epi else None
    yref = torch.randn(N, Co, Ho, Ho, generator=g).to(dev)
    os.environ['GGAN_DG16_FORCE'] = '1'
    os.environ['GGAN_DG16_KQ'] = str(kq)

    def call():
        if masked:
            return F.ConvDgradMasked.apply(gy, yref, w, geom, F.ACT_LRELU, 0.2)
        return F.ConvDgrad.apply(gy, w, b, geom, F.ACT_RELU if act_epi else F.ACT_NONE, 0.0)
    os.environ['GGAN_DG16'] = '1'
    L = _lib.load()
    L.ggan_prof_reset(); L.ggan_prof_enable(1)
    new = call()
    torch.cuda.synchronize(); L.ggan_prof_enable(0)
    names = [r['name'] for r in _lib.prof_report()]
    L.ggan_prof_reset()
    os.environ['GGAN_DG16'] = '0'
    old = call()
    torch.cuda.synchronize()
    r = rel(new, old)
    ro = None
    if use_oracle:
        from oracle import ops as O
        gm = gy.cpu().double().numpy()
        if masked:
            gm = gm * np.where(yref.cpu().numpy() > 0, 1.0, 0.2)
        ref = O.conv2d_bwd_data(gm, w.cpu().double().numpy(), (H, H), 2, 'SAME')
        if act_epi:
            ref = np.maximum(ref + b.cpu().double().numpy().reshape(1, -1, 1, 1), 0)
        ro = rel(new.cpu().double(), torch.from_numpy(ref))
    used = any('dg16' in n for n in names)
    ok = used and r < 2e-5 and (ro is None or ro < 2e-5)
    print('%-24s kq %d masked %d epi %d: vs old %.2e vs oracle %s  %s  %s' % (case[:4], kq, masked, act_epi, r, ro, names, 'ok' if ok else 'FAIL'), flush=True)
    return ok


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
    return a.elapsed_time(b) / iters * 1e3


def main():
    ok = True
    for case in CASES:
        for kq in (4, 2):
            if kq == 2 and case[1] % 32:
                continue
            for masked, epi in ((0, 0), (1, 0), (0, 1)):
                ok &= run(case, kq, masked, epi)
    print('ALL OK' if ok else 'FAILURES')
    if '--time' in sys.argv:
        os.environ.pop('GGAN_DG16_FORCE', None)
        for case in CASES:
            N, Ci, H, Co, _ = case
            geom = F.conv_geom(N, Ci, H, H, Co, 5, 2, 'SAME')
            Ho = geom[5]
            gy = torch.randn(N, Co, Ho, Ho, device=dev); w = torch.randn(5, 5, Ci, Co, device=dev) * .05
            yref = torch.randn(N, Co, Ho, Ho, device=dev)
            fl = 2.0 * N * Co * Ho * Ho * Ci * 25
            for masked in (0, 1):
                fn = (lambda: F.ConvDgradMasked.apply(gy, yref, w, geom, F.ACT_LRELU, 0.2)) if masked else (lambda: F.ConvDgrad.apply(gy, w, None, geom, 0, 0.0))
                res = []
                for tgt in ('200', '128'):
                    os.environ['GGAN_TARGET_WGS'] = tgt
                    for mode in ('0', '1:4', '1:2'):
                        os.environ['GGAN_DG16'] = mode[0]
                        os.environ['GGAN_DG16_KQ'] = mode[2:] or '0'
                        os.environ['GGAN_DG16_FORCE'] = '1' if mode != '0' else '0'
                        if mode == '1:2' and Ci % 32:
                            res.append('   -  ')
                            continue
                        us = timeit(fn)
                        res.append('%6.1f' % us)
                print('%-22s masked %d | target 200: old %s kq4 %s kq2 %s | target 128: old %s kq4 %s kq2 %s  (%.2f GF)' % ((case[:4], masked) + tuple(res) + (fl / 1e9,)), flush=True)


if __name__ == '__main__':
    main()
