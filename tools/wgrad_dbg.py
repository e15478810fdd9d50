"""four-wave filter gradient against the eight-wave kernel on the same operands: where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphical_gan_amd import functional as F
dev = torch.device('cuda:0')
for (N, Ci, H, Co) in [(4, 8, 16, 64), (4, 16, 16, 32), (2, 16, 16, 32), (64, 64, 16, 128), (8, 16, 8, 32), (6, 32, 32, 64), (2, 16, 64, 32)]:
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(N, Ci, H, H, generator=g).to(dev); gy = torch.randn(N, Co, H // 2, H // 2, generator=g).to(dev)
    geom = F.conv_geom(N, Ci, H, H, Co, 5, 2, 'SAME')
    os.environ['GGAN_WGRAD_W4'] = '0'; a = F.ConvWgrad.apply(x, gy, geom).cpu().numpy()
    os.environ['GGAN_WGRAD_W4'] = '1'; b = F.ConvWgrad.apply(x, gy, geom).cpu().numpy()
    d = np.abs(a - b)
    print((N, Ci, H, Co), 'max diff %.3g of %.3g' % (d.max(), np.abs(a).max()))
    if d.max() > 1e-3 * np.abs(a).max():
        bad = d > 1e-3 * np.abs(a).max()
        print('  bad taps kh:', sorted(set(np.where(bad)[0])), 'kw:', sorted(set(np.where(bad)[1])), 'ci:', sorted(set(np.where(bad)[2]))[:20], 'co:', sorted(set(np.where(bad)[3]))[:20])
        print('  ratio sample', (b[bad][:6] / a[bad][:6]))
