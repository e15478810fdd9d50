"""Per-workgroup timeline of the corr kernel (GGAN_DBG=4): prints median cycle deltas between stamps."""
import os, sys
os.environ['GGAN_DBG'] = os.environ.get('GGAN_DBG', '4')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from graphical_gan_amd import functional as F, _lib
dev = torch.device('cuda:0')
op = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
ci, h, co = (64, 16, 128) if (len(sys.argv) < 3 or sys.argv[2] == 'B') else (128, 8, 256)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 64
geom = F.conv_geom(N, ci, h, h, co, 5, 2)
x = torch.randn(N, ci, h, h, device=dev); w = torch.randn(5, 5, ci, co, device=dev) * .05
gy = torch.randn(N, co, geom[5], geom[6], device=dev); b = torch.randn(co, device=dev)
ws = F.workspace(dev)
fn = {'fwd': lambda: F.ConvFwd.apply(x, w, b, geom, 1, 0.2), 'dgrad': lambda: F.ConvDgrad.apply(gy, w, None, geom, 0, 0.0),
      'wgrad': lambda: F.ConvWgrad.apply(x, gy, geom)}[op]
if len(sys.argv) > 3: N = int(sys.argv[3])
for _ in range(3): fn()
torch.cuda.synchronize()
st = ws[-(32 << 20):].view(torch.int64)
st.zero_(); torch.cuda.synchronize()
fn(); torch.cuda.synchronize()
a = st.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]
# (round 5: s_memtime is per XCD -- the eight counters are ~1e12 ticks apart -- so a row is compared with its own start only, and rows
#  left over from another launch are those whose duration is absurd)
a = a[(a[:, 14] - a[:, 0] > 0) & (a[:, 14] - a[:, 0] < 10_000_000)]
print(op, sys.argv[2:] , 'workgroups', len(a))
names = ['start', 'descr', 'commit0', 'bar0'] + ['chunk%d' % i for i in range(8)] + ['loop_end', 'reduced', 'stored', 'x15']
for i, nme in enumerate(names):
    col = a[:, i]
    ok = col != 0
    if not ok.any() or i == 0: continue
    rel = col[ok] - a[ok, 0]
    print('%-9s since WG start: median %7d  max %7d' % (nme, np.median(rel), rel.max()))
print('WG duration median %d max %d' % (np.median(a[:, 14] - a[:, 0]), (a[:, 14] - a[:, 0]).max()))
# wall-clock of the same launch (HIP events) -> shader clock during the kernel
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
os.environ['GGAN_DBG'] = '0'
for _ in range(3): fn()
torch.cuda.synchronize(); e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
print('launch wall %.2f us (back to back, no stamps)' % (e0.elapsed_time(e1) * 1e3 / 20))
