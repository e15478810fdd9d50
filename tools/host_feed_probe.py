import time, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphical_gan_amd.engine import Trainer
from graphical_gan_amd.models import Config
from graphical_gan_amd import functional as F
dev = torch.device('cuda:0')
cfg = Config('cifar10', batch_size=64, mode='ali')
np.random.seed(0)
tr = Trainer(cfg, device=dev, graph=True)
host = [np.random.randint(0, 256, size=(64, 3072)).astype(np.int32) for _ in range(8)]
mode = sys.argv[1] if len(sys.argv) > 1 else 'pin'
if mode == 'pin':
    pinned = [torch.empty((64, 3072), dtype=torch.int32).pin_memory() for _ in range(8)]
else:   # ordinary cached host memory, registered with the runtime (device-addressable at the same address)
    _keep = [np.empty((64, 3072), dtype=np.int32) for _ in range(8)]
    rt = torch.cuda.cudart()
    for a in _keep:
        r = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 2 if mode == 'mapped' else 0)
        print('register ->', r)
    pinned = [torch.from_numpy(a) for a in _keep]
pinned_np = [t.numpy() for t in pinned]
devb = [torch.empty((64, 3072), dtype=torch.int32, device=dev) for _ in range(8)]
T = {'pin': 0.0, 'pack': 0.0, 'd2d': 0.0, 'step': 0.0}
k = [0]
def feed_one():
    i = k[0] % 8; k[0] += 1
    t0 = time.perf_counter()
    np.copyto(pinned_np[i], host[i])
    t1 = time.perf_counter()
    n = pinned[i].numel()
    F.pack_([pinned[i].view(torch.float32).reshape(-1)], [(0, n)], devb[i].view(torch.float32).reshape(-1))
    t2 = time.perf_counter()
    tr.set_batch(devb[i])
    t3 = time.perf_counter()
    T['pin'] += t1 - t0; T['pack'] += t2 - t1; T['d2d'] += t3 - t2
def it(which_first):
    if which_first:
        feed_one(); t = time.perf_counter(); tr.step('gen'); T['step'] += time.perf_counter() - t
    feed_one(); t = time.perf_counter(); tr.step('disc'); T['step'] += time.perf_counter() - t
for i in range(8):
    it(i > 0)
torch.cuda.synchronize()
for key in T: T[key] = 0.0
t = time.perf_counter()
for i in range(100):
    it(True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('host %.3f total %.3f ms/it; per-it ms:' % ((t1 - t) * 10, (t2 - t) * 10), {a: round(b * 10, 3) for a, b in T.items()})
