#!/usr/bin/env python
"""One rank over RCCL (GGAN_FORCE_ALLREDUCE=1, launched by torch.distributed.run --nproc-per-node 1): train a few one-graph
iterations with the gradient exchange inside the graph and print a checksum of the final weights.  Used by
tests/test_step_gpu.py to compare the two-bucket critic / generator steps with GGAN_ONE_BUCKET=1 and with no exchange at all."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
mode, K = sys.argv[1], int(sys.argv[2])
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
if os.environ.get('GGAN_FORCE_ALLREDUCE'):
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from graphical_gan_amd.models import Config
from graphical_gan_amd.engine import Trainer
np.random.seed(0)
cfg = Config('cifar10', batch_size=16, n_coms=K, mode=mode, dim=16, dim_latent=32)
tr = Trainer(cfg, device=dev, graph=not os.environ.get('CHECK_EAGER'), seed=4321, sync_bn=bool(os.environ.get('CHECK_SYNC_BN')))
ring = tr.model.synthetic_ring(dev, n=5, seed=99)
b = iter(ring * 40)
for it in range(2):
    tr.iteration(it, b)
tr.use_ring(ring)
from graphical_gan_amd import optim as _optim
for it in range(2, 8):
    if it == 7:
        tr._iter_graph = None          # re-capture the iteration once more with the exchange's issue order logged
        _optim.EXCHANGE_LOG[0] = []
    res = tr.iteration(it, b)
    if it == 7:
        log, _optim.EXCHANGE_LOG[0] = _optim.EXCHANGE_LOG[0], None
tr.flush(); torch.cuda.synchronize()
h = hashlib.sha256()
for k, v in sorted(tr.get_params().items()):
    h.update(k.encode()); h.update(np.ascontiguousarray(v).tobytes())
import json
# (the capture's warm-up steps log too: the LAST captured iteration's events are the tail)
print('XLOG ' + json.dumps(log))
print('CHECK dp_graph=%s one_graph=%s sync_bn=%s %s' % (tr.dp_graph, getattr(tr, '_iter_graph', None) is not None, tr.sync_bn, h.hexdigest()))
if dist.is_initialized():
    dist.destroy_process_group()
