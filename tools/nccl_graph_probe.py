#!/usr/bin/env python
"""Does an RCCL all-reduce captured in a HIP graph survive the process group's watchdog thread?  One rank is enough:
MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 python tools/nccl_graph_probe.py [sync|async|selftest]"""
import os, sys, time
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29555')
import torch, torch.distributed as dist
mode = sys.argv[1] if len(sys.argv) > 1 else 'sync'
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, **({} if os.environ.get('PROBE_NO_DEVICE_ID') else {'device_id': dev}))
if mode == 'selftest':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from graphical_gan_amd.engine import dp_graph_selftest
    from graphical_gan_amd import rccl
    print('selftest ->', dp_graph_selftest(dev, rccl.get(dev))); time.sleep(3); print('alive after 3 s'); sys.exit(0)
t = torch.ones(1 << 20, device=dev)
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(s):
    dist.all_reduce(t.clone()); torch.cuda.synchronize()
    time.sleep(float(os.environ.get('PROBE_SLEEP', '0')))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
        u = t * 1.0
        if mode == 'sync':
            dist.all_reduce(u)
        else:
            w = dist.all_reduce(u, async_op=True); w.wait()
        v = u + 1.0
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
print(mode, 'value', float(v[0])); time.sleep(3); print('alive after 3 s')
