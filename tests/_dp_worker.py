"""Helper of test_dp_two_ranks_on_one_gpu (launched through torch.distributed.run): a tiny Trainer per rank, every rank on the
same device and the same data/seed, gloo collectives on device tensors.  Rank 0 saves the final weights."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, mode = sys.argv[1], sys.argv[2]
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    import graphical_gan_amd  # noqa: F401
    from graphical_gan_amd.engine import Trainer, broadcast_params
    from graphical_gan_amd.models import Config
    from oracle import nets as N, step as S
    sync = mode.endswith(':syncbn')      # global batch 8 either way: one process x 8, or `world` processes x 8/world
    mode = mode.split(':')[0]
    B = 8 // world if sync else 8
    ocfg = N.Cfg('cifar10', batch_size=8, n_coms=0, dim=8, dim_latent=16)
    cfg = Config('cifar10', batch_size=B, mode=mode, dim=8, dim_latent=16)
    tr = Trainer(cfg, device=dev, graph=True, inject_noise=True, sync_bn=sync)
    tr.load_params(N.init_params(ocfg, 0))
    feeds = [S.make_feed(ocfg, np.random.default_rng(900 + i), mode) for i in range(80)]
    if sync and world > 1:               # this replica's rows of every fed array
        feeds = [{k: (v[rank * B:(rank + 1) * B] if np.ndim(v) and np.shape(v)[0] == 8 else v) for k, v in f.items()} for f in feeds]
    feeds = iter(feeds)
    for it in range(3 if sync else 6):
        if it == 2:
            broadcast_params(0)
        tr.iteration(it, feeds)
    P = tr.get_params()
    torch.cuda.synchronize()
    if world > 1:
        # replicas must be in lock-step
        for n in sorted(P):
            t = torch.as_tensor(P[n]).clone()
            dist.broadcast(t, src=0)
            assert torch.equal(t, torch.as_tensor(P[n])), n
        dist.barrier()
    if rank == 0:
        np.savez(out, **P)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
