"""not-gpu: the N>1 data-parallel path with world_size=2 over gloo (127.0.0.1).

Each rank computes the oracle's gradients on ITS minibatch, packs them into the flat bucket layout the HIP
optimizer uses (optim.layout_slots), runs the product's GradBucket.all_reduce over gloo and applies Adam with
grad_scale = bucket.scale.  Result must equal one process applying Adam to the mean of the two gradients, and
both ranks must end bit-identical (replicated optimizer state, no parameter broadcast after init)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _grads(rank_seed):
    from oracle import nets as N, step as S, tape as tp
    cfg = N.Cfg('face', batch_size=2, n_coms=0, dim=2, dim_latent=4)       # face nets: no BatchNorm
    P0 = N.init_params(cfg, 0)
    feed = S.make_feed(cfg, np.random.default_rng(rank_seed))
    Pt = {k: tp.T(v.astype(np.float64)) for k, v in P0.items()}
    out = S.forward(cfg, Pt, feed, 'ali')
    names = sorted(N.trainable(N.params_with_name(P0, 'Discriminator')))
    gs = tp.grad(out['disc_cost'], [Pt[n] for n in names])
    return names, P0, [g.v for g in gs]


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from graphical_gan_amd.optim import GradBucket, layout_slots
    from oracle import ops as O
    names, P0, gs = _grads(100 + rank)
    slots, total = layout_slots([g.size for g in gs])
    flat = torch.zeros(total, dtype=torch.float64)
    for (o, n), g in zip(slots, gs):
        flat[o:o + n] = torch.as_tensor(g.reshape(-1))
    bucket = GradBucket(flat)
    assert bucket.world == world and bucket.scale == 1.0 / world
    # the engine's exchange pattern: two sub-buckets, each asynchronous (the first overlaps the rest of the backward pass),
    # waited for before the optimizer reads the buffer; must equal one all-reduce of the whole bucket
    whole = flat.clone()
    dist.all_reduce(whole)
    cut = slots[len(slots) // 2][0]
    w1 = bucket.all_reduce(async_op=True, lo=0, hi=cut)
    w2 = bucket.all_reduce(async_op=True, lo=cut, hi=None)
    assert w1 is not None and w2 is not None
    w1.wait(); w2.wait()
    assert torch.equal(flat, whole)
    theta = np.concatenate([P0[n].astype(np.float64).reshape(-1) for n in names])
    g = np.concatenate([flat[o:o + n].numpy() for o, n in slots]) * bucket.scale
    th, m, v = O.adam_update(theta, g, np.zeros_like(theta), np.zeros_like(theta), 1, 2e-4, .5, .999)
    # cross-replica BatchNorm statistics: the host side gathers every replica's [2][C] rows in rank order; merged with the
    # equal-count form of Chan's update (csrc/bn.hip bn_apply_sync_k) they must equal the statistics of the whole batch
    from graphical_gan_amd.functional import _all_gather_rows
    full = np.random.default_rng(5).standard_normal((world * 6, 7)) * 3 + 1
    mine = full[rank * 6:(rank + 1) * 6]
    st = torch.as_tensor(np.stack([mine.mean(0), ((mine - mine.mean(0)) ** 2).sum(0)]))
    rows = _all_gather_rows(st, dist.group.WORLD).numpy()
    assert rows.shape == (world, 2, 7) and np.array_equal(rows[rank], st.numpy())
    mean = rows[:, 0].mean(0)
    var = (rows[:, 1].sum(0) + 6 * ((rows[:, 0] - mean) ** 2).sum(0)) / (6 * world)
    assert np.abs(mean - full.mean(0)).max() < 1e-12 and np.abs(var - full.var(0)).max() < 1e-12
    q.put((rank, th))
    dist.barrier()
    dist.destroy_process_group()


def test_two_replica_gradient_exchange_matches_global_mean():
    from oracle import ops as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(res[0], res[1])                                  # replicas stay in lock-step
    names, P0, g0 = _grads(100)
    _, _, g1 = _grads(101)
    theta = np.concatenate([P0[n].astype(np.float64).reshape(-1) for n in names])
    g = np.concatenate([(a + b).reshape(-1) / 2 for a, b in zip(g0, g1)])
    th, _, _ = O.adam_update(theta, g, np.zeros_like(theta), np.zeros_like(theta), 1, 2e-4, .5, .999)
    assert np.abs(res[0] - th).max() < 1e-12
