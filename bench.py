#!/usr/bin/env python
"""bench.py -- images/sec of the Graphical-GAN training iteration on N MI355X GPUs (one process per GPU).

A "step" is one iteration of the reference loop (gmgan_inference_cifar10.py:480-494): one generator+extractor
session.run followed by CRITIC_ITERS critic session.runs, each on a fresh synthetic minibatch + fresh noise,
each including forward, backward and the TF-Adam update.  Workload at N=1: BASELINE.json configs[1],
gan_inference_cifar10.py (32x32x3, BATCH_SIZE=64, MODE='ali'); per-GPU batch stays 64 as N grows (weak scaling).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def algorithmic_gflop_per_iteration(cfg):
    """GEMM-like layers only, 2*M*N*K, forward + exactly the gradients the step needs."""
    B, d, nl = cfg.B, cfg.dim, cfg.nl
    chans = [cfg.C] + [d * 2 ** i for i in range(nl)]
    conv = []
    s = cfg.S
    for i in range(nl):
        s //= 2 if cfg.dataset != 'mnist' else 1
        if cfg.dataset == 'mnist':
            s = {0: 14, 1: 7, 2: 4}[i]
        conv.append(2.0 * B * chans[i + 1] * s * s * chans[i] * 25)
    if cfg.dataset == 'mnist':
        deconv = [2.0 * B * 256 * 16 * 128 * 25, 2.0 * B * 128 * 49 * 64 * 25, 2.0 * B * 64 * 196 * 1 * 25]
    else:
        deconv = list(reversed(conv))                # the generator mirrors the extractor
    lin = lambda i, o: 2.0 * B * i * o
    fE = sum(conv) + lin(cfg.flat, cfg.dim_latent)
    fG = sum(deconv) + lin(cfg.dim_latent, cfg.flat)
    fDconv = sum(conv)
    fz1, fzx1, fout = lin(cfg.dim_latent, 512), lin(cfg.flat + 512, 512), lin(512, 1)
    fDlin = fz1 + fzx1 + fout
    fH = fHin = 0.0
    if cfg.K:
        fHin = lin(cfg.dim_latent + cfg.K, 512)
        fH = fHin + 2 * lin(512, 512) + lin(512, 1)
    fwd = fE + fG + 2 * (fDconv + fDlin) + 2 * fH
    gen_bwd = (2 * fE - conv[0]) + (2 * fG - (0 if cfg.K else lin(cfg.dim_latent, cfg.flat))) + fDconv + 2 * fDlin + 2 * fH
    disc_bwd = 2 * (fDconv + fDlin) + 2 * (fDconv - conv[0] + fzx1 + fout) + 2 * fH + 2 * (fH - fHin)
    total = (fwd + gen_bwd) + cfg.critic_iters * (fwd + disc_bwd)
    if cfg.mode == 'wali-gp':
        # extra critic pass + x-gradient + double backward ~ 6x one critic-branch forward per critic step
        total += cfg.critic_iters * 6 * (fDconv + fDlin)
    return total / 1e9


def host_cores():
    """CPUs this process may actually use: scheduler affinity capped by the cgroup CPU quota (the GPU boxes expose 256 logical
    CPUs but grant 16; running the CPU baseline on 256 threads measured 0.6 img/s instead of 670)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def ssgan_gflop_per_iteration(cfg):
    """ssgan_inference_moving_mnist.py (MODE local_ep, mean-field posterior): GEMM-like layers, forward + exactly the
    gradients each step needs (critic frozen in the generator step; no data-gradient into input frames)."""
    B, L, d = cfg.B, cfg.LEN, cfg.dim
    F = B * L
    chans, sizes = [getattr(cfg, 'C', 1), d, 2 * d, 4 * d, 8 * d], [32, 16, 8, 4]
    conv = [2.0 * chans[i + 1] * sizes[i] ** 2 * chans[i] * 25 for i in range(4)]             # per image
    convG1 = 2.0 * d * 32 ** 2 * L * getattr(cfg, 'C', 1) * 25                                 # G_Extractor layer 1 (C*LEN channels)
    lin = lambda n, i, o: 2.0 * n * i * o
    zin = cfg.dim_g + cfg.dim_l + cfg.n_c
    fE = F * sum(conv) + lin(F, cfg.flat + cfg.n_c, cfg.dim_l)
    fEG = B * (convG1 + sum(conv[1:])) + lin(B, cfg.flat + cfg.n_c, cfg.dim_g)
    fG = F * sum(conv) + lin(F, zin, cfg.flat)                                                 # the frame generator mirrors the extractor
    fDc = F * sum(conv)
    fDl = lin(F, zin, 512) + lin(F, cfg.flat + 512 + cfg.n_c, 512) + lin(F, 512, 1)
    mlp = lambda n, i: lin(n, i, 512) + 2 * lin(n, 512, 512) + lin(n, 512, 1)
    fDs = mlp((L - 1) * B, 2 * cfg.dim_l) + mlp(B, cfg.dim_g)
    fdyn = (L - 1) * (lin(B, cfg.dim_l + cfg.dim_t, cfg.dim_op) + lin(B, cfg.dim_op, cfg.dim_op) + lin(B, cfg.dim_op, cfg.dim_l))
    fwd = fE + fEG + fG + fdyn + 2 * (fDc + fDl + fDs)
    gen_bwd = (2 * fE - F * conv[0]) + (2 * fEG - B * convG1) + 2 * fG + 2 * fdyn + fDc + 2 * fDl + 2 * fDs
    disc_bwd = 2 * (2 * (fDc + fDl + fDs) - F * conv[0])
    return ((fwd + gen_bwd) + cfg.critic_iters * (fwd + disc_bwd)) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--dataset', default='cifar10')
    ap.add_argument('--mode', default='ali', help="ali | wali-gp | local_ep (gmgan, N_COMS=30)")
    ap.add_argument('--ssgan-mode', default='local_ep', help="moving_mnist / chairs: local_ep | local_epce-z | ali | alice-z, optionally "
                    "':concat_x' | ':concat_z' | ':3dcnn' (the sequence critic of ali / alice-z)")
    ap.add_argument('--n-coms', type=int, default=None, help='mixture components of the gmgan prior (default: the script value, 30 / 50 / 100)')
    ap.add_argument('--batch-size', type=int, default=None, help='per-GPU minibatch (default 64; 32 sequences for moving_mnist)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-fuse', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--cpu-iters', type=int, default=2)
    ap.add_argument('--host-feed', action='store_true',
                    help='minibatches start in host memory (pinned double-buffered H->D copies): the PCIe-inclusive rate')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (there is no CPU path); use gpurun')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' %
                         (args.gpus, world, args.gpus))
    # (GGAN_DIST_BACKEND=gloo + fewer devices than ranks: a control-flow rehearsal of the N>1 path on a single-GPU box)
    backend = os.environ.get('GGAN_DIST_BACKEND', 'nccl')
    local_dev = local_rank if backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    ge.build()
    from graphical_gan_amd import _lib
    from graphical_gan_amd.engine import Trainer, broadcast_params
    from graphical_gan_amd.models import Config

    ssgan = args.dataset in ('moving_mnist', 'chairs')
    if args.batch_size is None:
        args.batch_size = (32 if args.dataset == 'moving_mnist' else 16) if ssgan else 64
    np.random.seed(0)                                  # reference initialisers draw from numpy's global RNG
    if ssgan:                                          # BASELINE configs[4]: ssgan_inference_moving_mnist.py, T=16
        from graphical_gan_amd.models_ssgan import SSConfig, StateSpaceGAN
        K = 0
        args.mode, _, ali_mode = args.ssgan_mode.partition(':')
        skw = dict(mode=args.mode, ali_mode=ali_mode or 'concat_x')
        if args.dataset == 'chairs':                   # ssgan_inference_chairs.py: 31 RGB views, no labels, res_w operator
            cfg = SSConfig(batch_size=args.batch_size, fuse=not args.no_fuse, length=31, n_c=0, channels=3, op_dyn_mode='res_w',
                           dataset='chairs', **skw)
        else:
            cfg = SSConfig(batch_size=args.batch_size, fuse=not args.no_fuse, **skw)
        model = StateSpaceGAN(cfg)
        tr = Trainer(cfg, device=dev, graph=not args.no_graph, seed=1234 + rank, model=model)
    else:
        # N_COMS of the gmgan scripts: 30 (cifar10 :78, mnist), 50 (svhn :72), 100 (face :68)
        K = ({'svhn': 50, 'face': 100}.get(args.dataset, 30)) if args.mode in ('local_ep', 'local_epce') else 0
        if K and args.n_coms:
            K = args.n_coms
        # the code-space objectives run with the scripts' own settings for them: DIM_LATENT = 8, BN_FLAG = False (gan_inference_cifar10.py:72-77)
        code = args.mode in ('vegan', 'vegan-wgan-gp', 'vegan-kl', 'vegan-ikl', 'vegan-jsd')
        cfg = Config(args.dataset, batch_size=args.batch_size, n_coms=K, mode=args.mode, fuse=not args.no_fuse,
                     **(dict(dim_latent=8, bn=False) if code else {}))
        tr = Trainer(cfg, device=dev, graph=not args.no_graph, seed=1234 + rank)
    torch.manual_seed(1234 + rank)
    ring = tr.model.synthetic_ring(dev, n=4 if ssgan else 8, seed=1234 + rank)

    def batches():
        i = 0
        while True:
            yield ring[i % len(ring)]
            i += 1
    bi = batches()
    if args.host_feed:                                 # same minibatches, but every one crosses PCIe (graphical_gan_amd/data.py)
        from graphical_gan_amd.data import DevicePrefetcher
        host_ring = [tuple(t.cpu().numpy() for t in b) if isinstance(b, tuple) else b.cpu().numpy() for b in ring]
        bi = DevicePrefetcher(lambda: iter(host_ring), dev, depth=2)
    it = 0
    tr.iteration(it, bi); it += 1                      # creates parameters + optimizers (eager)
    tr.iteration(it, bi); it += 1
    broadcast_params(0)
    for _ in range(max(args.warmup, 2)):               # includes graph capture
        tr.iteration(it, bi); it += 1

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    tr.flush()
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = tr.iteration(it, bi); it += 1
    tr.flush()                                         # (DP: the last critic step's exchange + Adam belong to the timed work)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = all(np.isfinite(float(v)) for v in last.values())

    ms_per_step = 1e3 * dt / args.steps
    images_per_s = cfg.B * world * args.steps / dt
    gflop_it = ssgan_gflop_per_iteration(cfg) if ssgan else algorithmic_gflop_per_iteration(cfg)
    step_tflops = gflop_it * args.steps / dt / 1e3      # per GPU

    # ---- per-kernel timing of the same step with HIP events (eager, on the launch stream) ---------------------
    roofline = None
    kernels = None
    if not args.no_kernel_profile:
        # EVERY rank runs these eager iterations (they contain the gradient all-reduce: a collective issued by rank 0 alone
        # would hang the job); only rank 0 records and reports
        tr.flush()
        tr_graph = tr.graph_enabled
        tr.graph_enabled = False
        L = _lib.load()
        for _ in range(2):
            tr.iteration(it, bi); it += 1
        torch.cuda.synchronize(dev)
        L.ggan_prof_reset(); L.ggan_prof_enable(1 if rank == 0 else 0)
        n_prof = 5
        for _ in range(n_prof):
            tr.iteration(it, bi); it += 1
        torch.cuda.synchronize(dev)
        L.ggan_prof_enable(0)
        recs = _lib.prof_report() if rank == 0 else []
        L.ggan_prof_reset()
        tr.graph_enabled = tr_graph
    if rank == 0 and not args.no_kernel_profile:
        recs.sort(key=lambda r: -r['total_ms'])
        kernels = [dict(name=r['name'], launches_per_iter=r['launches'] / n_prof,
                        ms_per_iter=round(r['total_ms'] / n_prof, 4),
                        avg_us=round(1e3 * r['total_ms'] / r['launches'], 2),
                        tflops=round(r['flops'] / (r['total_ms'] * 1e-3) / 1e12, 2) if r['flops'] else None)
                   for r in recs]
        dom = next((r for r in recs if r['flops'] > 0), None)
        if dom is not None:
            ach = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
            # HBM traffic per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this same
            # workload, committed by tools/prof_round.sh; counters cannot be read from inside the process
            traffic, pmc = None, None
            try:
                tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')))
                pmc = tab.get(dom['name'])
                traffic = pmc['traffic_bytes'] if pmc else None
            except (OSError, ValueError):
                pass
            roofline = dict(bound='mfma', kernel=dom['name'], achieved=round(ach, 2), peak=MFMA_F32_PEAK_TFLOPS,
                            unit='TFLOP/s', frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4), traffic=traffic,
                            traffic_unit='bytes/launch (rocprofv3 PMC, profiles/pmc_traffic.json)', pmc=pmc,
                            avg_launch_us=round(1e3 * dom['total_ms'] / dom['launches'], 2),
                            flop_per_launch=dom['flops'] / dom['launches'],
                            whole_step_tflops=round(step_tflops, 2),
                            whole_step_frac=round(step_tflops / MFMA_F32_PEAK_TFLOPS, 4))
    if world > 1:
        dist.barrier()

    # ---- CPU baseline: the numpy oracle (a port; the reference is Python2+TF1 and cannot run) ------------------
    cpu = None
    blas_limit = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from threadpoolctl import threadpool_limits
            blas_limit = threadpool_limits(limits=host_cores())
        except ImportError:
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline and ssgan:
        from oracle import ssgan as OSS
        ob = 2                                         # bounded sample: 2 sequences per minibatch instead of cfg.B
        ocfg = OSS.Cfg(batch_size=ob, length=cfg.LEN, n_c=cfg.n_c, channels=cfg.C, op_dyn_mode=cfg.op_dyn_mode)
        otr = OSS.Trainer(ocfg, OSS.init_params(ocfg, 0), np.float32)
        feeds = iter([OSS.make_feed(ocfg, np.random.default_rng(i)) for i in range(8)])
        otr.iteration(1, feeds)
        t1 = time.perf_counter()
        otr.iteration(2, feeds)
        cdt = time.perf_counter() - t1
        cpu = dict(value=round(ob / cdt, 3), unit='sequences/sec', cores=host_cores(), kind='port',
                   sample='1 iteration (gen step + critic step) at %d sequences x 16 frames per minibatch, numpy fp32 '
                   'oracle with multi-threaded BLAS' % ob)
    elif rank == 0 and world == 1 and not args.no_cpu_baseline and args.dataset == 'cifar10' and args.mode == 'ali':
        # the same step on the host cores with PyTorch-CPU (oneDNN convolutions, every core): the closest stand-in for an
        # optimised TensorFlow-CPU build of the reference, which cannot run here (oracle/torch_cpu.py)
        from oracle import nets as ON, step as OS, torch_cpu
        torch.set_num_threads(host_cores())
        ocfg = ON.Cfg('cifar10', batch_size=cfg.B)
        ts = torch_cpu.Step(ocfg, ON.init_params(ocfg, 0), torch.float32)
        rng = np.random.default_rng(0)

        def cpu_iteration():
            for which in ('gen', 'disc'):
                f = OS.make_feed(ocfg, rng, 'ali')
                ts.step(which, OS.real_x_from_feed(ocfg, f, np.float32), f['p_z_noise'])
        cpu_iteration()                                # warm-up (thread pool, oneDNN primitive cache)
        n_cpu, t1 = 0, time.perf_counter()
        while n_cpu < 3 or (time.perf_counter() - t1 < 10.0 and n_cpu < 2000):
            cpu_iteration()
            n_cpu += 1
        cdt = time.perf_counter() - t1
        cpu = dict(value=round(cfg.B * n_cpu / cdt, 1), unit='images/sec', cores=torch.get_num_threads(), kind='port',
                   sample='%d iterations (gen step + critic step) of the same workload in %.1f s, PyTorch-CPU fp32 restatement '
                   '(oneDNN convolutions, TF-SAME padding and TF-Adam emulated)' % (n_cpu, cdt))
    elif rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import nets as ON, step as OS
        ocfg = ON.Cfg(args.dataset, batch_size=cfg.B, n_coms=K)
        otr = OS.Trainer(ocfg, ON.init_params(ocfg, 0), 'wali-gp' if args.mode == 'wali-gp' else 'ali', np.float32)
        feeds = iter([OS.make_feed(ocfg, np.random.default_rng(i), otr.mode) for i in range(64)])
        otr.iteration(1, feeds)                        # warm numpy/BLAS
        t1 = time.perf_counter()
        for j in range(args.cpu_iters):
            otr.iteration(2 + j, feeds)
        cdt = time.perf_counter() - t1
        cpu = dict(value=round(cfg.B * args.cpu_iters / cdt, 2), unit='images/sec', cores=host_cores(),
                   kind='port', sample='%d iterations (gen step + %d critic step(s)) of the same workload, numpy fp32 '
                   'oracle with multi-threaded BLAS' % (args.cpu_iters, cfg.critic_iters))

    if rank == 0:
        out = {
            'metric': ('sequences/sec (G+D step) %s T=%d 64x64 bs=%d' % (args.dataset, cfg.LEN, cfg.B)) if ssgan else
                      'images/sec (G+D%s step) %s %dx%d bs=%d' % ('+GP' if args.mode == 'wali-gp' else '', args.dataset, cfg.S, cfg.S, cfg.B),
            'value': round(images_per_s, 1), 'unit': 'sequences/sec' if ssgan else 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('ssgan_inference_%s.py MODE=%s POS_MODE=%s LEN=%d BATCH_SIZE=%d sequences '
                                    '(per GPU) of 64x64x%d frames%s' % (args.dataset, cfg.mode + ((' ALI_MODE=' + cfg.ali_mode) if cfg.seq_critic else ''),
                                                                        cfg.pos_mode, cfg.LEN, cfg.B, cfg.C,
                                                                        ', eager' if args.no_graph else ''))
                       if ssgan else '%s_inference_%s.py MODE=%s%s BATCH_SIZE=%d (per GPU) %dx%dx%d, CRITIC_ITERS=%d%s' % (
                'gmgan' if K else 'gan', args.dataset, args.mode, ' N_COMS=%d' % K if K else '', cfg.B, cfg.S, cfg.S, cfg.C,
                cfg.critic_iters, '' if not args.no_graph else ', eager'),
                'parallelism': 'dp%d' % world, 'global_batch': cfg.B * world, 'hip_graph': not args.no_graph, 'host_feed': bool(args.host_feed),
                'fused_epilogues': not args.no_fuse, 'minibatches_per_step': 1 + cfg.critic_iters,
                'algorithmic_gflop_per_step': round(gflop_it, 2), 'finite_costs': bool(finite),
                **({'frames_per_sec': round(images_per_s * cfg.LEN, 1)} if ssgan else {})},
            'roofline': roofline, 'cpu_baseline': cpu, 'kernels': kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
