#!/usr/bin/env python
"""bench.py -- images/sec of the Graphical-GAN training iteration on N MI355X GPUs (one process per GPU).

A "step" is one iteration of the reference loop (gan_inference_cifar10.py:480-494): one generator+extractor session.run followed by
CRITIC_ITERS critic session.runs, each on a fresh synthetic minibatch + fresh noise, each including forward, backward and the TF-Adam
update.  Headline workload = BASELINE.json's metric, "images/sec (G+D+GP step) CIFAR-10 32x32 bs=64": BASELINE configs[1],
gan_inference_cifar10.py (32x32x3, BATCH_SIZE=64) with the one MODE of that script that has a gradient penalty, MODE='wali-gp'
(CRITIC_ITERS=5, Adam 1e-4 / .5 / .9, LAMBDA=10: gan_inference_cifar10.py:57,351-366) -- so an iteration is 1 generator step + 5 critic
steps, each critic step with the penalty's double backward, and consumes 6 minibatches; `value` counts BATCH_SIZE images per iteration
as the reference's own iteration counter does.  Per-GPU batch stays 64 as N grows (weak scaling).

The one JSON line also carries `variants`: the other BASELINE configurations, each measured in a process of its own at N = 1
(`variant_leg`; GGAN_BENCH_VARIANTS_IN_PROCESS=1: behind the headline in this process, as rounds 2-5 did), each with its own ms_per_step, algorithmic GFLOP, roofline (dominant kernel) and cpu_baseline:
  ali                gan_inference_cifar10.py at the script's default MODE='ali' ("G+D", CRITIC_ITERS=1: the headline of rounds 1-4)
  gmgan-cifar10-K30  gmgan_inference_cifar10.py with the script's N_COMS=30;  gmgan-cifar10-K10: BASELINE configs[2] (K=10)
  gan-face           gan_inference_face.py 64x64x3 bs=64 (configs[3])
  ssgan-moving-mnist ssgan_inference_moving_mnist.py 64x64 T=16 bs=32 (configs[4])
  ssgan-moving-mnist-3dcnn  the same script with MODE='ali', ALI_MODE='3dcnn' (the Conv3D sequence critic, :352-405)

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense

VARIANTS = [
    dict(key='ali', dataset='cifar10', mode='ali'),
    dict(key='gmgan-cifar10-K30', dataset='cifar10', mode='local_ep', n_coms=30),
    dict(key='gmgan-cifar10-K10', dataset='cifar10', mode='local_ep', n_coms=10),
    dict(key='gan-face', dataset='face', mode='ali'),
    dict(key='ssgan-moving-mnist', dataset='moving_mnist', ssgan_mode='local_ep'),
    dict(key='ssgan-moving-mnist-3dcnn', dataset='moving_mnist', ssgan_mode='ali:3dcnn'),
]


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


def _build_id():
    from graphical_gan_amd import build
    return build.build_id()


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
    first = F * conv[0]                              # the critic's first layer: no data gradient in the critic step
    if getattr(cfg, 'seq_critic', False) and cfg.ali_mode == '3dcnn':
        # MODE ali / alice-z with the Conv3D sequence critic (:352-405): filter 4x4x4, spatial stride 2, the LEN plan of strides
        # along the sequence; one logit per sequence
        sls, vox, c3 = ([2, 2, 2, 2] if L == 16 else [2, 1, 2, 1]), L, []
        for i in range(4):
            vox = -(-vox // sls[i])
            c3.append(2.0 * vox * sizes[i] ** 2 * 64 * [1, d, 2 * d, 4 * d][i] * chans[i + 1])
        fDc, first = B * sum(c3), B * c3[0]
        fDl = lin(B, cfg.dim_g + cfg.dim_l * L + cfg.n_c, 512) + lin(B, cfg.flat + 512, 512) + lin(B, 512, 1)
        fDs = 0.0
        fwd = fE + fEG + fG + fdyn + 2 * (fDc + fDl)
        gen_bwd = (2 * fE - F * conv[0]) + (2 * fEG - B * convG1) + 2 * fG + 2 * fdyn + fDc + 2 * fDl
        disc_bwd = 2 * (2 * (fDc + fDl) - first)
        return ((fwd + gen_bwd) + cfg.critic_iters * (fwd + disc_bwd)) / 1e9
    fwd = fE + fEG + fG + fdyn + 2 * (fDc + fDl + fDs)
    gen_bwd = (2 * fE - F * conv[0]) + (2 * fEG - B * convG1) + 2 * fG + 2 * fdyn + fDc + 2 * fDl + 2 * fDs
    disc_bwd = 2 * (2 * (fDc + fDl + fDs) - first)
    return ((fwd + gen_bwd) + cfg.critic_iters * (fwd + disc_bwd)) / 1e9


_STATIC_BUILD = None


def _static_key(spec):
    """the profiles/pmc_traffic.json table of a workload: its VARIANTS key, also when it runs as the headline of a process of its own
    (variant_leg, or `python bench.py --mode ali` by hand)"""
    key = spec.get('key', 'headline')
    if key != 'headline':
        return key
    ds = spec.get('dataset')
    if ds == 'cifar10' and spec.get('mode') == 'wali-gp' and not spec.get('n_coms') and not spec.get('batch_size'):
        return 'headline'
    for v in VARIANTS:
        if v['dataset'] != ds or spec.get('batch_size'):
            continue
        if ds in ('moving_mnist', 'chairs'):
            if v.get('ssgan_mode') == spec.get('ssgan_mode'):
                return v['key']
        elif v.get('mode') == spec.get('mode') and (v.get('n_coms') or 0) == (spec.get('n_coms') or 0):
            return v['key']
    return '(no static table)'


def _pmc_table(workload_key):
    """Static figures that cannot be read from inside the process: per (kernel, grid) the duration inside the graph-replayed step
    (rocprofv3 --kernel-trace) and the counters of separate rocprofv3 --pmc passes of the same workload, written by tools/prof_round.sh /
    tools/collect_profiles.py into profiles/pmc_traffic.json.  -> (kernel -> {'by_grid': {grid: record}, ...}, tag)"""
    try:
        tab = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    except (OSError, ValueError):
        return {}, None
    tag = tab.get('_tag')
    global _STATIC_BUILD
    _STATIC_BUILD = tab.get('_build')
    if workload_key == 'gmgan-cifar10-K30' and workload_key not in tab:      # (profiled at K = 10: the same kernels and launch shapes but the mixture size)
        workload_key = 'gmgan-cifar10-K10'
    w = tab.get(workload_key)
    return (w, tag) if isinstance(w, dict) else ({}, tag)


def cpu_baseline(spec, cfg, K, np, torch, budget_s=float(os.environ.get('GGAN_BENCH_CPU_BUDGET_S', '3'))):
    """The same step on the host cores (oracle/, test infrastructure; the reference itself is Python 2 + TF1 and cannot run):
    PyTorch-CPU restatement (oneDNN convolutions, every granted core) for the image scripts, the numpy restatement on a
    bounded sample for the state-space script."""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=host_cores())
    except ImportError:
        pass
    if spec['dataset'] in ('moving_mnist', 'chairs') and cfg.mode in ('local_ep', 'ali') and cfg.pos_mode == 'naive_mean_field' \
            and (not cfg.seq_critic or cfg.ali_mode in ('concat_x', '3dcnn')):
        # the SAME iteration at the SAME size (all cfg.B sequences) on the PyTorch-CPU restatement of the state-space step (oneDNN convolutions,
        # every granted core) -- round-5 review: the numpy tape only managed a 2-sequence sample
        from oracle import ssgan as OSS, torch_cpu_ssgan
        torch.set_num_threads(host_cores())
        ocfg = OSS.Cfg(batch_size=cfg.B, length=cfg.LEN, n_c=cfg.n_c, channels=cfg.C, op_dyn_mode=cfg.op_dyn_mode, mode=cfg.mode,
                       ali_mode=cfg.ali_mode)
        ts = torch_cpu_ssgan.Step(ocfg, OSS.init_params(ocfg, 0), torch.float32)
        rng = np.random.default_rng(0)

        def feeds():
            while True:
                yield OSS.make_feed(ocfg, rng)
        fi = feeds()
        ts.iteration(1, fi)                            # warm-up (thread pool, oneDNN primitive cache)
        n_cpu, t1 = 0, time.perf_counter()
        while n_cpu < 1 or (time.perf_counter() - t1 < budget_s and n_cpu < 100):
            ts.iteration(2 + n_cpu, fi)
            n_cpu += 1
        cdt = time.perf_counter() - t1
        return dict(value=round(cfg.B * n_cpu / cdt, 3), unit='sequences/sec', cores=torch.get_num_threads(), kind='port',
                    sample='%d iteration(s) (gen step + critic step) of the same script at the same size (%d sequences x %d frames) in %.1f s, '
                           'PyTorch-CPU fp32 restatement (oneDNN convolutions, TF-SAME padding and TF-Adam emulated)' % (n_cpu, cfg.B, cfg.LEN, cdt))
    if spec['dataset'] in ('moving_mnist', 'chairs'):
        from oracle import ssgan as OSS
        ob = 2                                         # bounded sample: 2 sequences per minibatch instead of cfg.B
        ocfg = OSS.Cfg(batch_size=ob, length=cfg.LEN, n_c=cfg.n_c, channels=cfg.C, op_dyn_mode=cfg.op_dyn_mode, mode=cfg.mode,
                       ali_mode=cfg.ali_mode)
        otr = OSS.Trainer(ocfg, OSS.init_params(ocfg, 0), np.float32)
        feeds = iter([OSS.make_feed(ocfg, np.random.default_rng(i)) for i in range(8)])
        otr.iteration(1, feeds)
        t1 = time.perf_counter()
        otr.iteration(2, feeds)
        cdt = time.perf_counter() - t1
        # (the work of an iteration is linear in the number of sequences -- every layer is per frame or per sequence -- except the
        #  two Adam updates; the sample runs `ob` of the GPU workload's cfg.B sequences per minibatch)
        return dict(value=round(ob / cdt, 3), unit='sequences/sec', cores=host_cores(), kind='port',
                    sample='1 iteration (gen step + critic step) of the same script at %d of the %d sequences per minibatch x %d frames '
                           '(1/%d of the GPU iteration: per-sequence work is batch-independent), numpy fp32 oracle, multi-threaded BLAS'
                           % (ob, cfg.B, cfg.LEN, cfg.B // ob))
    mode = spec['mode']
    if mode not in ('ali', 'wali-gp', 'local_ep'):
        from oracle import nets as ON, step as OS
        ocfg = ON.Cfg(spec['dataset'], batch_size=cfg.B, n_coms=K)
        otr = OS.Trainer(ocfg, ON.init_params(ocfg, 0), mode, np.float32)
        feeds = iter([OS.make_feed(ocfg, np.random.default_rng(i), otr.mode) for i in range(64)])
        otr.iteration(1, feeds)
        t1 = time.perf_counter()
        otr.iteration(2, feeds)
        cdt = time.perf_counter() - t1
        return dict(value=round(cfg.B / cdt, 2), unit='images/sec', cores=host_cores(), kind='port',
                    sample='1 iteration (gen step + %d critic step(s)) of the same workload, numpy fp32 oracle with '
                    'multi-threaded BLAS' % cfg.critic_iters)
    from oracle import nets as ON, step as OS, torch_cpu
    torch.set_num_threads(host_cores())
    ocfg = ON.Cfg(spec['dataset'], batch_size=cfg.B, n_coms=K)
    ts = torch_cpu.Step(ocfg, ON.init_params(ocfg, 0), torch.float32, mode)
    rng = np.random.default_rng(0)
    omode = 'wali-gp' if mode == 'wali-gp' else 'ali'

    def feeds():
        while True:
            yield OS.make_feed(ocfg, rng, omode)
    fi = feeds()
    ts.iteration(1, fi)                                # warm-up (thread pool, oneDNN primitive cache)
    n_cpu, t1 = 0, time.perf_counter()
    while n_cpu < 1 or (time.perf_counter() - t1 < budget_s and n_cpu < 2000):
        ts.iteration(2 + n_cpu, fi)
        n_cpu += 1
    cdt = time.perf_counter() - t1
    return dict(value=round(cfg.B * n_cpu / cdt, 1), unit='images/sec', cores=torch.get_num_threads(), kind='port',
                sample='%d iteration(s) (gen step + %d critic step(s)) of the same workload in %.1f s, PyTorch-CPU fp32 restatement '
                '(oneDNN convolutions, TF-SAME padding and TF-Adam emulated)' % (n_cpu, ts.critic_iters, cdt))


def run_workload(spec, args, env, steps, warmup, top_kernels=None):
    """Build the workload, time `steps` iterations of graph replay, bracket its kernels (eager), time the CPU restatement.
    -> result dict (rank 0) / None.  Leaves the parameter registry and the optimizers empty."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from graphical_gan_amd import _lib, optim
    from graphical_gan_amd import tflib as lib
    from graphical_gan_amd.engine import Trainer, broadcast_params
    from graphical_gan_amd.models import Config
    dev, world, rank = env['dev'], env['world'], env['rank']
    optim.reset_optimizers()
    lib.delete_all_params()
    dataset = spec['dataset']
    ssgan = dataset in ('moving_mnist', 'chairs')
    batch = spec.get('batch_size') or ((32 if dataset == 'moving_mnist' else 16) if ssgan else 64)
    np.random.seed(0)                                  # reference initialisers draw from numpy's global RNG
    if ssgan:                                          # BASELINE configs[4]: ssgan_inference_moving_mnist.py, T=16
        from graphical_gan_amd.models_ssgan import SSConfig, StateSpaceGAN
        K = 0
        mode, _, ali_mode = spec.get('ssgan_mode', 'local_ep').partition(':')
        skw = dict(mode=mode, ali_mode=ali_mode or 'concat_x')
        if dataset == 'chairs':                        # ssgan_inference_chairs.py: 31 RGB views, no labels, res_w operator
            cfg = SSConfig(batch_size=batch, fuse=not args.no_fuse, length=31, n_c=0, channels=3, op_dyn_mode='res_w',
                           dataset='chairs', **skw)
        else:
            cfg = SSConfig(batch_size=batch, fuse=not args.no_fuse, **skw)
        model = StateSpaceGAN(cfg)
        tr = Trainer(cfg, device=dev, graph=not args.no_graph, seed=1234 + rank, model=model)
    else:
        mode = spec['mode']
        # N_COMS of the gmgan scripts: 30 (cifar10 :78, mnist), 50 (svhn :72), 100 (face :68)
        K = ({'svhn': 50, 'face': 100}.get(dataset, 30)) if mode in ('local_ep', 'local_epce') else 0
        if K and spec.get('n_coms'):
            K = spec['n_coms']
        # the code-space objectives run with the scripts' own settings for them: DIM_LATENT = 8, BN_FLAG = False (gan_inference_cifar10.py:72-77)
        code = mode in ('vegan', 'vegan-wgan-gp', 'vegan-kl', 'vegan-ikl', 'vegan-jsd')
        cfg = Config(dataset, batch_size=batch, n_coms=K, mode=mode, fuse=not args.no_fuse,
                     **(dict(dim_latent=8, bn=False) if code else {}))
        tr = Trainer(cfg, device=dev, graph=not args.no_graph, seed=1234 + rank)
    spec = dict(spec, mode=mode)
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
    feed_mode = 'host' if args.host_feed else 'staging buffer (one device copy per step)'
    if not args.host_feed and not args.no_ring and not ssgan:
        # the synthetic minibatches already sit in HBM: the steps read them in place (ring index = the optimizers' device-side step
        # counts) instead of through a staging buffer, and one iteration is one graph replay (Trainer.use_ring)
        try:
            tr.use_ring(ring)
            feed_mode = 'device-resident ring, read in place' + ('; one graph per iteration' if ((world == 1 or getattr(tr, 'dp_graph', False)) and not args.no_graph) else '')
        except (ValueError, RuntimeError):
            pass
    if args.host_feed and not args.no_ring and not ssgan and cfg.dataset != 'mnist':
        # host minibatches go straight into the ring's slots on a copy stream, one iteration ahead (Trainer.use_host_ring)
        try:
            tr.use_host_ring(lambda: iter(host_ring))
            feed_mode = 'host -> device ring on a copy stream, one iteration ahead; read in place'
        except (ValueError, RuntimeError, KeyError):
            pass
    for _ in range(max(warmup, 2)):                    # includes graph capture
        tr.iteration(it, bi); it += 1

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    tr.flush()
    fence()
    import contextlib
    # (--no-graph, the mode the rocprofv3 --pmc passes run in: eager steps on the launch plan of the graphs, so that the counters see the
    #  timed configuration's kernel mix -- the forked nets passes and their 128-workgroup plans)
    with (tr.eager_as_captured() if args.no_graph else contextlib.nullcontext()):
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last = tr.iteration(it, bi); it += 1
        tr.flush()                                         # (DP: the last critic step's exchange + Adam belong to the timed work)
        fence()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = all(np.isfinite(float(v)) for v in last.values())
    # run-to-run spread of the same K-step window (not part of `value`: the contract's timed region is the one above)
    repeats = []
    for _ in range(args.repeats if spec.get('key') == 'headline' else 0):
        fence()
        t1 = time.perf_counter()
        for _ in range(steps):
            tr.iteration(it, bi); it += 1
        tr.flush()
        fence()
        repeats.append(1e3 * (time.perf_counter() - t1) / steps)

    ms_per_step = 1e3 * dt / steps
    units_per_s = cfg.B * world * steps / dt
    gflop_it = ssgan_gflop_per_iteration(cfg) if ssgan else algorithmic_gflop_per_iteration(cfg)
    step_tflops = gflop_it * steps / dt / 1e3          # per GPU

    # ---- per-kernel timing of the same step with HIP events (eager, on the launch stream) ---------------------
    roofline = None
    kernels = None
    if not args.no_kernel_profile:
        # EVERY rank runs these eager iterations (they contain the gradient all-reduce: a collective issued by rank 0 alone
        # would hang the job); only rank 0 records and reports
        tr.flush()
        L = _lib.load()
        # (eager replay on the launch plan of the timed graph: the two-stream nets pass and its 128-workgroup plans are otherwise switched
        #  on for captures only, and the table would describe another kernel mix than `value` was timed on -- round-3 review)
        with tr.eager_as_captured():
            for _ in range(2):
                tr.iteration(it, bi); it += 1
            torch.cuda.synchronize(dev)
            L.ggan_prof_reset(); L.ggan_prof_enable(1 if rank == 0 else 0)
            n_prof = 5 if not ssgan else 3
            for _ in range(n_prof):
                tr.iteration(it, bi); it += 1
            torch.cuda.synchronize(dev)
            L.ggan_prof_enable(0)
        recs = _lib.prof_report() if rank == 0 else []
        L.ggan_prof_reset()
    if rank == 0 and not args.no_kernel_profile:
        # libggan's profiler keeps one record per (kernel, grid): a kernel a step launches on two problem sizes is two rows, and the
        # kernel-level figures below are SUMS over its rows -- flops, bytes and time of the same launches, so every ratio of them
        # describes the same launch mix (the mix of the timed graph: eager replay on its launch plan)
        byname = {}
        for r in recs:
            a = byname.setdefault(r['name'], dict(name=r['name'], total_ms=0.0, launches=0, flops=0.0, bytes=0.0, rows=[]))
            for k in ('total_ms', 'launches', 'flops', 'bytes'):
                a[k] += r[k]
            a['rows'].append(r)
        agg = sorted(byname.values(), key=lambda r: -r['total_ms'])
        tab, tag = _pmc_table(_static_key(spec))

        def rows_of(a):
            """rows of one kernel, one per GRID (what rocprofv3 can tell apart): live eager-bracket figures next to the static in-graph / counter
            figures of the same grid.  The profiler's records are per problem SHAPE (grid, flop per launch); where several shapes of a kernel
            share a grid (two layers whose tile counts multiply to the same number; 1-D split-K launches of a fixed workgroup count) the row
            is their launch-weighted mix and lists them under `shapes` -- its flop_per_launch, launches and durations are still sums over the
            same launches, so every kernel-level ratio stays exact; only a per-shape in-graph duration cannot be read from a trace."""
            st = (tab.get(a['name']) or {}).get('by_grid') or {}
            by_grid = {}
            for r in a['rows']:
                by_grid.setdefault(r['grid'], []).append(r)
            out = []
            for grid in sorted(by_grid):
                rs = sorted(by_grid[grid], key=lambda r: r['flops'] / r['launches'])
                n = sum(r['launches'] for r in rs)
                g = st.get(str(grid)) or {}
                row = dict(grid=grid, launches_per_step=n / n_prof, flop_per_launch=sum(r['flops'] for r in rs) / n,
                           algorithmic_bytes=round(sum(r['bytes'] for r in rs) / n), avg_us=round(1e3 * sum(r['total_ms'] for r in rs) / n, 2),
                           avg_us_in_graph=g.get('avg_us_in_graph'), traffic=g.get('traffic_bytes'), mfma_util_pct=g.get('mfma_util_pct'))
                if len(rs) > 1:
                    row['shapes'] = [dict(flop_per_launch=r['flops'] / r['launches'], launches_per_step=r['launches'] / n_prof,
                                          avg_us=round(1e3 * r['total_ms'] / r['launches'], 2)) for r in rs]
                out.append(row)
            return out

        def mixed(rows, field, weight='launches_per_step'):
            """a static per-grid figure weighted by the LIVE launch mix (None unless every launched shape has the figure)"""
            if not rows or any(r[field] is None for r in rows):
                return None
            return sum(r[weight] * r[field] for r in rows) / sum(r[weight] for r in rows)
        kernels = [dict(name=a['name'], launches_per_iter=a['launches'] / n_prof, ms_per_iter=round(a['total_ms'] / n_prof, 4),
                        avg_us=round(1e3 * a['total_ms'] / a['launches'], 2),
                        tflops=round(a['flops'] / (a['total_ms'] * 1e-3) / 1e12, 2) if a['flops'] else None,
                        grids=[dict(grid=r['grid'], flop_per_launch=r['flops'] / r['launches'], launches_per_iter=r['launches'] / n_prof,
                                    avg_us=round(1e3 * r['total_ms'] / r['launches'], 2))
                               for r in sorted(a['rows'], key=lambda r: (r['grid'], r['flops'] / r['launches']))]) for a in agg]
        launches_per_iter = sum(a['launches'] for a in agg) / n_prof
        # The dominant kernel = the one with the most time INSIDE THE TIMED GRAPH: per-shape duration of the rocprofv3 kernel trace of the
        # graph replay (profiles/pmc_traffic.json, the rows of profiles/<tag>_kernel_trace.md) x the launches counted here.  The live
        # eager bracket ranks kernels differently from run to run (its events sit around launches on two streams, so a launch's bracket
        # includes what it waited for on the other stream); it is kept as `frac_eager` / `avg_launch_us_eager`.
        for a in agg:
            rws = rows_of(a)
            a['_rows'] = rws
            a['_ig_ms'] = (sum(r['launches_per_step'] * r['avg_us_in_graph'] for r in rws) * 1e-3
                           if rws and all(r['avg_us_in_graph'] is not None for r in rws) else None)
        flop_k = [a for a in agg if a['flops'] > 0]
        static_ok = bool(flop_k) and all(a['_ig_ms'] is not None for a in flop_k[:6])       # (the eager top six all have in-graph figures)
        dom = (max((a for a in flop_k if a['_ig_ms'] is not None), key=lambda a: a['_ig_ms']) if static_ok
               else next(iter(flop_k), None))
        if dom is not None:
            ach_e = dom['flops'] / (dom['total_ms'] * 1e-3) / 1e12
            rows = dom['_rows']
            avg_us = 1e3 * dom['total_ms'] / dom['launches']
            fpl = dom['flops'] / dom['launches']
            ig_us = mixed(rows, 'avg_us_in_graph')
            ach = fpl / (ig_us * 1e-6) / 1e12 if ig_us else ach_e
            traffic = mixed(rows, 'traffic')
            # counter MFMA utilisation of the kernel = its rows weighted by the TIME each contributes
            for r in rows:
                r['_t'] = r['launches_per_step'] * (r['avg_us_in_graph'] or r['avg_us'])
            util = mixed(rows, 'mfma_util_pct', '_t')
            for r in rows:
                del r['_t']
            roofline = dict(bound='mfma', kernel=dom['name'], achieved=round(ach, 2), peak=MFMA_F32_PEAK_TFLOPS,
                            unit='TFLOP/s', frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                            frac_basis='in_graph' if ig_us else 'eager_bracket',
                            frac_eager=round(ach_e / MFMA_F32_PEAK_TFLOPS, 4),
                            frac_in_graph=round(fpl / (ig_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4) if ig_us else None,
                            avg_launch_us=round(ig_us, 2) if ig_us else round(avg_us, 2),
                            avg_launch_us_eager=round(avg_us, 2), avg_launch_us_in_graph=round(ig_us, 2) if ig_us else None,
                            time_in_graph_ms_per_step=round(dom['_ig_ms'], 4) if dom['_ig_ms'] is not None else None,
                            flop_per_launch=fpl,
                            traffic=round(traffic) if traffic else None,
                            algorithmic_bytes=round(dom['bytes'] / dom['launches']) if dom['bytes'] else None,
                            mfma_util_pct=round(util, 2) if util is not None else None,
                            launch_mix=rows,
                            static_tag=tag if (tab.get(dom['name']) is not None) else None,
                            static_matches_build=(_STATIC_BUILD == _build_id()) if _STATIC_BUILD else None,
                            source='kernel: the one with the most time inside the timed graph (per-shape duration of profiles/%s_kernel_trace.md x the launches '
                                   'counted here); achieved / frac / avg_launch_us: flop_per_launch / that trace\'s average duration of the kernel, weighted by the '
                                   'launch mix measured HERE (rocprofv3 --kernel-trace of the graph-replayed step; a trace of the same command is '
                                   'committed per round); frac_eager / avg_launch_us_eager: HIP-event bracket around every launch of this kernel on the stream '
                                   'it is launched on, eager replay of the same step ON THE LAUNCH PLAN OF THE TIMED GRAPH in this process (%d iterations: the '
                                   'bracket of a launch includes what it waited for on the other stream); flop_per_launch and algorithmic_bytes: sums over the '
                                   'same launches / their number; launch_mix: one row per problem shape (grid = work-items, flop per launch); traffic, '
                                   'mfma_util_pct: separate rocprofv3 --pmc passes (FETCH_SIZE x 2, WRITE_SIZE as counted: tools/pmc_summary.py), '
                                   'profiles/pmc_traffic.json @%s' % (tag, n_prof, tag),
                            whole_step_tflops=round(step_tflops, 2),
                            whole_step_frac=round(step_tflops / MFMA_F32_PEAK_TFLOPS, 4),
                            libggan_launches_per_step=launches_per_iter)
            # the conv stack as a whole (north star: "MFMA utilisation on the Conv2D / Deconv2D stack"): counter MFMA utilisation of every
            # MFMA conv kernel (static per-grid table, median per dispatch) weighted by the kernel time measured here
            cw = ca = 0.0
            for a in agg:
                if not a['name'].startswith(('corr_kernel', 'wgrad_kernel', 'wgrad4_kernel', 'dg16_kernel', 'conv3d_igemm')):
                    continue
                for row in a['_rows']:
                    if row['mfma_util_pct'] is not None:
                        t_ms = sum(r['total_ms'] for r in a['rows'] if r['grid'] == row['grid'])
                        cw += t_ms
                        ca += t_ms * row['mfma_util_pct']
            # the chip's matrix pipes over the WHOLE step: MFMA-busy SIMD-cycles of every launch of an iteration (static per-(kernel, grid)
            # mean of SQ_VALU_MFMA_BUSY_CYCLES x the launches counted here) / (1024 SIMDs x 2.4 GHz x the measured iteration time).
            # Unlike a per-kernel figure it does not care how many kernels share the chip at a time
            busy = fl_cov = fl_all = 0.0
            for a in agg:
                st = (tab.get(a['name']) or {}).get('by_grid') or {}
                for r in a['rows']:
                    g = st.get(str(r['grid'])) or {}       # (busy cycles of a grid's launches: a mean over the shapes that share it, times their launches)
                    fl_all += r['flops']
                    if g.get('mfma_busy_cycles') is not None:
                        busy += g['mfma_busy_cycles'] * r['launches'] / n_prof
                        fl_cov += r['flops']
            if busy > 0 and fl_cov >= 0.9 * fl_all:       # (rows without a counter sample: the Linear products, whose symbols differ between the two profilers)
                roofline['whole_step_mfma_util_pct'] = round(100.0 * busy / (ms_per_step * 1e-3 * 2.4e9 * 1024), 1)
                roofline['whole_step_mfma_util_flop_coverage'] = round(fl_cov / fl_all, 3)
            if cw > 0:
                roofline['conv_stack_mfma_util_pct'] = round(ca / cw, 1)
                roofline['conv_stack_share_of_kernel_time'] = round(cw / sum(a['total_ms'] for a in agg), 3)
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(spec, cfg, K, np, torch)

    res = None
    if rank == 0:
        unit = 'sequences/sec' if ssgan else 'images/sec'
        if ssgan:
            metric = 'sequences/sec (G+D step) %s T=%d 64x64 bs=%d' % (dataset, cfg.LEN, cfg.B)
            workload = ('ssgan_inference_%s.py MODE=%s POS_MODE=%s LEN=%d BATCH_SIZE=%d sequences (per GPU) of 64x64x%d frames%s' % (
                dataset, cfg.mode + ((' ALI_MODE=' + cfg.ali_mode) if cfg.seq_critic else ''), cfg.pos_mode, cfg.LEN, cfg.B, cfg.C,
                ', eager' if args.no_graph else ''))
        else:
            metric = 'images/sec (G+D%s step) %s %dx%d bs=%d' % ('+GP' if mode == 'wali-gp' else '', dataset, cfg.S, cfg.S, cfg.B)
            workload = '%s_inference_%s.py MODE=%s%s BATCH_SIZE=%d (per GPU) %dx%dx%d, CRITIC_ITERS=%d%s' % (
                'gmgan' if K else 'gan', dataset, mode, ' N_COMS=%d' % K if K else '', cfg.B, cfg.S, cfg.S, cfg.C,
                cfg.critic_iters, '' if not args.no_graph else ', eager')
        res = {
            'metric': metric, 'value': round(units_per_s, 1), 'unit': unit, 'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload, 'parallelism': 'dp%d' % world, 'global_batch': cfg.B * world,
                       'hip_graph': not args.no_graph, 'host_feed': bool(args.host_feed), 'minibatch_feed': feed_mode, 'fused_epilogues': not args.no_fuse,
                       'minibatches_per_step': 1 + cfg.critic_iters, 'algorithmic_gflop_per_step': round(gflop_it, 2),
                       'finite_costs': bool(finite), **({'frames_per_sec': round(units_per_s * cfg.LEN, 1)} if ssgan else {})},
            'algorithmic_gflop_per_step': round(gflop_it, 2),
            'whole_step_tflops': round(step_tflops, 2), 'whole_step_frac': round(step_tflops / MFMA_F32_PEAK_TFLOPS, 4),
            'roofline': roofline, 'cpu_baseline': cpu,
            **({'repeat_ms_per_step': [round(x, 4) for x in repeats]} if repeats else {}),
            'kernels': kernels[:top_kernels] if (kernels and top_kernels) else kernels,
        }
    # ---- tear down: graphs, flat optimizer buffers, parameters ------------------------------------------------
    tr.flush()
    torch.cuda.synchronize(dev)
    del tr, ring, bi
    optim.reset_optimizers()
    lib.delete_all_params()
    from graphical_gan_amd import functional as _F
    _F._WS.clear()                     # (per-stream scratch of the streams that just died)
    gc.collect()
    torch.cuda.empty_cache()
    return res


def supervise():
    """N > 1: every rank process that torch.distributed.run starts is only a supervisor; the measurement runs in a child process
    per rank (same arguments, process group of its own on MASTER_PORT + 1 + attempt).  Attempt 0 captures the gradient exchange
    inside the step graphs.  Where captured collectives do not work on a stack the failure is not an exception but an ABORT of the
    rank process, and the other ranks would then sit in a collective until its timeout -- so the
    supervisors watch a key in the launcher's store: the first child that dies sets it, everybody stops their child, and
    attempt 1 runs the same benchmark with the exchange issued by the host between cut graphs (GGAN_DP_GRAPH=0).  The JSON line
    of the attempt that finished is passed through by rank 0."""
    import datetime
    import subprocess
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    base_port = int(os.environ.get('MASTER_PORT', '29500'))
    try:
        if os.environ.get('TORCHELASTIC_USE_AGENT_STORE', 'True') not in ('True', 'true', '1'):
            raise RuntimeError('the launcher hosts no store on MASTER_PORT')
        store = dist.TCPStore(os.environ.get('MASTER_ADDR', '127.0.0.1'), base_port, is_master=False, timeout=datetime.timedelta(seconds=60))
    except Exception as e:                                # noqa: BLE001
        if rank == 0:
            sys.stderr.write('[bench] no launcher store to supervise through (%s): measuring in this process, exchanges between cut graphs\n' % type(e).__name__)
        return None
    modes = ([os.environ['GGAN_DP_GRAPH']] if 'GGAN_DP_GRAPH' in os.environ else ['1']) + ['0']
    for attempt, dp_graph in enumerate(modes[:2] if modes[0] != '0' else modes[:1]):
        env = dict(os.environ, GGAN_BENCH_CHILD='1', GGAN_DP_GRAPH=dp_graph, MASTER_PORT=str(base_port + 1 + attempt))
        if attempt > 0:
            # the retry takes the most conservative exchange there is: torch.distributed's own process group issuing the all-reduce from
            # the host between cut graphs -- not the directly bound communicator, whose first attempt just failed or hung
            env['GGAN_NO_DIRECT_RCCL'] = '1'
        env.pop('TORCHELASTIC_USE_AGENT_STORE', None)      # (the children's rank 0 hosts the store of their process group)
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE, text=True)
        fail_key = 'ggan_bench/attempt%d/failed' % attempt
        t_start, limit = time.time(), float(os.environ.get('GGAN_BENCH_ATTEMPT_TIMEOUT_S', '420'))
        while p.poll() is None:
            hung = attempt == 0 and len(modes) > 1 and time.time() - t_start > limit      # (a capture that deadlocks instead of aborting)
            if hung or store.check([fail_key]):
                p.kill()
                break
            time.sleep(0.5)
        out = p.communicate()[0]
        rc = p.returncode
        if rc != 0:
            store.set(fail_key, b'1')
        store.set('ggan_bench/attempt%d/rank%d' % (attempt, rank), str(rc).encode())
        store.wait(['ggan_bench/attempt%d/rank%d' % (attempt, r) for r in range(world)], datetime.timedelta(seconds=3600))
        if not store.check([fail_key]):
            if rank == 0:
                lines = out.splitlines()
                js = [l for l in lines if l.startswith('{"metric"')]
                sys.stdout.write(''.join(l + '\n' for l in lines if not l.startswith('{"metric"')))
                if js:
                    sys.stdout.write(js[-1] + '\n')      # (whatever else a library wrote to the child's stdout: the contract line stays last)
                sys.stdout.flush()
            return 0
        if rank == 0:
            sys.stderr.write('[bench] attempt %d (GGAN_DP_GRAPH=%s) did not finish on every rank%s\n' % (
                attempt, dp_graph, ': retrying with host-issued exchanges between cut graphs' if attempt == 0 and len(modes) > 1 and modes[0] != '0' else ''))
    return 1


# environment switches that do NOT change what a step launches (bookkeeping of the benchmark itself, rendezvous, tracing to stderr);
# every other GGAN_* variable that is set is a diagnostic switch of libggan / the engine and is stamped into the line
BENIGN_ENV = ('GGAN_DIST_BACKEND', 'GGAN_DP_GRAPH', 'GGAN_NO_DIRECT_RCCL', 'GGAN_FORCE_ALLREDUCE', 'GGAN_TRACE_LAUNCHES', 'GGAN_TRACE_NAIVE',
              'GGAN_TRACE_CONV', 'GGAN_BUILD_DIAG', 'GGAN_CAPTURE_MODE', 'GGAN_TEST_REPORT')


def diag_env():
    """diagnostic switches of libggan / the engine that change what is launched: a line measured with any of them set says so"""
    return sorted(k for k, v in os.environ.items()
                  if k.startswith('GGAN_') and not k.startswith('GGAN_BENCH_') and k not in BENIGN_ENV and v not in (None, ''))


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (the form the driver uses for its 1-GPU run): start the N ranks
    here -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N on 127.0.0.1 and a free port, this script, the same
    arguments -- pass their output through and keep the contract line the last line of stdout."""
    import socket
    import subprocess
    def free(p):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            try:
                sk.bind(('127.0.0.1', p))
            except OSError:
                return None
            return sk.getsockname()[1]
    port = free(0)
    if not (20000 <= port <= 60000 and free(port + 1) and free(port + 2)):      # (bench children use MASTER_PORT + 1 + attempt)
        port = next((p for p in range(29600, 60000, 7) if free(p) and free(p + 1) and free(p + 2)), port)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), GGAN_BENCH_SELF_LAUNCHED='1')
    sys.stderr.write('[bench] --gpus %d without a launcher: starting the ranks with %s\n' % (args.gpus, ' '.join(cmd[1:9])))
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    out = p.communicate()[0]
    lines = out.splitlines()
    js = [l for l in lines if l.startswith('{"metric"')]
    sys.stdout.write(''.join(l + '\n' for l in lines if not l.startswith('{"metric"')))
    if js:
        sys.stdout.write(js[-1] + '\n')
    sys.stdout.flush()
    return p.returncode if (p.returncode != 0 or js) else 1


LINE_LIMIT = 4096      # the driver keeps the tail of stdout: the LAST line must be one parseable JSON object well under that


def _short_cpu(c):
    return None if not c else dict(value=c['value'], unit=c['unit'], cores=c['cores'], kind=c['kind'], sample=c['sample'][:120])


def _short_roofline(r):
    if not r:
        return None
    keep = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_basis', 'frac_eager', 'frac_in_graph', 'avg_launch_us', 'avg_launch_us_eager',
            'flop_per_launch', 'traffic', 'algorithmic_bytes', 'mfma_util_pct', 'conv_stack_mfma_util_pct', 'whole_step_mfma_util_pct', 'static_tag',
            'static_matches_build')
    d = {k: r.get(k) for k in keep}
    d['source'] = ('kernel = most time inside the timed graph; frac = flop_per_launch / avg_launch_us / peak, avg_launch_us = its average duration in profiles/'
                   '%s_kernel_trace.md (rocprofv3 trace of the graph replay) on the launch mix counted live; frac_eager: live HIP-event bracket; traffic / '
                   'mfma_util: --pmc passes, profiles/pmc_traffic.json' % r.get('static_tag'))
    return d


def compact_line(full):
    """The contract line: headline keys + roofline + cpu_baseline + one short record per variant; everything else
    (kernel tables, prose, per-variant configs) lives in the full record (bench_full.json)."""
    line = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                     'vs_baseline', 'dtype', 'data')}
    c = full.get('config') or {}
    line['config'] = {k: c[k] for k in ('workload', 'parallelism', 'global_batch', 'hip_graph', 'minibatch_feed', 'minibatches_per_step',
                                         'algorithmic_gflop_per_step', 'finite_costs') if k in c}
    line['whole_step_frac'] = full.get('whole_step_frac')
    if full.get('repeat_ms_per_step'):      # the K-step window repeated after the timed one: the spread a reader of `value` should see
        line['repeat_ms_per_step'] = full['repeat_ms_per_step']
    if full.get('roofline'):
        line['launches_per_step'] = full['roofline'].get('libggan_launches_per_step')
    line['roofline'] = _short_roofline(full.get('roofline'))
    line['cpu_baseline'] = _short_cpu(full.get('cpu_baseline'))
    if full.get('data_parallel'):
        line['data_parallel'] = full['data_parallel']
    if full.get('dp_schedule_n1'):
        line['dp_schedule_n1'] = {k: full['dp_schedule_n1'].get(k) for k in ('ms_per_step', 'value', 'vs_single_replica_schedule_pct', 'error')
                                  if full['dp_schedule_n1'].get(k) is not None}
    vs = []
    for v in full.get('variants') or []:
        r = v.get('roofline') or {}
        vs.append(dict(key=v['key'], value=v['value'], unit=v['unit'], ms_per_step=v['ms_per_step'],
                       whole_step_frac=v.get('whole_step_frac'), kernel=r.get('kernel'), frac=r.get('frac'),
                       frac_in_graph=r.get('frac_in_graph'), conv_mfma_util=r.get('conv_stack_mfma_util_pct'), cpu=(v.get('cpu_baseline') or {}).get('value'),
                       finite=(v.get('config') or {}).get('finite_costs')))
    line['variants'] = vs
    gd = next((v for v in vs if v['key'] == 'ali'), None)
    if gd is not None:          # the same script at its default MODE='ali' (CRITIC_ITERS 1, no penalty): the "G+D step", headline of rounds 1-4
        line['g_d_step'] = dict(value=gd['value'], unit=gd['unit'], ms_per_step=gd['ms_per_step'], whole_step_frac=gd['whole_step_frac'])
    line['full_record'] = full.get('_full_path')
    if full.get('diag_env'):
        line['diag_env'] = full['diag_env']
    txt = json.dumps(line, separators=(',', ':'))
    # never let the line outgrow the driver's window: drop detail step by step, keep the contract keys
    if len(txt) >= LINE_LIMIT:
        for v in vs:
            v.pop('kernel', None)
        if line.get('roofline'):
            line['roofline'].pop('source', None)
        txt = json.dumps(line, separators=(',', ':'))
    for drop in ('repeat_ms_per_step', 'dp_schedule_n1', 'variants', 'data_parallel', 'g_d_step', 'cpu_baseline', 'roofline', 'config'):
        if len(txt) < LINE_LIMIT:
            break
        if drop == 'cpu_baseline' and line.get(drop):
            line[drop] = {k: line[drop].get(k) for k in ('value', 'unit', 'cores', 'kind')}
        elif drop == 'roofline' and line.get(drop):
            line[drop] = {k: line[drop].get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
        elif drop == 'config':
            line[drop] = {'workload': str((line.get(drop) or {}).get('workload'))[:200]}
        else:
            line.pop(drop, None)
        txt = json.dumps(line, separators=(',', ':'))
    return txt


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                      # noqa: BLE001
        pass


def emit(full):
    """Full record -> gpurun_out/bench_full.json (GGAN_BENCH_FULL overrides; merged back from the GPU box), compact line -> the
    last line of stdout."""
    path = os.environ.get('GGAN_BENCH_FULL') or os.path.join(ROOT, 'gpurun_out', 'bench_full.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(full, f, indent=1)
        full['_full_path'] = os.path.relpath(path, ROOT)
    except OSError:
        full['_full_path'] = None
    sys.stdout.flush()
    _flush_c_stdio()
    print(compact_line(full))
    sys.stdout.flush()


def dp_schedule_leg(args, head_ms):
    import subprocess
    import tempfile
    port = 29600 + (os.getpid() % 300)
    tmp = tempfile.NamedTemporaryFile(prefix='ggan_dp_n1_', suffix='.json', delete=False)
    tmp.close()
    env = dict(os.environ, GGAN_FORCE_ALLREDUCE='1', GGAN_BENCH_NO_DP_LEG='1', GGAN_BENCH_FULL=tmp.name, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env = {k: v for k, v in env.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port',
           str(port), os.path.abspath(__file__), '--gpus', '1', '--steps', str(min(args.steps, 100)), '--warmup', str(min(args.warmup, 5)),
           '--no-variants', '--no-cpu-baseline', '--no-kernel-profile', '--repeats', '0']
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get('GGAN_BENCH_DP_LEG_TIMEOUT_S', '240')), env=env, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode != 0 or not lines:
            return dict(error='child exited %d: %s' % (r.returncode, (r.stderr or '')[-300:]))
        d = json.loads(lines[-1])
        rec = dict(ms_per_step=d['ms_per_step'], value=d['value'], unit=d['unit'], steps=d['steps'], ranks=1,
                   exchange='one-rank RCCL all-reduce captured in the iteration graph (GGAN_FORCE_ALLREDUCE=1), child process',
                   minibatch_feed=(d.get('config') or {}).get('minibatch_feed'), finite=(d.get('config') or {}).get('finite_costs'))
        if head_ms:
            rec['vs_single_replica_schedule_pct'] = round(100.0 * (d['ms_per_step'] / head_ms - 1.0), 2)
        return rec
    except Exception as e:                                 # noqa: BLE001  (a leg that fails must not take the contract line with it)
        return dict(error='%s: %s' % (type(e).__name__, str(e)[:200]))
    finally:
        try:
            os.unlink(tmp.name)
        except OSError:
            pass


def variant_leg(v, args, vsteps, warmup):
    """one of the other BASELINE configurations in a process of its own (N = 1): `python bench.py --dataset .. --mode .. --no-variants`, the
    full record read back.  A workload run BEHIND another one in the same process maps its graph branches onto hardware queues the earlier
    Trainer's streams left behind (gmgan: 1.10-1.11 ms behind the headline, 1.06-1.07 ms alone -- profiles/r06_notes.md section 13; the others
    measure the same either way); a process of its own is also how a user runs the script.  None: the caller measures in this process."""
    import subprocess
    import tempfile
    if os.environ.get('GGAN_BENCH_VARIANTS_IN_PROCESS') or os.environ.get('GGAN_FORCE_ALLREDUCE'):      # (the one-rank RCCL rehearsal keeps its process group)
        return None
    tmp = tempfile.NamedTemporaryFile(prefix='ggan_variant_', suffix='.json', delete=False)
    tmp.close()
    env = dict(os.environ, GGAN_BENCH_NO_DP_LEG='1', GGAN_BENCH_FULL=tmp.name)
    env = {k: val for k, val in env.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', str(vsteps), '--warmup', str(warmup), '--no-variants',
           '--repeats', '0', '--dataset', v['dataset']]
    if v.get('mode'):
        cmd += ['--mode', v['mode']]
    if v.get('ssgan_mode'):
        cmd += ['--ssgan-mode', v['ssgan_mode']]
    if v.get('n_coms') is not None:
        cmd += ['--n-coms', str(v['n_coms'])]
    for flag, on in (('--no-graph', args.no_graph), ('--no-fuse', args.no_fuse), ('--no-cpu-baseline', args.no_cpu_baseline),
                     ('--no-kernel-profile', args.no_kernel_profile), ('--no-ring', args.no_ring)):
        if on:
            cmd.append(flag)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get('GGAN_BENCH_VARIANT_TIMEOUT_S', '300')), env=env, cwd=ROOT)
        if r.returncode != 0:
            sys.stderr.write('[bench] variant %s: child exited %d (%s); measuring it in this process\n' % (v['key'], r.returncode, (r.stderr or '')[-200:]))
            return None
        with open(tmp.name) as f:
            rec = json.load(f)
        for drop in ('variants', 'dp_schedule_n1', '_full_path', 'repeat_ms_per_step'):
            rec.pop(drop, None)
        if rec.get('kernels'):
            rec['kernels'] = rec['kernels'][:8]
        rec['process'] = 'own'
        return rec
    except Exception as e:                                 # noqa: BLE001
        sys.stderr.write('[bench] variant %s: %s: %s; measuring it in this process\n' % (v['key'], type(e).__name__, str(e)[:200]))
        return None
    finally:
        try:
            os.unlink(tmp.name)
        except OSError:
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--dataset', default='cifar10')
    ap.add_argument('--mode', default='wali-gp', help="wali-gp (the BASELINE metric's G+D+GP step) | ali | local_ep (gmgan, N_COMS=30) | ...")
    ap.add_argument('--ssgan-mode', default='local_ep', help="moving_mnist / chairs: local_ep | local_epce-z | ali | alice-z, optionally "
                    "':concat_x' | ':concat_z' | ':3dcnn' (the sequence critic of ali / alice-z)")
    ap.add_argument('--n-coms', type=int, default=None, help='mixture components of the gmgan prior (default: the script value, 30 / 50 / 100)')
    ap.add_argument('--batch-size', type=int, default=None, help='per-GPU minibatch (default 64; 32 sequences for moving_mnist)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-fuse', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-profile', action='store_true')
    ap.add_argument('--no-variants', action='store_true', help='headline workload only')
    ap.add_argument('--repeats', type=int, default=3, help='extra repetitions of the K-step window after the timed one (spread; headline only)')
    ap.add_argument('--variants', default=None, help='comma-separated subset of: ' + ', '.join(v['key'] for v in VARIANTS))
    ap.add_argument('--variant-steps', type=int, default=None, help='timed iterations per variant (default: min(steps, 60))')
    ap.add_argument('--no-ring', action='store_true', help='minibatches through the staging buffer (one device copy per step) instead of read in place from the resident ring')
    ap.add_argument('--host-feed', action='store_true',
                    help='minibatches start in host memory (pinned double-buffered H->D copies): the PCIe-inclusive rate')
    args = ap.parse_args()
    diag0 = diag_env()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ and not os.environ.get('GGAN_BENCH_CHILD'):
        return self_launch(args)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and not os.environ.get('GGAN_BENCH_CHILD') and not os.environ.get('GGAN_BENCH_NO_SUPERVISOR'):
        rc = supervise()
        if rc is not None:
            return rc
        os.environ.setdefault('GGAN_DP_GRAPH', '0')       # (no launcher store to coordinate a retry through: take the path that cannot abort)
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
    if world > 1 or os.environ.get('GGAN_FORCE_ALLREDUCE'):     # (forced: a one-rank rehearsal of the RCCL-in-graph path)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if os.environ.get('GGAN_BENCH_ABORT_TEST') == '%s:%s' % (os.environ.get('GGAN_DP_GRAPH', '1'), rank):
        os.abort()                                         # (tests: a rank that dies the way a failed captured collective kills it)
    import __graft_entry__ as ge
    ge.build()
    env = dict(dev=dev, world=world, rank=rank)

    head_spec = dict(key='headline', dataset=args.dataset, mode=args.mode, ssgan_mode=args.ssgan_mode, n_coms=args.n_coms,
                     batch_size=args.batch_size)
    out = run_workload(head_spec, args, env, args.steps, args.warmup)

    # the other BASELINE configurations, same measurement, each in a process of its own at N = 1 (only next to the default headline workload)
    default_head = args.dataset == 'cifar10' and args.mode == 'wali-gp' and args.batch_size is None and not args.host_feed
    variants = []
    if not args.no_variants and (default_head or args.variants):
        want = [k.strip() for k in args.variants.split(',')] if args.variants else [v['key'] for v in VARIANTS]
        vsteps = args.variant_steps or min(args.steps, 60)
        for v in VARIANTS:
            if v['key'] not in want:
                continue
            t0 = time.perf_counter()
            r = variant_leg(v, args, vsteps, min(args.warmup, 5)) if world == 1 else None
            if r is None:
                r = run_workload(v, args, env, vsteps, min(args.warmup, 5), top_kernels=8)
                if r is not None:
                    r['process'] = 'shared with the headline'
            if r is not None:
                r['key'] = v['key']
                r['wall_s'] = round(time.perf_counter() - t0, 1)
                variants.append(r)
    # N > 1: the same headline step (a) without its gradient exchange -> what the exchange costs a step once overlap is
    # accounted for, (b) strong scaling: the reference's global batch of 64 split over the replicas
    dp_extra = None
    rccl_ranks = None
    if world > 1:
        try:
            from graphical_gan_amd import rccl as _rccl
            c = _rccl.get(create=False)
            rccl_ranks = c.count() if c is not None else None
        except Exception:                                  # noqa: BLE001
            rccl_ranks = None
    if world > 1 and default_head and not args.no_variants:
        os.environ['GGAN_SKIP_ALLREDUCE'] = '1'
        r0 = run_workload(head_spec, args, env, min(args.steps, 100), min(args.warmup, 5), top_kernels=1)
        del os.environ['GGAN_SKIP_ALLREDUCE']
        dp_extra = {}
        if rank == 0:
            dp_extra['ms_per_step_without_exchange'] = r0['ms_per_step']
            dp_extra['exposed_exchange_ms_per_step'] = round(out['ms_per_step'] - r0['ms_per_step'], 4)
        if 64 % world == 0 and 64 // world >= 4:
            sargs = argparse.Namespace(**dict(vars(args), no_kernel_profile=True, no_cpu_baseline=True))
            r1 = run_workload(dict(head_spec, batch_size=64 // world), sargs, env, min(args.steps, 100), min(args.warmup, 5))
            if rank == 0:
                dp_extra['strong_scaling'] = {'global_batch': 64, 'per_gpu_batch': 64 // world, 'images_per_sec': r1['value'],
                                              'ms_per_step': r1['ms_per_step']}
    # N = 1: the same headline iteration on the DATA-PARALLEL schedule -- one rank over RCCL, the gradient exchange captured inside the
    # iteration graph (pack -> all-reduce on the communicator's stream -> Adam, instead of the update riding in the pack launch), the
    # critic steps' nets passes ahead of time as in the single-replica graph -- measured in a child process of its own, so that nothing
    # of RCCL lives in the process whose line the driver parses.  What a replica of an N > 1 run executes, minus the wire.
    dp_n1 = None
    if world == 1 and default_head and not args.no_variants and not args.no_graph and not os.environ.get('GGAN_BENCH_NO_DP_LEG') \
            and not os.environ.get('GGAN_FORCE_ALLREDUCE'):
        dp_n1 = dp_schedule_leg(args, out['ms_per_step'] if out else None)
    # the contract line must be the LAST thing on stdout: librccl writes its version banner to stdout through C stdio (block-buffered on a
    # pipe, so it would surface at exit, behind the line) -- tear the process groups down first, flush C stdio, then print
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        try:
            from graphical_gan_amd import rccl
            rccl.reset()
        except Exception:                                  # noqa: BLE001
            pass
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        if diag0:
            out['diag_env'] = diag0
            sys.stderr.write('[bench] diagnostic switches set: %s -- this line does not describe the product configuration\n' % ', '.join(diag0))
        if world > 1:
            out['data_parallel'] = dict(dp_extra or {}, ranks=world, rccl_ranks=rccl_ranks, backend=os.environ.get('GGAN_DIST_BACKEND', 'nccl'),
                                        launched_by='bench.py itself' if os.environ.get('GGAN_BENCH_SELF_LAUNCHED') else 'external launcher',
                                        exchange='captured in the step graph' if os.environ.get('GGAN_DP_GRAPH', '1') != '0'
                                        else 'host-issued between cut graphs', weak_scaling_per_gpu_batch=64)
        out['variants'] = variants
        if dp_n1 is not None:
            out['dp_schedule_n1'] = dp_n1
        emit(out)


if __name__ == '__main__':
    sys.exit(main() or 0)
