#!/usr/bin/env python
"""Find the launch-SITE plan of a workload's step graph by measurement (functional.site_scope, engine.Trainer._site_plan).

The pass-level plans (functional.target_workgroups / launch_hint) ask every conv launch of a chain that runs beside another chain for
~128 workgroups.  In the captured graph some of those launches run alone; which ones is a property of the schedule, so it is measured:
  A. every site whose plan is not the default is flipped to the default (full chip) ALONE, the iteration graph is re-captured and timed;
  B. the flips that gained are applied cumulatively, best first, each kept only if the iteration got shorter again;
  C. the result is timed against the empty table in alternating rounds and written as a site_plans.json entry.
The critic steps of an iteration that are built alike (disc1 .. disc{n-2}: each with one ahead-of-time nets pass beside it) are tied
together in A (one trial flips the site in all of them) -- they are separate entries in the table that is written.

usage (GPU box): python tools/site_sweep.py [--dataset cifar10 --mode wali-gp --n-coms 0] [--iters 150] [--out gpurun_out/site_plan.json]
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='cifar10')
    ap.add_argument('--mode', default='wali-gp')
    ap.add_argument('--n-coms', type=int, default=0)
    ap.add_argument('--iters', type=int, default=150)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'site_plan.json'))
    ap.add_argument('--min-gain-us', type=float, default=2.0)
    ap.add_argument('--max-trials', type=int, default=400)
    ap.add_argument('--max-stage-b', type=int, default=30)
    ap.add_argument('--batch-size', type=int, default=64)
    ap.add_argument('--start', default='', help='a site_plans.json to start from (its entry for this workload)')
    ap.add_argument('--values', default='0', help='comma-separated plan values a non-default site is tried on (0 = full chip)')
    args = ap.parse_args()
    import numpy as np
    import torch
    from graphical_gan_amd import tflib as lib, optim
    from graphical_gan_amd.engine import Trainer
    from graphical_gan_amd.models import Config
    dev = torch.device('cuda:0')
    os.environ.pop('GGAN_SITE_PLAN', None)                # (tables come from site_plan_override here; '0' would switch them off)
    np.random.seed(0)
    K = args.n_coms
    cfg = Config(args.dataset, batch_size=args.batch_size, n_coms=K, mode=args.mode)
    tr = Trainer(cfg, device=dev, graph=True, seed=1234)
    torch.manual_seed(1234)
    ring = tr.model.synthetic_ring(dev, n=8, seed=1234)
    bi = iter(lambda: ring[0], None)
    it = 0
    for _ in range(2):
        tr.iteration(it, bi); it += 1
    tr.use_ring(ring)
    tr.record_site_log = True
    tr.site_plan_override = {}

    def timed(table, iters=args.iters):
        nonlocal it
        tr.site_plan_override = dict(table)
        tr._iter_graph = None
        tr._graphs = {}
        gc.collect()
        for _ in range(3):
            tr.iteration(it, bi); it += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            tr.iteration(it, bi); it += 1
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / iters

    base = [timed({}) for _ in range(3)]
    key = tr.last_site_plan_key
    log = list(tr.site_log)
    print('workload %s: %d sites, base %s ms' % (key, len(log), ['%.4f' % b for b in base]), flush=True)
    scopes = sorted(set(s.split(':')[0] for s, *_ in log))
    discs = sorted((s for s in scopes if s.startswith('disc')), key=lambda s: int(s[4:]))
    tied = {}                                             # scope -> group name
    for s in scopes:
        tied[s] = 'discM' if (len(discs) >= 4 and s in discs[1:-1]) else s
    # candidates: (group, ordinal, field) whose planned value is not the default
    by_site = {s: (g, w, wf) for s, g, w, wf in log}
    cands = {}
    for s, g, w, wf in log:
        sc, o = s.split(':')
        for field, v in (('wgs', w), ('wgs_filter', wf)):
            if v != 0:
                cands.setdefault((tied[sc], int(o), field), []).append(s)
    # (tied scopes must hold the same geometry at the same ordinal, else they are not built alike: untie)
    for (grp, o, field), sites in list(cands.items()):
        if len(set(by_site[s][0] for s in sites)) > 1:
            del cands[(grp, o, field)]
            for s in sites:
                cands[(s.split(':')[0], o, field)] = [s]
    print('%d candidate (site group, field) pairs' % len(cands), flush=True)

    start = {}
    if args.start:
        start = (json.load(open(args.start)).get(key) or {}).get('sites') or {}

    def with_flip(table, sites, field, value):
        t = {k: dict(v) for k, v in table.items()}
        for s in sites:
            e = t.setdefault(s, dict(geom=list(by_site[s][0][:5]), wgs=-1, wgs_filter=-1))
            e[field] = value
        return t

    values = [int(v) for v in args.values.split(',')]
    # A re-captured graph of the SAME table times +-0.3 % apart (where its buffers land, how its branches map onto the hardware queues):
    # single-flip deltas are taken against the median of the three most recent timings of the starting table, re-timed every 10 trials
    bases = list(base)
    ref = sorted(bases[-3:])[1]
    results = []
    trials = 0
    t_start = time.time()
    for ci, ((grp, o, field), sites) in enumerate(sorted(cands.items())):
        if trials >= args.max_trials:
            break
        for val in values:
            ms = timed(with_flip(start, sites, field, val))
            trials += 1
            results.append(dict(group=grp, ordinal=o, field=field, value=val, sites=sites, geom=by_site[sites[0]][0], ms=ms, delta_us=1e3 * (ms - ref)))
            print('A %3d/%d %-6s %3d %-10s -> %d  geom %s  %.4f ms  %+.1f us' % (ci + 1, len(cands), grp, o, field, val, by_site[sites[0]][0][:5], ms, 1e3 * (ms - ref)), flush=True)
        if (ci + 1) % 10 == 0:
            bases.append(timed(start))
            ref = sorted(bases[-3:])[1]
            print('  base re-timed %.4f ms (ref %.4f), %.0f s elapsed' % (bases[-1], ref, time.time() - t_start), flush=True)
    json.dump(dict(key=key, base=bases, results=results), open(args.out + '.stageA.json', 'w'), indent=1)

    # ---- B: cumulative, best first; every decision is an A/B of two alternating pairs -------------------------------------------------
    best = {}
    for r in results:                                      # best value per (group, ordinal, field)
        k = (r['group'], r['ordinal'], r['field'])
        if k not in best or r['ms'] < best[k]['ms']:
            best[k] = r
    order = sorted((r for r in best.values() if r['delta_us'] <= -args.min_gain_us), key=lambda r: r['ms'])[:args.max_stage_b]
    table = {k: dict(v) for k, v in start.items()}
    print('B: %d flips gained alone (the best %d are tried)' % (sum(1 for r in best.values() if r['delta_us'] <= -args.min_gain_us), len(order)), flush=True)
    for r in order:
        cand = with_flip(table, r['sites'], r['field'], r['value'])
        a1, b1, a2, b2 = timed(table), timed(cand), timed(table), timed(cand)
        cur, ms = 0.5 * (a1 + a2), 0.5 * (b1 + b2)
        keep = ms < cur - 1e-3 * args.min_gain_us and max(b1, b2) < max(a1, a2)
        print('B %-6s %3d %-10s -> %d  %.4f / %.4f ms vs %.4f / %.4f  %s' % (r['group'], r['ordinal'], r['field'], r['value'], b1, b2, a1, a2, 'KEEP' if keep else 'drop'), flush=True)
        if keep:
            table = cand
    # ---- C: alternating rounds -----------------------------------------------------------------------------------------------------
    rounds = [(timed({}, 2 * args.iters), timed(table, 2 * args.iters)) for _ in range(3)]
    print('C: (no site plan, site plan) ms per iteration: %s' % [('%.4f' % a, '%.4f' % b) for a, b in rounds], flush=True)
    out = {key: dict(sites=table, measured=dict(base_ms=[round(a, 4) for a, _ in rounds], plan_ms=[round(b, 4) for _, b in rounds],
                                                trials=trials, tool='tools/site_sweep.py'))}
    json.dump(out, open(args.out, 'w'), indent=1, sort_keys=True)
    print('wrote %s (%d site entries)' % (args.out, len(table)))
    optim.reset_optimizers()
    lib.delete_all_params()


if __name__ == '__main__':
    main()
