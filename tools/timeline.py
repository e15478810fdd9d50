#!/usr/bin/env python
"""Ordered kernel timeline of ONE steady-state training iteration from a rocprofv3 --kernel-trace CSV.

A step starts with `noise_fill_k` (the step's single noise launch); an iteration is `--steps-per-iter` consecutive steps
(2 for CRITIC_ITERS = 1: generator step + critic step).  Prints, in start order: offset from the iteration start, duration,
gap to the previous kernel's end (negative: overlapped, i.e. a second stream), grid/workgroup size, registers, LDS, name.
usage: python tools/timeline.py <dir or csv> [--steps-per-iter 2] [--skip 10]"""
import csv
import glob
import os
import re
import sys


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:64]


def main(argv):
    path = argv[0]
    spi = int(argv[argv.index('--steps-per-iter') + 1]) if '--steps-per-iter' in argv else 2
    skip = int(argv[argv.index('--skip') + 1]) if '--skip' in argv else 10
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append(r)
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    starts = [i for i, r in enumerate(rows) if 'noise_fill_k' in r['Kernel_Name']]
    if not starts:            # no marker kernel: show the last 400 dispatches
        starts, spi, skip = [max(0, len(rows) - 400)], 1, 0
    if len(starts) < (skip + 2) * spi:
        skip = max(0, len(starts) // spi - 3)
    # iterations begin at a generator step; the trace's last steps are graph replays: take iteration `skip` from the END
    a = starts[-(skip + 1) * spi - 1] if len(starts) > (skip + 1) * spi else starts[0]
    b = starts[-(skip) * spi - 1] if skip else len(rows)
    seg = rows[a:b]
    t0 = int(seg[0]['Start_Timestamp'])
    prev_end = t0
    print('| # | start us | dur us | gap us | grid | wg | vgpr | lds | kernel |')
    print('|---|---|---|---|---|---|---|---|---|')
    busy = 0.0
    for i, r in enumerate(seg):
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        grid = int(r.get('Grid_Size_X', 0) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        wg = int(r.get('Workgroup_Size_X', 0) or 0) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
        print('| %d | %.2f | %.2f | %.2f | %d | %d | %s | %s | %s |' % (
            i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, grid // max(wg, 1), wg, r.get('VGPR_Count', ''),
            r.get('LDS_Block_Size', ''), short(r['Kernel_Name'])))
        busy += (e - s) / 1e3
        prev_end = max(prev_end, e)
    span = (prev_end - t0) / 1e3
    print('\n%d kernels, span %.2f us, sum of kernel durations %.2f us' % (len(seg), span, busy))
    # kernels in flight over the iteration, and how much of it ONE conv kernel on a half-chip plan (<= 160 workgroups) has the chip to itself
    ev = []
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        grid = int(r.get('Grid_Size_X', 0) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
        wg = int(r.get('Workgroup_Size_X', 0) or 0) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
        nm = short(r['Kernel_Name'])
        half = nm.startswith(('corr_kernel', 'dg16_kernel', 'wgrad4_kernel', 'wgrad_kernel')) and grid // max(wg, 1) <= 160
        ev.append((s, 1, half, nm))
        ev.append((e, -1, half, nm))
    ev.sort(key=lambda x: (x[0], x[1]))
    hist, alone_half, alone_by = {}, 0.0, {}
    live, live_half, names, prev = 0, 0, [], ev[0][0]
    for t, d, half, nm in ev:
        dt = (t - prev) / 1e3
        hist[min(live, 3)] = hist.get(min(live, 3), 0.0) + dt
        if live == 1 and live_half == 1:
            alone_half += dt
            alone_by[names[0]] = alone_by.get(names[0], 0.0) + dt
        prev = t
        live += d
        live_half += d if half else 0
        if d > 0:
            names.append(nm)
        else:
            names.remove(nm)
    tot = sum(hist.values()) or 1.0
    print('kernels in flight (share of the span): ' + ', '.join('%s: %.1f %%' % ('3+' if k == 3 else k, 100 * v / tot) for k, v in sorted(hist.items())))
    print('ONE conv kernel on a half-chip plan (<= 160 workgroups) alone on the chip: %.1f us = %.1f %% of the span' % (alone_half, 100 * alone_half / tot))
    for nm, v in sorted(alone_by.items(), key=lambda kv: -kv[1])[:8]:
        print('    %-44s %.1f us' % (nm, v))


if __name__ == '__main__':
    main(sys.argv[1:])
