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
    print('\n%d kernels, span %.2f us, sum of kernel durations %.2f us' % (len(seg), (prev_end - t0) / 1e3, busy))


if __name__ == '__main__':
    main(sys.argv[1:])
