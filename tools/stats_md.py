#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> markdown: per kernel (calls, total ms, avg / min / max us, %) and per (kernel, grid).

Only the dispatches in the LAST `--window` fraction of the trace are counted (default 0.5): the head of a bench.py trace holds the eager
warm-up iterations, the dress rehearsal and the capture, the tail is graph replays only -- the timed configuration.  This is the same
window tools/pmc_summary.py takes `avg_us_in_graph` from, so the first row here is the kernel bench.py's `roofline` names (dominant by
total time inside the graph-replayed step) and `roofline.avg_launch_us` is that row's "avg us".
usage: python tools/stats_md.py <dir or *_kernel_trace.csv> [--window 0.5]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:70]


def main(argv):
    window = 0.5
    if '--window' in argv:
        i = argv.index('--window'); window = float(argv[i + 1]); argv = argv[:i] + argv[i + 2:]
    path = argv[0]
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if 'Start_Timestamp' not in r:
                continue
            grid = int(r.get('Grid_Size_X', 0) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
            rows.append((float(r['Start_Timestamp']), short(r['Kernel_Name']), grid, (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3))
    rows.sort()
    n_all = len(rows)
    if rows:
        t0, t1 = rows[0][0], rows[-1][0]
        cut = t1 - window * (t1 - t0)
        rows = [r for r in rows if r[0] >= cut]
    agg, agg_g = defaultdict(list), defaultdict(list)
    for _, k, g, us in rows:
        agg[k].append(us)
        agg_g[(k, g)].append(us)
    tot = sum(sum(v) for v in agg.values()) or 1.0
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print('| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |' % (k, len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    print('\ntotal kernel time: %.3f ms over %d dispatches' % (tot / 1e3, sum(len(v) for v in agg.values())))
    print('(kernels of the graph-replayed step: the %d dispatches in the last %.0f %% of the trace, of %d; the head is eager warm-up, rehearsal and capture)'
          % (len(rows), 100 * window, n_all))
    print('\nper (kernel, grid = work-items per launch): one row per problem size\n')
    print('| kernel | grid | calls | total ms | avg us | min us | max us |')
    print('|---|---|---|---|---|---|---|')
    for (k, g), v in sorted(agg_g.items(), key=lambda kv: -sum(kv[1])):
        print('| %s | %d | %d | %.3f | %.2f | %.2f | %.2f |' % (k, g, len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v)))


if __name__ == '__main__':
    main(sys.argv[1:])
