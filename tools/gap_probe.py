#!/usr/bin/env python
"""Where a step graph idles: from a rocprofv3 trace directory (--kernel-trace --memory-copy-trace), every idle interval longer than
--min-us between consecutive device activities (kernels and copies merged, overlaps accounted for), with the activities around it.
usage: python tools/gap_probe.py <trace dir> [--min-us 20] [--from-frac 0.5]"""
import csv
import glob
import os
import sys


def main(argv):
    d = argv[0]
    min_us = float(argv[argv.index('--min-us') + 1]) if '--min-us' in argv else 20.0
    frac = float(argv[argv.index('--from-frac') + 1]) if '--from-frac' in argv else 0.5
    ev = []
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:70], r.get('Queue_Id', '')))
    for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s %s B' % (r.get('Direction', ''), r.get('Bytes', r.get('Size', '?'))), ''))
    ev.sort()
    t0 = ev[int(len(ev) * frac)][0]
    ev = [e for e in ev if e[0] >= t0]
    busy_until = ev[0][1]
    n = 0
    for i, e in enumerate(ev[1:], 1):
        gap = (e[0] - busy_until) / 1e3
        if gap >= min_us:
            n += 1
            print('--- idle %.1f us' % gap)
            for j in range(max(0, i - 4), min(len(ev), i + 3)):
                s, t, name, q = ev[j]
                print('   %s %10.2f us  dur %7.2f  q=%s  %s' % ('>>' if j == i else '  ', (s - t0) / 1e3, (t - s) / 1e3, q, name))
            if n >= 12:
                break
        busy_until = max(busy_until, e[1])


if __name__ == '__main__':
    main(sys.argv[1:])
