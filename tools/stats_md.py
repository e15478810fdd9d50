#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV (or its --stats summary) -> markdown table: kernel | calls | total ms | avg/min/max us | %.
usage: python tools/stats_md.py <dir or *_kernel_trace.csv / *_kernel_stats.csv>"""
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


def main(path):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, '**', '*kernel_trace.csv'), recursive=True)
    agg = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if 'Start_Timestamp' in r:
                agg[short(r['Kernel_Name'])].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
            elif 'TotalDurationNs' in r:        # --stats summary
                n = int(r['Calls'])
                agg[short(r['Name'])] += [float(r['TotalDurationNs']) / n / 1e3] * n
    tot = sum(sum(v) for v in agg.values())
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print('| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |' % (k, len(v), sum(v) / 1e3, sum(v) / len(v), min(v), max(v), 100 * sum(v) / tot))
    print('\ntotal kernel time: %.3f ms over %d dispatches' % (tot / 1e3, sum(len(v) for v in agg.values())))


if __name__ == '__main__':
    main(sys.argv[1])
