#!/usr/bin/env python
"""Kernel timeline of ONE graph-replayed iteration out of a rocprofv3 kernel-trace CSV (iteration = two adam_k launches).
usage: python tools/timeline.py <x_kernel_trace.csv> [iteration index from the end, default 8]"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 8
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); return re.sub(r'\(.*$', '', n)[:46]
idx = [i for i, r in enumerate(rows) if 'adam_k' in r['Kernel_Name']]
a, b = idx[-2 * back - 1] + 1, idx[-2 * back + 1] + 1
it = rows[a:b]
t0 = int(it[0]['Start_Timestamp']); prev = None; tot = 0
cat = defaultdict(lambda: [0, 0.0])
for r in it:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    prev = e; tot += e - s
    c = cat[short(r['Kernel_Name'])]; c[0] += 1; c[1] += (e - s) / 1e3
    print('%8.1f %6.1f gap %5.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(r['Kernel_Name'])))
print('kernels', len(it), 'sum us', tot / 1e3, 'span us', (int(it[-1]['End_Timestamp']) - t0) / 1e3)
for k, v in sorted(cat.items(), key=lambda kv: -kv[1][1]):
    print('%-48s %3d %8.1f' % (k, v[0], v[1]))
