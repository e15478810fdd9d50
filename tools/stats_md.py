#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats (csv) -> markdown table.  usage: python tools/stats_md.py <x_kernel_stats.csv>"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('| kernel | calls | total ms | avg us | min us | max us | % |')
print('|---|---|---|---|---|---|---|')
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)[:70]
    print('| %s | %s | %.3f | %.2f | %.2f | %.2f | %.1f |' % (n, r['Calls'], float(r['TotalDurationNs']) / 1e6,
          float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, float(r['Percentage'])))
print('\ntotal kernel time: %.3f ms over %d dispatches' % (tot / 1e6, sum(int(r['Calls']) for r in rows)))
