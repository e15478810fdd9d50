#!/usr/bin/env python
"""print bench.py's per-kernel table from its JSON line on stdin"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'launches/iter', sum(k['launches_per_iter'] for k in d['kernels']),
      'sum ms', round(sum(k['ms_per_iter'] for k in d['kernels']), 3))
for k in d['kernels']:
    print('  %-34s %5.1f %8.4f %7.2f' % (k['name'], k['launches_per_iter'], k['ms_per_iter'], k['avg_us']))
