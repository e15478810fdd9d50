#!/usr/bin/env python
"""Merge the flip sets a -m gpu run reported (GGAN_FLIP_REPORT=<file>, tests/test_golden_full_gpu.py) into
tests/golden/full_flips.json: fixture -> step -> list of distinct flip sets (a flip set = sorted [[kink key, flat position], ...]).
Afterwards: python tests/golden/make_golden_full.py <fixtures> writes the float64 variant of every set into the fixture.
usage: python tools/merge_flips.py gpurun_out/flips.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, 'tests', 'golden', 'full_flips.json')
try:
    tab = json.load(open(path))
except OSError:
    tab = {}
changed = set()
for line in open(sys.argv[1]):
    r = json.loads(line)
    if not r['flips']:
        continue
    sets = tab.setdefault(r['fixture'], {}).setdefault(r['step'], [])
    if r['flips'] not in sets:
        sets.append(r['flips'])
        changed.add(r['fixture'])
    print('%-24s %-6s %-5s %d flip(s), worst %.2e rms  %s' % (r['fixture'], r['config'], r['step'], len(r['flips']), r['worst_rms'], r['flips']))
json.dump(tab, open(path, 'w'), indent=1, sort_keys=True)
print('fixtures to regenerate:', ' '.join(sorted(changed)) or '(none)')
