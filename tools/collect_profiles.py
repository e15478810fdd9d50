#!/usr/bin/env python
"""Copy the summaries of a profiling round from gpurun_out/<tag>*/ into profiles/ (tracked) and rebuild profiles/pmc_traffic.json
(per workload: kernel -> HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes, MFMA utilisation from the SQ passes).
usage: python tools/collect_profiles.py r02d"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
WORK = {'': 'headline', '_face': 'gan-face', '_ali': 'ali', '_ssgan': 'ssgan-moving-mnist', '_gmgan': 'gmgan-cifar10-K10',
        '_ssgan3d': 'ssgan-moving-mnist-3dcnn'}
table = {'_tag': tag}
bid = os.path.join(ROOT, 'gpurun_out', tag, 'build_id.txt')
if os.path.exists(bid):
    table['_build'] = open(bid).read().strip()       # (graphical_gan_amd.build.build_id() on the box that profiled)
for suffix, key in WORK.items():
    src = os.path.join(ROOT, 'gpurun_out', tag + suffix)
    if not os.path.isdir(src):
        continue
    for f in ('kernel_trace.md', 'pmc.md', 'timeline.md', 'critical_path.md', 'bench.json'):
        p = os.path.join(src, f)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(ROOT, 'profiles', '%s%s_%s' % (tag, suffix, f)))
    p = os.path.join(src, 'pmc_traffic.json')
    if os.path.exists(p):
        table[key] = json.load(open(p))
json.dump(table, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1, sort_keys=True)
print(sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.startswith(tag)))
