#!/usr/bin/env python
"""Longest path through a captured step graph (GGAN_GRAPH_DOT dump) with every node weighted by its kernel's average duration inside the
replayed graph (a profiles/<tag>_kernel_trace.md): what the graph's EDGES allow, next to what the replay takes.  Per-kernel totals along
the path say which chain an iteration waits for.
usage: python tools/graph_critical_path.py <file.dot> <kernel_trace.md>"""
import re
import subprocess
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
label, par = {}, {}
for m in re.finditer(r'"(graph_\d+_node_\d+)"\[[^\]]*label="(\d+)\s*\n([^\n"]*)', txt):
    label[m.group(1)] = (int(m.group(2)), m.group(3))
for m in re.finditer(r'"(graph_\d+_node_\d+)"\s*->\s*"(graph_\d+_node_\d+)"', txt):
    par.setdefault(m.group(2), []).append(m.group(1))
dur = {}
for line in open(sys.argv[2]):
    if line.startswith('total kernel time'):
        break                                             # (the per-(kernel, grid) table follows: only the per-kernel averages are wanted)
    c = [x.strip() for x in line.split('|')]
    if len(c) == 9 and c[1] and c[1] != 'kernel' and not c[1].startswith('-'):
        try:
            dur[c[1]] = float(c[4])
        except ValueError:
            pass
names = sorted(set(v[1] for v in label.values() if v[1] != 'MEMCPY'))
short = {}
for n, d in zip(names, subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.strip().split('\n')):
    d = re.sub(r'\(anonymous namespace\)::', '', d)
    d = re.sub(r'^void ', '', d)
    short[n] = re.sub(r'\(.*$', '', d)[:70]
order = sorted(label, key=lambda k: label[k][0])          # capture order is a topological order
D = {k: (4.0 if label[k][1] == 'MEMCPY' else dur.get(short[label[k][1]], 6.0)) for k in order}
best, prev = {}, {}
for k in order:
    p = max(par.get(k, []), key=lambda q: best[q], default=None)
    best[k], prev[k] = (best[p] if p else 0.0) + D[k], p
k = max(order, key=lambda q: best[q])
total, path = best[k], []
while k:
    path.append(k)
    k = prev[k]
print('%d nodes, sum of their durations %.0f us; longest path %.0f us over %d nodes' % (len(order), sum(D.values()), total, len(path)))
c = Counter()
for k in path:
    c[short.get(label[k][1], 'MEMCPY')] += D[k]
for n, v in c.most_common(20):
    print('%8.1f us  %s' % (v, n))
