#!/usr/bin/env python
"""Read a hipGraphDebugDotPrint dump of a captured step graph (GGAN_GRAPH_DOT=<file>, engine._dump_graph_dot): print every kernel node with
its parents, and the nodes with more than one parent (the joins: where one chain of launches waits for another).
usage: python tools/graph_edges.py <file.dot> [substring of the kernels to show]"""
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ''
label = {}
for m in re.finditer(r'"?([\w\.]+)"?\s*\[([^\]]*)\]', txt):
    lab = re.search(r'label\s*=\s*"([^"]*)"', m.group(2))
    if lab:
        label[m.group(1)] = re.sub(r'\\n|\\l', ' ', lab.group(1))[:90]
parents = {}
for m in re.finditer(r'"?([\w\.]+)"?\s*->\s*"?([\w\.]+)"?', txt):
    parents.setdefault(m.group(2), []).append(m.group(1))
print('%d nodes, %d edges, %d joins' % (len(label), sum(len(v) for v in parents.values()), sum(1 for v in parents.values() if len(v) > 1)))
for n in label:
    ps = parents.get(n, [])
    if want and want not in label[n] and not any(want in label.get(p, '') for p in ps):
        continue
    print('%-10s %-70s <- %s' % (n, label[n], ' | '.join('%s %s' % (p, label.get(p, '?')[:40]) for p in ps)))
