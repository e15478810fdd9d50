#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launches, total/avg/min/max duration.
usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/<round>_<what>.md"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'(\w+_kernel|\w+_k)\b', name)
    n = name
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n[:110]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, (end-start) as dur from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0, 1e30, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---|---|---|---|---|---|')
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |' % (short(name), a[0], a[1] / 1e6, a[1] / a[0] / 1e3,
                                                               a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    print('\ntotal kernel time: %.3f ms over %d dispatches' % (tot / 1e6, len(rows)))


if __name__ == '__main__':
    main(sys.argv[1])
