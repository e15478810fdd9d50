#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (--output-format csv) per (kernel, grid size): mean counter value per dispatch, mean duration.

usage: python tools/pmc_summary.py <trace dir> <pmc dir> [more pmc dirs...] [--json out.json] [--window 0.5]

A kernel that a step launches on two problem sizes (the critic on [fake; real] = 128 images, the Extractor on 64) is two rows: counters
and durations are never averaged over different problems.  The kernel-level figures (`avg_us_in_graph`, `traffic_bytes`,
`mfma_util_pct` in the JSON) are the per-grid rows weighted by the launch mix of the TIMED configuration: the dispatches in the last
`--window` fraction of the kernel trace (graph replays only -- the trace's head holds the eager warm-up iterations and the capture),
whatever mix the counter passes ran (they run eager iterations on the graphs' launch plan, warm-up included).  Round-4 review: the
counter passes' own mix (9 : 31) described another kernel population than the timed 3 : 2.
Grid size = work-items per launch (rocprofv3's Grid_Size = grid x block), the same number libggan's own profiler records."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:60]


# FETCH_SIZE on gfx950 counts 64 B per 128-B fabric request: tools/fetch_calib.hip reads a known GiB through each access path the
# kernels here stage their operands with and the counter reports exactly half of it on ALL of them -- coalesced dword loads, 16-byte
# loads, LDS-DMA in its dword and its 16-byte form (profiles/r05_fetch_calib.json: 2.0000 / 2.0000 / 1.9999 / 2.0000 bytes per counted
# byte).  WRITE_SIZE is exact for a streamed GiB on every store path (dword, 16-byte global, 16-byte buffer stores: 1.0000) -- what it
# counts is bytes leaving L2 for the fabric DURING the dispatch, so a kernel whose few-MB output is still dirty in the eight 4-MB L2s
# when it ends shows less than it wrote (round 4: exactly half of a 2 / 4 MB forward output): the write-back happens under a later
# dispatch.  `write_bytes` below is therefore a lower bound for small outputs; `traffic_bytes` = corrected fetches + counted writes.
FETCH_CORRECTION = 2.0


def collect(dirs, window):
    ctr = defaultdict(lambda: defaultdict(list))       # (kernel, grid) -> counter -> [value per dispatch]
    util = defaultdict(list)                           # (kernel, grid) -> [MFMA busy % per dispatch]
    trace = []                                         # (start, kernel, grid, duration ns)
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per = defaultdict(float)                   # (dispatch, key, counter) -> summed over dimensions
            span = {}
            for r in csv.DictReader(open(f)):
                key = (short(r['Kernel_Name']), int(float(r['Grid_Size'])))
                per[(r['Dispatch_Id'], key, r['Counter_Name'])] += float(r['Counter_Value'])
                if r.get('End_Timestamp'):
                    span[(r['Dispatch_Id'], key)] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            for (disp, key, c), v in per.items():
                ctr[key][c].append(v)
                # MFMA utilisation per DISPATCH (busy cycles / that dispatch's own duration at 2.4 GHz x 1024 SIMDs); the tables report
                # the median, so that one dispatch stretched by something outside the kernel cannot drag the figure
                if c == 'SQ_VALU_MFMA_BUSY_CYCLES' and span.get((disp, key)):
                    util[key].append(100.0 * v / (span[(disp, key)] * 2.4 * 1024))
            for (disp, key), v in span.items():
                ctr[key]['_dur_ns_under_pmc'].append(v)
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
                trace.append((float(r['Start_Timestamp']), short(r['Kernel_Name']), grid, float(r['End_Timestamp']) - float(r['Start_Timestamp'])))
    trace.sort()
    if trace:
        t0, t1 = trace[0][0], trace[-1][0]
        cut = t1 - window * (t1 - t0)
        trace = [t for t in trace if t[0] >= cut]
    dur = defaultdict(list)
    for _, k, g, d_ns in trace:
        dur[(k, g)].append(d_ns)
    return ctr, util, dur


def mean(v):
    return sum(v) / len(v) if v else None


def median(v):
    return sorted(v)[len(v) // 2] if v else None


def main(argv):
    jout, window = None, 0.5
    if '--json' in argv:
        i = argv.index('--json'); jout = argv[i + 1]; argv = argv[:i] + argv[i + 2:]
    if '--window' in argv:
        i = argv.index('--window'); window = float(argv[i + 1]); argv = argv[:i] + argv[i + 2:]
    ctr, util, dur = collect(argv, window)
    names = sorted(set(c for key in ctr for c in ctr[key]))
    keys = sorted(set(ctr) | set(dur), key=lambda key: (-sum(dur.get(key, [0.0])), key))
    print('| kernel | grid (work-items) | launches in the timed window | avg us (graph replay) | pmc dispatches | ~MFMA_util_% (median) | ' + ' | '.join(names) + ' |')
    print('|---|---|---|---|---|---|' + '---|' * len(names))
    fmt = lambda v, f='%.4g': '' if v is None else f % v
    for key in keys:
        if not dur.get(key) and not any(key[0] == k2[0] and dur.get(k2) for k2 in keys):
            continue                                   # (a kernel of the set-up only: not part of the timed step)
        n_p = max([len(v) for v in ctr[key].values()] or [0])
        print('| %s | %d | %d | %s | %d | %s | ' % (key[0], key[1], len(dur.get(key, [])), fmt(mean(dur.get(key, [])) and mean(dur[key]) / 1e3, '%.2f'),
                                                   n_p, fmt(median(util.get(key, [])), '%.1f'))
              + ' | '.join(fmt(mean(ctr[key].get(c, []))) for c in names) + ' |')
    if not jout:
        return
    tab = {}
    kernels = sorted(set(k for k, _ in keys))
    for k in kernels:
        grids = sorted(g for kk, g in keys if kk == k)
        by_grid, unmatched = {}, []
        for g in grids:
            key = (k, g)
            c = ctr.get(key, {})
            f = mean(c.get('FETCH_SIZE', []))
            w = mean(c.get('WRITE_SIZE', []))
            rec = dict(launches_in_window=len(dur.get(key, [])),
                       avg_us_in_graph=round(mean(dur[key]) / 1e3, 2) if dur.get(key) else None,
                       fetch_bytes_raw=round(1024.0 * f) if f is not None else None,
                       write_bytes=round(1024.0 * w) if w is not None else None,
                       traffic_bytes=round(1024.0 * (f * FETCH_CORRECTION + w)) if (f is not None and w is not None) else None,
                       mfma_util_pct=round(median(util[key]), 2) if util.get(key) else None,
                       mfma_busy_cycles=round(mean(c['SQ_VALU_MFMA_BUSY_CYCLES'])) if c.get('SQ_VALU_MFMA_BUSY_CYCLES') else None,
                       pmc_dispatches=max([len(v) for v in c.values()] or [0]))
            by_grid[str(g)] = rec
            if rec['launches_in_window'] and rec['traffic_bytes'] is None:
                unmatched.append(g)
        # kernel-level figures = the rows weighted by the timed window's launch mix
        n = sum(r['launches_in_window'] for r in by_grid.values())
        agg = dict(by_grid=by_grid, launches_in_window=n, mix='launches of the last %.0f %% of the kernel trace (graph replays)' % (100 * window))
        if n:
            wsum = lambda field: (sum(r['launches_in_window'] * r[field] for r in by_grid.values() if r['launches_in_window'] and r[field] is not None),
                                  sum(r['launches_in_window'] for r in by_grid.values() if r['launches_in_window'] and r[field] is not None))
            for field in ('avg_us_in_graph', 'traffic_bytes', 'fetch_bytes_raw', 'write_bytes'):
                s, m = wsum(field)
                agg[field] = round(s / m, 2) if m else None
            # utilisation: weighted by the TIME each row contributes
            tw = [(r['launches_in_window'] * r['avg_us_in_graph'], r['mfma_util_pct']) for r in by_grid.values()
                  if r['launches_in_window'] and r['avg_us_in_graph'] and r['mfma_util_pct'] is not None]
            agg['mfma_util_pct'] = round(sum(t * u for t, u in tw) / sum(t for t, _ in tw), 2) if tw else None
            agg['fetch_correction'] = FETCH_CORRECTION
            if unmatched:
                agg['grids_without_counters'] = unmatched
        tab[k] = agg
    json.dump(tab, open(jout, 'w'), indent=1, sort_keys=True)
    miss = {k: v['grids_without_counters'] for k, v in tab.items() if v.get('grids_without_counters')}
    if miss:
        sys.stderr.write('pmc_summary: timed launches without a counter sample of the same grid: %s\n' % miss)


if __name__ == '__main__':
    main(sys.argv[1:])
