#!/usr/bin/env python
"""Summarise rocprofv3 CSV output (--output-format csv): per kernel, mean counter value per dispatch and mean duration.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv / *_kernel_trace.csv> [more dirs...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:60]


def main(dirs):
    jout = None
    if '--json' in dirs:
        i = dirs.index('--json'); jout = dirs[i + 1]; dirs = dirs[:i] + dirs[i + 2:]
    ctr = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))     # kernel -> counter -> [sum, dispatches]
    dur = defaultdict(lambda: [0.0, 0])
    util = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per = defaultdict(float)                               # (dispatch, kernel, counter) -> summed over dimensions
            span = {}
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                per[(r['Dispatch_Id'], k, r['Counter_Name'])] += float(r['Counter_Value'])
                if 'Start_Timestamp' in r and r.get('End_Timestamp'):
                    span[(r['Dispatch_Id'], k)] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            for (_, k, c), v in per.items():
                a = ctr[k][c]; a[0] += v; a[1] += 1
            for (_, k), v in span.items():
                a = ctr[k]['_dur_ns_under_pmc']; a[0] += v; a[1] += 1
            # MFMA utilisation per DISPATCH (busy cycles / that dispatch's own duration); the table reports the median, so that one
            # dispatch stretched by something outside the kernel (r02o_face: a 118 us outlier of a 22 us kernel read as 9 %) cannot
            # drag the figure
            for (disp, k, c), v in per.items():
                if c == 'SQ_VALU_MFMA_BUSY_CYCLES' and span.get((disp, k)):
                    util[k].append(100.0 * v / (span[(disp, k)] * 2.4 * 1024))
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                a = dur[short(r['Kernel_Name'])]
                a[0] += float(r['End_Timestamp']) - float(r['Start_Timestamp']); a[1] += 1
    # derived columns (MI355X: 256 CUs x 4 SIMDs, 2.4 GHz; FETCH_SIZE / WRITE_SIZE are KiB)
    for k in ctr:
        c = ctr[k]
        if util[k]:
            u = sorted(util[k])
            c['~MFMA_util_%'] = [u[len(u) // 2], 1]
    names = sorted(set(c for k in ctr for c in ctr[k] if ctr[k][c][1]))
    print('| kernel | dispatches | avg us (trace) | ' + ' | '.join(names) + ' |')
    print('|---|---|---|' + '---|' * len(names))
    keys = sorted(set(ctr) | set(dur), key=lambda k: -dur[k][0] if k in dur else 0)
    for k in keys:
        n = max([ctr[k][c][1] for c in ctr[k]] + [dur[k][1]])
        us = '%.2f' % (dur[k][0] / dur[k][1] / 1e3) if dur[k][1] else ''
        print('| %s | %d | %s | ' % (k, n, us) + ' | '.join(
            ('%.4g' % (ctr[k][c][0] / ctr[k][c][1]) if ctr[k][c][1] else '') for c in names) + ' |')
    if jout:
        import json
        # FETCH_SIZE on gfx950 counts 64 B per 128-B fabric request: tools/fetch_calib.hip reads a known GiB through each access path
        # the kernels here stage their operands with and the counter reports exactly half of it on ALL of them -- coalesced dword
        # loads, 16-byte loads, LDS-DMA (buffer_load ... lds) in its dword and its 16-byte form (profiles/r03_fetch_calib.json:
        # 2.0000 / 2.0000 / 1.9999 / 1.9999 bytes per counted byte); WRITE_SIZE reports a GiB fill as a GiB.  One factor for every kernel.
        FETCH_CORRECTION = 2.0
        tab = {}
        for k in ctr:
            c = ctr[k]
            if not (c['FETCH_SIZE'][1] and c['WRITE_SIZE'][1]):
                continue
            f = 1024.0 * c['FETCH_SIZE'][0] / c['FETCH_SIZE'][1]
            w = 1024.0 * c['WRITE_SIZE'][0] / c['WRITE_SIZE'][1]
            corr = FETCH_CORRECTION
            tab[k] = dict(fetch_bytes_raw=round(f), write_bytes=round(w), fetch_correction=corr,
                          traffic_bytes=round(f * corr + w),
                          mfma_util_pct=round(c['~MFMA_util_%'][0], 2) if c['~MFMA_util_%'][1] else None,
                          avg_us_in_graph=round(dur[k][0] / dur[k][1] / 1e3, 2) if dur[k][1] else None)
        json.dump(tab, open(jout, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main(sys.argv[1:])
