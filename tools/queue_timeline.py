#!/usr/bin/env python
"""One step of a rocprofv3 kernel trace with queue ids and idle gaps per queue: python tools/queue_timeline.py <trace dir> [n kernels]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 95
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'noise_fill' in r['Kernel_Name']]
s = idx[-7]
t0 = int(rows[s]['Start_Timestamp'])
busy_end = t0
for r in rows[s:s + n]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    idle = max(0, st - busy_end)
    print('%8.1f %6.1f idle %5.1f q=%s %s' % ((st - t0) / 1e3, (en - st) / 1e3, idle / 1e3, r['Queue_Id'], r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]))
    busy_end = max(busy_end, en)
