#!/bin/bash
# GPU box: run the calibration kernels under rocprofv3 (counters only) and write bytes-per-counted-byte factors per access path.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/fetch_calib
mkdir -p $O; cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- tools/_fetch_calib > $O/run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- tools/_fetch_calib > $O/run_w.log 2>&1
python - "$O" <<'P'
import csv, glob, json, os, sys
from collections import defaultdict
O = sys.argv[1]
bytes_per_launch = float(1 << 30)
res = {}
for ctr in ('fetch', 'write'):
    acc = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(O, ctr, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'].split('(')[0]][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, d in acc.items():
        v = sorted(d.values())[len(d) // 2] * 1024.0       # median dispatch, KiB -> bytes
        res.setdefault(k, {})[ctr.upper() + '_SIZE_bytes'] = v
for k, d in res.items():
    store = k.strip().startswith('st_')
    key = 'WRITE_SIZE_bytes' if store else 'FETCH_SIZE_bytes'
    if d.get(key, 0) > 0:
        if 'seg_k' in k:
            d['useful_bytes'] = bytes_per_launch / 2
        d['true_bytes'] = bytes_per_launch
        d['bytes_per_counted_byte'] = round(bytes_per_launch / d[key], 4)
json.dump(res, open(os.path.join(os.path.dirname(O), 'fetch_calib.json'), 'w'), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
P
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
