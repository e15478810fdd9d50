#!/bin/bash
# L2 behaviour of the conv kernels at the hot-path shapes: hit / miss / fabric read requests per kernel (rocprofv3 PMC, separate passes, no tracing)
# usage (GPU box): bash tools/l2_probe.sh <tag> [bench_conv args]
export TMPDIR=/tmp
TAG=${1:-l2}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
rocprofv3 -L > $O/avail.txt 2>&1
grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" $O/avail.txt | sort -u > $O/tcc_names.txt
S="python tools/bench_conv.py $*"
$S > $O/plain.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/p1 -o p -- $S > $O/p1.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum --output-format csv -d $O/p2 -o p -- $S > $O/p2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/p3 -o p -- $S > $O/p3.log 2>&1
python - <<PY
import csv, glob, collections
for p in ('p1','p2','p3'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('$O/%s/**/*counter_collection.csv' % p, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', p)
    for k, d in sorted(acc.items()):
        if not any(s in k for s in ('corr_kernel', 'wgrad', 'dg16', 'splitk')): continue
        print(k, ' '.join('%s=%.3g(n%d)' % (c, sum(v)/len(v), len(v)) for c, v in sorted(d.items())))
PY
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
