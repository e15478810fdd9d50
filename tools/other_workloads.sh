#!/bin/bash
# Bench lines of the non-headline workloads (same build, 1 GPU) -> gpurun_out/<tag>_other_workloads.json (a JSON list).
# usage: gpurun -- 'bash tools/other_workloads.sh r01k'
TAG=${1:-other}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
OUT=$R/gpurun_out/${TAG}_other_workloads.json
: > $OUT.lines
for a in "--mode local_ep" "--mode wali-gp" "--mode wali" "--dataset svhn --mode local_ep" "--dataset mnist" "--dataset mnist --mode local_ep" \
         "--dataset face" "--dataset face --mode local_ep" "--dataset moving_mnist" "--dataset chairs" "--host-feed" \
         "--dataset mnist --mode local_ep --batch-size 50" "--dataset face --mode local_ep --batch-size 128" \
         "--mode vegan" "--mode vegan-mmd" "--mode vegan-kl" "--mode vegan-jsd" \
         "--dataset moving_mnist --ssgan-mode ali" "--dataset moving_mnist --ssgan-mode ali:3dcnn"; do
    python bench.py $a --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep '^{' | tail -1 >> $OUT.lines
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1] + '.lines') if l.strip()]
json.dump(rows, open(sys.argv[1], 'w'), indent=1)
for d in rows:
    print('%-110s %10.1f %s  %.3f ms' % (d['config']['workload'][:110], d['value'], d['unit'], d['ms_per_step']))
PY
rm -f $OUT.lines
