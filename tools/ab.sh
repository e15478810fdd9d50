#!/bin/bash
# same-box A/B: bench the working tree against the snapshot in _prev/ (git archive of an older commit), alternating runs
ARGS="--steps 200 --warmup 10 --no-cpu-baseline --no-kernel-profile $*"
for i in 1 2; do
  (cd _prev && python bench.py $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['ms_per_step'])")
  python bench.py $ARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['ms_per_step'])"
done
