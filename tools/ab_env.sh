#!/bin/bash
# same-box A/B of environment switches on one build: usage  bash tools/ab_env.sh "<bench args>" "VAR=1" ["VAR2=x" ...]
# prints ms per step for the plain run and each setting, three alternating rounds
ARGS="--steps 300 --warmup 20 --no-cpu-baseline --no-kernel-profile --no-variants --repeats 0 $1"; shift
ms() { env $1 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'], d.get('launches_per_step'))"; }
for i in 1 2 3; do
  echo "plain      $(ms _X=0)"
  for s in "$@"; do echo "$s   $(ms $s)"; done
done
