#!/bin/bash
# A/B of HIP runtime switches on the headline iteration (same box, alternating with the default): ms per iteration.
# usage: tools/runtime_env_sweep.sh "VAR=val" "VAR2=val" ...
run() { env "$@" python bench.py --gpus 1 --steps 200 --warmup 20 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" 2>/dev/null || echo FAIL; }
echo "default $(run A=1)"
for kv in "$@"; do
  echo "$kv $(run $kv)   default $(run A=1)"
done
