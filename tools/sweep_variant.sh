#!/bin/bash
# usage: tools/sweep_variant.sh "<bench args>" VAR=val ...   -> ms per iteration, alternating with the default
ARGS="$1"; shift
run() { env "$@" python bench.py --gpus 1 --steps 60 --warmup 8 $ARGS --no-variants --no-cpu-baseline --no-kernel-profile --repeats 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" 2>/dev/null || echo FAIL; }
echo "default $(run A=1)"
for kv in "$@"; do echo "$kv $(run $kv)   default $(run A=1)"; done
