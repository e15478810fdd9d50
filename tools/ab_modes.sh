#!/bin/bash
# same-box A/B of an environment switch over the benchmark's workloads: bash tools/ab_modes.sh "VAR=0" [rounds-per-workload via tools/ab_env.sh]
cd "$(dirname "$0")/.."
for args in "" "--mode wali-gp" "--dataset face" "--mode local_ep"; do
  echo "== bench.py $args"
  bash tools/ab_env.sh "$args" "$@"
done
