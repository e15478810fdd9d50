#!/bin/bash
# default split-K workgroup target of the filter gradient (GGAN_WGRAD_WGS, launches without a plan hint) over the workloads
cd "$(dirname "$0")/.."
for args in "" "--mode ali" "--dataset face --mode ali" "--mode local_ep"; do
  echo "== bench.py $args"
  bash tools/ab_env.sh "$args" GGAN_WGRAD_WGS=64 GGAN_WGRAD_WGS=96 GGAN_WGRAD_WGS=128 GGAN_WGRAD_WGS=192 | sort | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print k, a[k]}'
done
