#!/bin/bash
# role-split filter gradient (GGAN_WGRAD_SPLIT=1) against the four-wave kernel: op tests, per-shape timings, whole-step A/B
cd "$(dirname "$0")/.."
out=gpurun_out/wgrad_split; mkdir -p $out; rm -f $out/*.log
GGAN_WGRAD_SPLIT=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv_family or filter or wgrad or plan" 2>&1 | tail -5 > $out/tests.log
for sp in 0 1; do
  for B in 64 128; do
    GGAN_WGRAD_SPLIT=$sp timeout 300 python tools/bench_conv.py --ops wgrad --B $B --shapes B,C,F2 2>&1 | grep wgrad | sed "s/^/SPLIT=$sp B=$B /" >> $out/times.log
  done
done
bash tools/ab_env.sh "" GGAN_WGRAD_SPLIT=1 > $out/ab_headline.log 2>&1
bash tools/ab_env.sh "--mode ali" GGAN_WGRAD_SPLIT=1 > $out/ab_ali.log 2>&1
cat $out/tests.log $out/times.log $out/ab_headline.log $out/ab_ali.log
