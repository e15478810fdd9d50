#!/bin/bash
# filter-gradient A/B on the GPU box: op tests, then per-shape timings of the eight-wave (GGAN_WGRAD_W4=0) and four-wave kernels
cd "$(dirname "$0")/.."
out=gpurun_out/wgrad_ab; mkdir -p $out
python -m pytest tests/test_ops_gpu.py -x -q -k "conv_family or filter or wgrad or plan" 2>&1 | tail -5 > $out/tests.log
for w4 in 0 1; do
  for B in 64 128; do
    GGAN_WGRAD_W4=$w4 python tools/bench_conv.py --ops wgrad --B $B --shapes B,C,F2 2>&1 | grep wgrad | sed "s/^/W4=$w4 B=$B /" >> $out/times.log
  done
done
for w4 in 0 1; do
  for sh in B C; do
    echo "== W4=$w4 shape $sh N=64" >> $out/stamps.log
    GGAN_WGRAD_W4=$w4 python tools/stamps.py wgrad $sh 64 >> $out/stamps.log 2>&1
  done
done
cat $out/tests.log $out/times.log $out/stamps.log
