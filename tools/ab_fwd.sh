python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "conv_family or conv_fused or masked or planned_for" 2>&1 | tail -3
for B in 64 128; do
 for cfgenv in "GGAN_CORR_NSTG=2 GGAN_CORR_XTAB=0" "GGAN_CORR_NSTG=3 GGAN_CORR_XTAB=0" "GGAN_CORR_NSTG=3 GGAN_CORR_XTAB=1"; do
  echo "== B=$B $cfgenv"; env $cfgenv python tools/bench_conv.py --B $B --shapes B,C,F2 --ops fwd 2>&1 | grep -v amdgpu.ids
 done
done
for s in B C; do for n in 64 128; do python tools/stamps.py fwd $s $n 2>&1 | grep -v amdgpu.ids | grep -v chunk[1-6]; done; done
B="python bench.py --steps 200 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 2"
run() { $B $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('repeat_ms_per_step'))"; }
for i in 1 2; do
GGAN_CORR_NSTG=2 GGAN_CORR_XTAB=0 run base
run new
done
for d in "--dataset face" "--mode local_ep"; do
GGAN_CORR_NSTG=2 GGAN_CORR_XTAB=0 run "base $d" "$d"
run "new $d" "$d"
done
