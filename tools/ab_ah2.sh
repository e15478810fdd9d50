python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "dg16 or masked_data" 2>&1 | tail -2
B="python bench.py --steps 200 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 2"
run() { $B $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('repeat_ms_per_step'))"; }
for i in 1 2; do
GGAN_DG16_AHEAD=2 run ah2
GGAN_DG16_AHEAD=1 run ah1
done
for d in "--dataset face" "--mode local_ep" "--mode wali-gp"; do
GGAN_DG16_AHEAD=2 run "ah2 $d" "$d"
GGAN_DG16_AHEAD=1 run "ah1 $d" "$d"
done
