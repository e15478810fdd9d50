B="python bench.py --steps 200 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 2"
run() { $B $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('repeat_ms_per_step'))"; }
GGAN_NO_PAIR_NETS=1 run nopair
GGAN_PAIR_TARGET_WGS=128 run pair_t128
GGAN_PAIR_NOFORK=1 GGAN_PAIR_TARGET_WGS=0 run pair_nofork_t0
GGAN_PAIR_NOFORK=1 GGAN_PAIR_TARGET_WGS=256 run pair_nofork_t256
GGAN_NO_PAIR_NETS=1 run nopair
