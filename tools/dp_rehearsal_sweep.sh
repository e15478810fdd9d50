run() { env "$@" python bench.py --gpus 1 --steps 100 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 0 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" 2>/dev/null || echo FAIL; }
export GGAN_FORCE_ALLREDUCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541
echo "two buckets $(run A=1)"; echo "one bucket $(run GGAN_ONE_BUCKET=1)"; echo "two buckets $(run A=1)"; echo "one bucket $(run GGAN_ONE_BUCKET=1)"; echo "generator step split $(run GGAN_GEN_TWO_BUCKETS=1)"; echo "default $(run A=1)"
