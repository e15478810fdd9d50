B="python bench.py --steps 200 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 2"
run() { $B $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('repeat_ms_per_step'))"; }
for i in 1 2; do
GGAN_DG16=0 run base
GGAN_DG16=1 run dg16
GGAN_DG16=1 GGAN_DG16_KQ=4 GGAN_DG16_MINQ=2 run dg16_kq4
GGAN_DG16=1 GGAN_DG16_MINQ=2 run dg16_min2
done
for d in "--dataset face" "--mode local_ep" "--mode wali-gp"; do
GGAN_DG16=0 run "base $d" "$d"
GGAN_DG16=1 run "dg16 $d" "$d"
done
for s in B C; do for n in 64 128; do python tools/stamps.py dgrad $s $n 2>&1 | grep -v amdgpu.ids; done; done
