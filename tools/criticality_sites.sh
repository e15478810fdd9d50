#!/bin/bash
# needs a diagnostic build of the library: GGAN_BUILD_DIAG=1 python graphical_gan_amd/build.py --force (the product build has no GGAN_SKIP_KERNELS)
# Per launch SITE: the step re-measured with the k-th of the n launches of a kernel family per iteration left out (GGAN_SKIP_KERNELS
# "name@k/n").  usage (GPU box): bash tools/criticality_sites.sh  -> gpurun_out/criticality_sites.txt
R=${GRAFT_REPO_ROOT:-.}
cd $R; mkdir -p gpurun_out
B="python bench.py --steps 200 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 0 $*"
ms() { timeout 300 env GGAN_SKIP_KERNELS="$1" $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo nan; }
base=$(ms ""); base2=$(ms "")
echo "baseline $base $base2" | tee gpurun_out/criticality_sites.txt
while IFS='|' read name n; do
  for ((k=0; k<n; k++)); do
    v=$(ms "$name@$k/$n")
    python - "$name@$k/$n" "$base" "$base2" "$v" <<'P' | tee -a gpurun_out/criticality_sites.txt
import sys
k,b,b2,v=sys.argv[1],float(sys.argv[2]),float(sys.argv[3]),float(sys.argv[4])
b=(b+b2)/2
print('%-44s %.4f ms  -> %+.1f us' % (k, v, 1e3*(b-v)))
P
  done
done <<'L'
kernel<false, false|8
kernel<true, false|3
kernel<false, true|3
gemm_group_kernel|1
splitk_reduce|5
head_out_fwd_k|2
noise_fill_k|2
cast_scale|2
L
