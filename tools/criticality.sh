#!/bin/bash
# needs a diagnostic build of the library: GGAN_BUILD_DIAG=1 python graphical_gan_amd/build.py --force (the product build has no GGAN_SKIP_KERNELS)
# What does each kernel family contribute to the CRITICAL PATH of an iteration?  Re-measure the step with that family not launched
# at all (GGAN_SKIP_KERNELS: results are garbage, the timing is not) and report the difference to the full step.
# usage (GPU box): bash tools/criticality.sh [bench args]   -> gpurun_out/criticality.txt
R=${GRAFT_REPO_ROOT:-.}
cd $R; mkdir -p gpurun_out
B="python bench.py --steps 200 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 0 $*"
ms() { timeout 300 env GGAN_SKIP_KERNELS="$1" $B 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null || echo nan; }
base=$(ms "")
base2=$(ms "")
echo "baseline $base $base2" | tee gpurun_out/criticality.txt
for k in "noise_fill_k" "cast_scale" "gemm_kernel<false, false" "gemm_kernel<true, false" "gemm_kernel<false, true" "gemm_group_kernel" \
         "thin_fwd_kernel" "thin_dgrad_kernel" "thin_wgrad_kernel" "bn_fwd_rows_k" "bn_fwd_nchw" "bn_bwd_nchw" "bn_bwd_rows_k" \
         "splitk_reduce" "head_out_fwd_k" "bce_head_bwd_k" "act_bwd_chansum_k" "pack_adam_k" "wgrad_kernel" "corr_kernel<0" \
         "corr_kernel<1" "corr_kernel<2" "gemm_;splitk_;head_out;bce_head" "bn_" "thin_"; do
  v=$(ms "$k")
  python - "$k" "$base" "$v" <<'P' | tee -a gpurun_out/criticality.txt
import sys
k,b,v=sys.argv[1],float(sys.argv[2]),float(sys.argv[3])
print('%-40s %.4f ms  -> on the critical path: %+.1f us (%.1f %%)' % (k, v, 1e3*(b-v), 100*(b-v)/b))
P
done
