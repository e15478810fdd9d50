#!/bin/bash
# HBM-side fetch traffic of the conv kernels with / without the XCD-aware tile order (GGAN_CORR_XCD=0: plain order), headline workload
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
S="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-profile --no-variants --no-graph $*"
for v in auto 0; do
  O=gpurun_out/xcd_$v; rm -rf $O; mkdir -p $O
  if [ $v = 0 ]; then export GGAN_CORR_XCD=0; else unset GGAN_CORR_XCD; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $S > $O/log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- $S >> $O/log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- $S >> $O/log 2>&1
  python tools/pmc_summary.py $O/trace $O/pmc_fetch $O/pmc_write --window 1.0 --json $O/t.json > $O/pmc.md 2>/dev/null
  echo "== GGAN_CORR_XCD=$v"; grep -E "corr_kernel|dg16_kernel" $O/pmc.md | awk -F'|' '{printf "%-40s grid %8s n %5s us %7s fetch KiB %10s write KiB %10s\n",$2,$3,$4,$5,$8,$9}'
  find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
done
