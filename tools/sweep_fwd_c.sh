#!/bin/bash
# forward / data-gradient plans for the 128->256 @8->4 layer at 64 images on the 128-workgroup plan (the wali-gp step's side-by-side chains)
cd "$(dirname "$0")/.."
export GGAN_TARGET_WGS=128
echo "== default"; GGAN_TRACE_LAUNCHES=1 python tools/bench_conv.py --ops fwd,dgrad --B 64 --shapes C 2>&1 | grep -E "wall|ggan launch" | sort | uniq -c | sort -rn | head -8
for cfg in 4 5 6 7 8; do for sk in 1 2 4; do
  echo "== fwd cfg $cfg sk $sk: $(GGAN_FWD_CFG=$cfg python tools/bench_conv.py --ops fwd --B 64 --shapes C --sk $sk 2>&1 | grep wall)"
done; done
for kq in 1 2; do for mq in 1 2 3 4; do
  echo "== dgrad dg16 KQ $kq MINQ $mq: $(GGAN_DG16_KQ=$kq GGAN_DG16_MINQ=$mq python tools/bench_conv.py --ops dgrad --B 64 --shapes C 2>&1 | grep wall)"
done; done
echo "== dgrad no dg16: $(GGAN_DG16=0 python tools/bench_conv.py --ops dgrad --B 64 --shapes C 2>&1 | grep wall)"
