#!/bin/bash
# same-box A/B of whole libraries (tools/variant_lib.sh): usage  bash tools/ab_lib.sh "<bench args>" NAME [NAME ...]
# prints ms per step for the product library and each _variants/libggan_NAME.so, three alternating rounds
cd "$(dirname "$0")/.."
ARGS="--steps 300 --warmup 20 --no-cpu-baseline --no-kernel-profile --no-variants --repeats 0 $1"; shift
cp graphical_gan_amd/libggan.so /tmp/libggan_orig.so
ms() { GGAN_SKIP_BUILD=1 python bench.py $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])"; }
for i in 1 2 3; do
  cp /tmp/libggan_orig.so graphical_gan_amd/libggan.so; echo "product    $(ms)"
  for v in "$@"; do cp _variants/libggan_$v.so graphical_gan_amd/libggan.so; echo "$v   $(ms)"; done
done
cp /tmp/libggan_orig.so graphical_gan_amd/libggan.so
