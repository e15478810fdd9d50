#!/bin/bash
# filter-gradient ablations on the GPU box: variants built by tools/variant_lib.sh (SRC=conv_wgrad_split, -DGGAN_ABL=bits) -> per-phase stamps
cd "$(dirname "$0")/.."
N=${N:-128}
out=gpurun_out/wgrad_abl; mkdir -p $out; : > $out/stamps.log
cp graphical_gan_amd/libggan.so /tmp/libggan_orig.so
for v in orig "$@"; do
  [ $v = orig ] || cp _variants/libggan_$v.so graphical_gan_amd/libggan.so
  for sh in B C; do
    echo "== variant $v shape $sh N=$N" >> $out/stamps.log
    GGAN_SKIP_BUILD=1 python tools/stamps.py wgrad $sh $N 2>&1 | grep -v amdgpu.ids >> $out/stamps.log
  done
done
cp /tmp/libggan_orig.so graphical_gan_amd/libggan.so
cat $out/stamps.log
