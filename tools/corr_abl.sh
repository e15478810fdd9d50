#!/bin/bash
# forward / data-gradient ablations on the GPU box: variants built by tools/variant_lib.sh (SRC=conv_corr, -DGGAN_ABL=bits) -> per-phase stamps
cd "$(dirname "$0")/.."
N=${N:-64}
out=gpurun_out/corr_abl; mkdir -p $out; : > $out/stamps.log
cp graphical_gan_amd/libggan.so /tmp/libggan_orig.so
for v in orig "$@"; do
  [ $v = orig ] || cp _variants/libggan_$v.so graphical_gan_amd/libggan.so
  for op in fwd; do for sh in B C; do
    echo "== variant $v $op shape $sh N=$N" >> $out/stamps.log
    python tools/stamps.py $op $sh $N 2>&1 | grep -v amdgpu.ids >> $out/stamps.log
  done; done
done
cp /tmp/libggan_orig.so graphical_gan_amd/libggan.so
grep "==\|chunk[1256] \|bar0\|loop_end\|stored\|wall" $out/stamps.log
