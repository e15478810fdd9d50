#!/bin/bash
# tools/variant_lib.sh NAME "-DGGAN_ABL=1 ..." : libggan.so with conv_corr.hip compiled under extra defines -> _variants/libggan_NAME.so
# (experiments only; on the GPU box: cp _variants/libggan_NAME.so graphical_gan_amd/libggan.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
src=${SRC:-conv_corr}
mkdir -p _variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on $@ -c graphical_gan_amd/csrc/$src.hip -o _variants/${src}_$name.o
objs=$(ls graphical_gan_amd/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _variants/libggan_$name.so $objs _variants/${src}_$name.o
echo _variants/libggan_$name.so
