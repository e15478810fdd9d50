#!/bin/bash
# same-box A/B of two builds of libggan.so: the in-tree one against _variants/libggan_<name>.so (tools/variant_lib.sh), alternating
# usage (GPU box): bash tools/ab_lib.sh <name> [pytest -k expression for a correctness pass of the variant] [bench args]
R=${GRAFT_REPO_ROOT:-.}; cd $R
name=$1; kexpr=$2; shift; shift
B="python bench.py --steps 200 --warmup 10 --no-variants --no-cpu-baseline --no-kernel-profile --repeats 2 $*"
cp graphical_gan_amd/libggan.so /tmp/libggan_base.so
run() { $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d.get('repeat_ms_per_step'))"; }
if [ -n "$kexpr" ]; then
  cp _variants/libggan_$name.so graphical_gan_amd/libggan.so
  python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "$kexpr" 2>&1 | tail -2
fi
for i in 1 2 3; do
  cp /tmp/libggan_base.so graphical_gan_amd/libggan.so; run base
  cp _variants/libggan_$name.so graphical_gan_amd/libggan.so; run $name
done
cp /tmp/libggan_base.so graphical_gan_amd/libggan.so
