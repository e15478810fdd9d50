for B in 64 128; do
 for cfgenv in "GGAN_DG16_AHEAD=2" "GGAN_DG16_AHEAD=1"; do
  echo "== B=$B $cfgenv"; env $cfgenv GGAN_DG16_MINQ=2 python tools/bench_conv.py --B $B --shapes B,C,F2 --ops dgrad 2>&1 | grep -v amdgpu.ids
 done
done
GGAN_DG16_AHEAD=1 python tools/stamps.py dgrad B 64 2>&1 | grep -v amdgpu.ids | grep -v "chunk[1-6]"
