cp graphical_gan_amd/libggan.so /tmp/base.so
for v in base wg_abl1 wg_abl5; do
  if [ $v = base ]; then cp /tmp/base.so graphical_gan_amd/libggan.so; else cp _variants/libggan_$v.so graphical_gan_amd/libggan.so; fi
  echo "=== $v"
  for s in B C; do for n in 64 128; do python tools/stamps.py wgrad $s $n 2>&1 | grep -v amdgpu.ids | grep -v "chunk[2-6]"; done; done
done
cp /tmp/base.so graphical_gan_amd/libggan.so
