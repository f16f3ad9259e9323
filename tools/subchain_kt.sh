PKG=gaussian-splatting-cuda_amd
cp $PKG/libgsx.so /tmp/cur.so
for v in base cur; do
  if [ $v = base ]; then cp tools/variants/libgsx_base.so $PKG/libgsx.so; else cp /tmp/cur.so $PKG/libgsx.so; fi
  echo "== $v"; bash tools/ktrace_blend.sh gpurun_out/kt_$v 1m
done
cp /tmp/cur.so $PKG/libgsx.so
