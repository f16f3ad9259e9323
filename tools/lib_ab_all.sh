#!/usr/bin/env bash
# Same-box A/B of the committed library (tools/ab_lib.sh build-base) against the working tree on the three workloads:
# S-1M bench rows, S-5M @4K bench rows, garden stand-in (alternating libraries).  On the GPU box: bash tools/lib_ab_all.sh
bash tools/ab_lib.sh run --steps 30
PKG=gaussian-splatting-cuda_amd
cp $PKG/libgsx.so /tmp/cur.so
for v in base cur base cur; do
  if [ $v = base ]; then cp tools/variants/libgsx_base.so $PKG/libgsx.so; else cp /tmp/cur.so $PKG/libgsx.so; fi
  python bench.py --scene 5m --steps 10 --warmup 3 --no-cpu-baseline --no-order-ablation --no-camera-batch --no-fwd-bwd 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('5m $v step %.4f bwd %.4f fe %.4f' % (d['ms_per_step'], k['rasterize_to_pixels_from_world_3dgs_bwd']['ms'], k['frontend_fused']['ms']))"
done
for v in base cur base cur; do
  if [ $v = base ]; then cp tools/variants/libgsx_base.so $PKG/libgsx.so; else cp /tmp/cur.so $PKG/libgsx.so; fi
  timeout 150 python examples/train_garden_standin.py 4000 --json /tmp/g.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('/tmp/g.json')); print('standin $v', d['iters_per_s'], d['iters_per_s_last_quarter'], d['psnr_after'])"
done
cp /tmp/cur.so $PKG/libgsx.so
