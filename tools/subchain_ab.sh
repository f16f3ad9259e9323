#!/usr/bin/env bash
# Same-box A/B of the backward's record chains (one vs four per Gaussian) and of the committed library (tools/ab_lib.sh build-base) against the
# working tree: S-1M bench rows, then the garden stand-in with alternating libraries.  Run on the GPU box: bash tools/subchain_ab.sh
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_guarded.py tests/test_gpu_ops.py tests/test_gpu_edge_cases.py tests/test_gpu_cameras.py tests/test_gpu_capi_ctypes.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab_lib.sh run --steps 30 2>&1 | tee gpurun_out/subchain_ab.txt
bash tools/env_ab.sh GSX_BWD_CHAINS "1 4" 2 --steps 30 2>&1 | tee -a gpurun_out/subchain_ab.txt
PKG=gaussian-splatting-cuda_amd
cp $PKG/libgsx.so /tmp/cur.so
timeout 150 python examples/train_garden_standin.py 4000 --json /tmp/g0.json > /dev/null 2>&1
for v in base cur base cur; do
  if [ $v = base ]; then cp tools/variants/libgsx_base.so $PKG/libgsx.so; else cp /tmp/cur.so $PKG/libgsx.so; fi
  timeout 150 python examples/train_garden_standin.py 4000 --json /tmp/g.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('/tmp/g.json')); print('$v', d['iters_per_s'], d['iters_per_s_last_quarter'], d['psnr_after'])" | tee -a gpurun_out/subchain_ab.txt
done
cp /tmp/cur.so $PKG/libgsx.so
for c in 1 4; do
GSX_TEST_SWITCHES=1 GSX_BWD_CHAINS=$c timeout 150 python examples/train_garden_standin.py 4000 --json /tmp/g.json > /dev/null 2>&1
python -c "
import json; d=json.load(open('/tmp/g.json')); print('cur chains=$c', d['iters_per_s'], d['iters_per_s_last_quarter'], d['psnr_after'])" | tee -a gpurun_out/subchain_ab.txt
done
