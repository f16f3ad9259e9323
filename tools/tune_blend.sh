#!/usr/bin/env bash
# Rebuild libgsx with alternative blend-kernel constants on the GPU box and time the bench step for each.
# Usage: bash tools/tune_blend.sh "-DGSX_FCH=64" "-DGSX_BCH=64" "-DGSX_GM_WAVES=3" ...
for flags in "$@"; do
    GSX_EXTRA_HIPCC_FLAGS="$flags" python gaussian-splatting-cuda_amd/build.py --force > /dev/null 2>&1 || { echo "$flags: build failed"; continue; }
    timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-bwd 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$flags', 'step', d['ms_per_step'], 'fwd', k['rasterize_to_pixels_from_world_3dgs_fwd']['ms'], 'bwd', k['rasterize_to_pixels_from_world_3dgs_bwd']['ms'])"
done
python gaussian-splatting-cuda_amd/build.py --force > /dev/null 2>&1
