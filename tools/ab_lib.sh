#!/usr/bin/env bash
# Same-box A/B of two builds of libgsx.so inside the training step (bench.py per-op events):
#   CPU side:  bash tools/ab_lib.sh build-base     -> tools/variants/libgsx_base.so from the committed (HEAD) kernel sources
#   GPU box:   bash tools/ab_lib.sh run [bench args] -> alternates base / current, prints ms_per_step and the per-op rows
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gaussian-splatting-cuda_amd"
if [ "$1" = "build-base" ]; then
    rm -rf /tmp/gsx_base && mkdir -p /tmp/gsx_base/csrc /tmp/gsx_base/include "$ROOT/tools/variants"
    git -C "$ROOT" archive HEAD gaussian-splatting-cuda_amd/csrc include | tar -x -C /tmp/gsx_base
    cd /tmp/gsx_base/gaussian-splatting-cuda_amd/csrc
    for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c "$f" -o "$f.o" 2>/dev/null & done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/variants/libgsx_base.so" *.hip.o
    echo "built tools/variants/libgsx_base.so from $(git -C "$ROOT" rev-parse --short HEAD)"
    exit 0
fi
shift || true
cp "$PKG/libgsx.so" /tmp/libgsx_cur.so
for round in 1 2 3; do
  for v in base cur; do
    if [ $v = base ]; then cp "$ROOT/tools/variants/libgsx_base.so" "$PKG/libgsx.so"; else cp /tmp/libgsx_cur.so "$PKG/libgsx.so"; fi
    (cd "$ROOT" && python bench.py --no-cpu-baseline --no-order-ablation --no-camera-batch --no-s5m --no-fwd-bwd "$@" > /tmp/ab.json 2>/tmp/ab.err) || tail -3 /tmp/ab.err
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["kernels"]
short = {"rasterize_to_pixels_from_world_3dgs_fwd": "fwd", "rasterize_to_pixels_from_world_3dgs_bwd": "bwd", "sh_colors_bwd_adam": "shbwd", "photometric_loss_fwd": "lossf",
         "photometric_loss_bwd": "lossb", "intersect_tile_binned": "isect", "frontend_fused": "fe", "splat_activations_bwd": "actb"}
print("%-5s step %.4f  " % (sys.argv[1], d["ms_per_step"]) + "  ".join("%s %.4f" % (short.get(n, n[:10]), v["ms"]) for n, v in k.items()))
PY
  done
done
cp /tmp/libgsx_cur.so "$PKG/libgsx.so"
