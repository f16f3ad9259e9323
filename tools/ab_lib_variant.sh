#!/usr/bin/env bash
# Same-box A/B inside the training step (bench.py, per-op events) between the package's libgsx.so and a variant build (tools/build_variant.sh <name> <flags>):
#   bash tools/ab_lib_variant.sh <name> [bench args]      (GPU box; alternates default / variant three times, restores the package's library)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gaussian-splatting-cuda_amd"
name=$1; shift
cp "$PKG/libgsx.so" /tmp/libgsx_cur.so
for round in 1 2 3; do
  for v in default $name; do
    if [ $v = default ]; then cp /tmp/libgsx_cur.so "$PKG/libgsx.so"; else cp "$ROOT/tools/variants/libgsx_$name.so" "$PKG/libgsx.so"; fi
    (cd "$ROOT" && python bench.py --no-cpu-baseline --no-order-ablation --no-camera-batch --no-s5m --no-fwd-bwd --sustained-steps 0 "$@" > /tmp/ab.json 2>/tmp/ab.err) || tail -3 /tmp/ab.err
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["kernels"]
short = {"rasterize_to_pixels_from_world_3dgs_fwd": "fwd", "rasterize_to_pixels_from_world_3dgs_bwd": "bwd", "sh_colors_bwd_adam": "shbwd", "photometric_loss_fwd": "lossf",
         "photometric_loss_bwd": "lossb", "intersect_tile_binned": "isect", "frontend_fused": "fe", "splat_activations_bwd": "actb"}
print("%-8s step %.4f (%s)  " % (sys.argv[1], d["ms_per_step"], " ".join("%.4f" % x for x in d.get("repeats", {}).get("ms_per_step_each", []))) + "  ".join("%s %.4f" % (short.get(n, n[:10]), v["ms"]) for n, v in k.items()))
PY
  done
done
cp /tmp/libgsx_cur.so "$PKG/libgsx.so"
