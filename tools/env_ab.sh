#!/usr/bin/env bash
# Same-box A/B of one test switch inside the training step:  bash tools/env_ab.sh VAR "v1 v2 ..." [rounds] [bench args]
# prints ms_per_step and the per-op rows of every run (alternating the values, `rounds` times)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
VAR="$1"; VALS="$2"; ROUNDS="${3:-2}"; shift 3 || true
for round in $(seq 1 "$ROUNDS"); do
  for v in $VALS; do
    (cd "$ROOT" && env GSX_TEST_SWITCHES=1 "$VAR=$v" python bench.py --no-cpu-baseline --no-order-ablation --no-camera-batch --no-s5m --no-fwd-bwd "$@" > /tmp/ab.json 2>/tmp/ab.err) || tail -3 /tmp/ab.err
    python - "$VAR=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["kernels"]
short = {"rasterize_to_pixels_from_world_3dgs_fwd": "fwd", "rasterize_to_pixels_from_world_3dgs_bwd": "bwd", "sh_colors_bwd_adam": "shbwd", "photometric_loss_fwd": "lossf",
         "photometric_loss_bwd": "lossb", "intersect_tile_binned": "isect", "frontend_fused": "fe", "splat_activations_bwd": "actb"}
print("%-18s step %.4f  " % (sys.argv[1], d["ms_per_step"]) + "  ".join("%s %.4f" % (short.get(n, n[:10]), v["ms"]) for n, v in k.items()))
PY
  done
done
