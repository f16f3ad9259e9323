#!/usr/bin/env bash
# Same-box comparison of several prebuilt libgsx variants (tools/variants/libgsx_<name>.so: e.g. the blend TU compiled with other LLVM scheduling
# flags) inside the training step: bench.py per-op rows, variants in rotation, two rounds.   GPU box: bash tools/variants_ab.sh base maxilp ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gaussian-splatting-cuda_amd"
cp "$PKG/libgsx.so" /tmp/libgsx_cur.so
for round in 1 2; do
  for v in "$@"; do
    if [ $v = cur ]; then cp /tmp/libgsx_cur.so "$PKG/libgsx.so"; else cp "$ROOT/tools/variants/libgsx_$v.so" "$PKG/libgsx.so"; fi
    (cd "$ROOT" && python bench.py --no-cpu-baseline --no-order-ablation --no-camera-batch --no-s5m --no-fwd-bwd --steps 30 > /tmp/ab.json 2>/tmp/ab.err) || tail -3 /tmp/ab.err
    python - "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json")); k = d["kernels"]
print("%-10s step %.4f  fwd %.4f  bwd %.4f" % (sys.argv[1], d["ms_per_step"], k["rasterize_to_pixels_from_world_3dgs_fwd"]["ms"], k["rasterize_to_pixels_from_world_3dgs_bwd"]["ms"]))
PY
  done
done
cp /tmp/libgsx_cur.so "$PKG/libgsx.so"
