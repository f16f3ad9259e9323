#!/usr/bin/env bash
# Kernel trace of the garden stand-in with two builds of libgsx.so (tools/variants/libgsx_base.so = `bash tools/ab_lib.sh build-base`, and the tree's).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; PKG="$ROOT/gaussian-splatting-cuda_amd"
cp "$PKG/libgsx.so" /tmp/libgsx_cur.so
for v in base cur; do
  if [ $v = base ]; then cp "$ROOT/tools/variants/libgsx_base.so" "$PKG/libgsx.so"; else cp /tmp/libgsx_cur.so "$PKG/libgsx.so"; fi
  echo "=== $v"; bash "$ROOT/tools/ktrace_garden.sh" 2>&1 | tail -14
  python -c "import json;d=json.load(open('$ROOT/gpurun_out/ktg/garden.json'));print('it/s',d['iters_per_s'],d['iters_per_s_last_quarter'])"
done
cp /tmp/libgsx_cur.so "$PKG/libgsx.so"
