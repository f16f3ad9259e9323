#!/usr/bin/env bash
# The garden stand-in with two builds of libgsx.so, alternating (tools/variants/libgsx_base.so vs the tree's), no profiler.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; PKG="$ROOT/gaussian-splatting-cuda_amd"
cp "$PKG/libgsx.so" /tmp/libgsx_cur.so
for round in 1 2 3; do
  for v in base cur; do
    if [ $v = base ]; then cp "$ROOT/tools/variants/libgsx_base.so" "$PKG/libgsx.so"; else cp /tmp/libgsx_cur.so "$PKG/libgsx.so"; fi
    (cd "$ROOT" && timeout 150 python examples/train_garden_standin.py 4000 --json /tmp/g.json > /dev/null 2>/tmp/g.err) || tail -2 /tmp/g.err
    python -c "import json;d=json.load(open('/tmp/g.json'));print('$v','it/s',d['iters_per_s'],'last quarter',d['iters_per_s_last_quarter'])"
  done
done
cp /tmp/libgsx_cur.so "$PKG/libgsx.so"
