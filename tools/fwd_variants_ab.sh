#!/usr/bin/env bash
# Same-box rotation of prebuilt libgsx variants (tools/build_variant.sh) through tools/fwd_quad_ab.py (forward blend op, S-1M by default):
#   GPU box: bash tools/fwd_variants_ab.sh "<scene> <n> [x]" base p_cur p_w7 ...        GSX_AB_MODES picks the GSX_FWD values per run
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PKG="$ROOT/gaussian-splatting-cuda_amd"
args=$1; shift
cp "$PKG/libgsx.so" /tmp/libgsx_cur.so
for round in 1 2; do
  for v in "$@"; do
    if [ $v = cur ]; then cp /tmp/libgsx_cur.so "$PKG/libgsx.so"; else cp "$ROOT/tools/variants/libgsx_$v.so" "$PKG/libgsx.so"; fi
    echo "== $v"
    (cd "$ROOT" && python tools/fwd_quad_ab.py $args 2>&1 | grep -E "fwd op|identical" | awk '{ if ($0 ~ /identical/) print "   " $0; else print $5, $8 }' | sort | awk '{a[$1]=a[$1] " " $2} END {for (k in a) print "   " k a[k]}')
  done
done
cp /tmp/libgsx_cur.so "$PKG/libgsx.so"
