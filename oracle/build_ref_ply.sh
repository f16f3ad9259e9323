#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Compiles the reference's own PLY library (tinyply as vendored by the reference: include/external/tinyply.hpp +
# src/core/tinyply.cpp, from where they lie) with the driver oracle/ref_ply/ply_ref_tool.cpp into oracle/_ref/ply_ref_tool.
# No reference source is copied; output only into oracle/_ref/ (git-ignored).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GSX_REFERENCE_ROOT:-/root/reference}"
if [ ! -f "$REF/src/core/tinyply.cpp" ]; then echo "reference not present at $REF — skipping ply_ref_tool" >&2; exit 0; fi
mkdir -p "$HERE/_ref"
OUT="$HERE/_ref/ply_ref_tool"
if [ -f "$OUT" ] && [ "$OUT" -nt "$HERE/ref_ply/ply_ref_tool.cpp" ] && [ "$OUT" -nt "$REF/src/core/tinyply.cpp" ]; then echo "up to date: $OUT"; exit 0; fi
g++ -std=c++17 -O2 -I"$REF/include" "$REF/src/core/tinyply.cpp" "$HERE/ref_ply/ply_ref_tool.cpp" -o "$OUT"
echo "built $OUT"
