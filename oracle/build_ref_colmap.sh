#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Compiles the reference's own COLMAP reader (src/loader/formats/colmap.cpp, from where it lies, unmodified)
# with the driver oracle/ref_colmap/colmap_ref_tool.cpp into oracle/_ref/colmap_ref_tool.  The reader's logging (spdlog, <format>) and
# image-probe (OpenImageIO) headers are replaced by the stand-ins under oracle/ref_colmap/shim/, which come first on the include path.
# No reference source is copied; output only into oracle/_ref/ (git-ignored).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GSX_REFERENCE_ROOT:-/root/reference}"
SRC="$REF/src/loader/formats/colmap.cpp"
if [ ! -f "$SRC" ]; then echo "reference not present at $REF — skipping colmap_ref_tool" >&2; exit 0; fi
mkdir -p "$HERE/_ref"
OUT="$HERE/_ref/colmap_ref_tool"
if [ -f "$OUT" ] && [ "$OUT" -nt "$HERE/ref_colmap/colmap_ref_tool.cpp" ] && [ "$OUT" -nt "$SRC" ] && [ "$OUT" -nt "$HERE/ref_colmap/shim/core/logger.hpp" ]; then
    echo "up to date: $OUT"; exit 0
fi
TORCH="$(python3 -c 'import torch, os; print(os.path.dirname(torch.__file__))')"
g++ -std=c++20 -O1 -w -D_GLIBCXX_USE_CXX11_ABI=1 -I"$HERE/ref_colmap/shim" -I"$HERE/ref_hip/shim" -I"$REF/src/loader/formats" -I"$REF/src" -I"$REF/include" -I"$REF/gsplat" \
    -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" "$SRC" "$HERE/ref_colmap/colmap_ref_tool.cpp" \
    -L"$TORCH/lib" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TORCH/lib" -o "$OUT"
echo "built $OUT"
