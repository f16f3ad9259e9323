#!/usr/bin/env bash
# Compile the reference's own CPU math (tests/torch_impl.cpp, unmodified, from where it lies under
# /root/reference) together with oracle/ref_wrapper.cpp into oracle/_ref/libtorchimpl_ref.so.
# No reference source is copied into this repository; outputs go only to oracle/_ref/ (git-ignored).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GSX_REFERENCE_ROOT:-/root/reference}"
if [ ! -f "$REF/tests/torch_impl.cpp" ]; then
  echo "reference not present at $REF — skipping oracle/_ref build" >&2
  exit 0
fi
TP="$(python3 -c 'import torch,os;print(os.path.dirname(torch.__file__))')"
mkdir -p "$HERE/_ref"
OUT="$HERE/_ref/libtorchimpl_ref.so"
if [ -f "$OUT" ] && [ "$OUT" -nt "$HERE/ref_wrapper.cpp" ] && [ "$OUT" -nt "$REF/tests/torch_impl.cpp" ]; then
  echo "up to date: $OUT"; exit 0
fi
g++ -std=c++17 -O2 -fPIC -shared -D_GLIBCXX_USE_CXX11_ABI=1 \
  -I"$REF/tests" -I"$TP/include" -I"$TP/include/torch/csrc/api/include" \
  "$REF/tests/torch_impl.cpp" "$HERE/ref_wrapper.cpp" \
  -L"$TP/lib" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TP/lib" -o "$OUT"
echo "built $OUT"
