#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Compiles the reference's own training kernels — fused SSIM (src/training/kernels/ssim.cu) and fused Adam
# (fastgs/optimizer/src/adam.cu, adam_api.cu), UNMODIFIED, from where they lie — for gfx950 with hipcc into
# oracle/_ref/gsplat_ref_train.so (Python extension: oracle/ref_hip/ref_train_bind.cpp).  Same stand-ins for the CUDA toolchain names as
# oracle/build_ref_hip.sh (oracle/ref_hip/shim/, prelude.h) plus shim/cuda_runtime.h.  No reference source is copied; outputs only into
# oracle/_ref/ (git-ignored, travels to the GPU box).  IEEE flags (the reference compiles these files with --use_fast_math as well; the
# parity tests state their tolerances against this flavour).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GSX_REFERENCE_ROOT:-/root/reference}"
if [ ! -f "$REF/src/training/kernels/ssim.cu" ]; then echo "reference not present at $REF — skipping gsplat_ref_train" >&2; exit 0; fi
NAME=gsplat_ref_train
SHIM="$HERE/ref_hip"
OUTDIR="$HERE/_ref"
OBJ="$OUTDIR/obj_$NAME"
OUT="$OUTDIR/$NAME.so"
mkdir -p "$OBJ"
newest_dep=$(ls -t "$SHIM"/ref_train_bind.cpp "$SHIM"/prelude.h $(find "$SHIM/shim" -type f) "$REF"/src/training/kernels/ssim.cu "$REF"/include/kernels/*ssim* \
  "$REF"/fastgs/optimizer/src/adam*.cu "$REF"/fastgs/optimizer/include/* "$0" | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest_dep" ]; then echo "up to date: $OUT"; exit 0; fi
TP="$(python3 -c 'import torch,os;print(os.path.dirname(torch.__file__))')"
PYINC="$(python3 -c 'import sysconfig;print(sysconfig.get_paths()["include"])')"
PB11="$(python3 -c 'import pybind11;print(pybind11.get_include())')"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="-std=c++20 -fPIC -I$SHIM/shim -I$REF/include -I$REF/fastgs/optimizer/include -I$REF/fastgs/utils -I$TP/include -I$TP/include/torch/csrc/api/include \
 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=1 -Wno-unknown-pragmas -Wno-deprecated-declarations -w"
pids=()
( "$HIPCC" -x hip --offload-arch=gfx950 -O3 -include "$SHIM/prelude.h" $COMMON -c "$REF/src/training/kernels/ssim.cu" -o "$OBJ/ssim.o" ) & pids+=($!)
for f in adam adam_api; do
  ( "$HIPCC" -x hip --offload-arch=gfx950 -O3 -include "$SHIM/prelude.h" $COMMON -c "$REF/fastgs/optimizer/src/$f.cu" -o "$OBJ/$f.o" ) & pids+=($!)
done
( g++ -O2 $COMMON -I"$PYINC" -I"$PB11" -DTORCH_EXTENSION_NAME=$NAME -DPYBIND11_MODULE_NAME=$NAME -c "$SHIM/ref_train_bind.cpp" -o "$OBJ/ref_train_bind.o" ) & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o -Wl,-Bsymbolic \
  -L"$TP/lib" -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -ltorch_python -Wl,-rpath,"$TP/lib"
echo "built $OUT"
