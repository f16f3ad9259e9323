#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Compiles the reference's own gsplat operators — the .cu kernels and their .cpp hosts, UNMODIFIED,
# from where they lie under /root/reference/gsplat — for gfx950 with hipcc into oracle/_ref/gsplat_ref_hip.so (a Python
# extension: oracle/ref_hip/ref_bind.cpp).  No reference source is copied into this repository; outputs go only to
# oracle/_ref/ (git-ignored, travels to the GPU box with the snapshot).  The reference's own build system is not run.
#
# What stands in for the pieces of the CUDA toolchain the sources name (all under oracle/ref_hip/, builder-written):
#   shim/glm/…               a minimal subset of the glm interface (glm is an un-vendored vcpkg dependency of the reference)
#   shim/cooperative_groups* HIP's cooperative groups + cg::reduce / cg::plus / cg::greater (absent from ROCm 7.2)
#   shim/c10/cuda, ATen/cuda the CUDA spellings of torch's stream / guard / allocator / atomics on a ROCm wheel
#   shim/cub/cub.cuh         hipCUB under the name cub
#   prelude.h                cudaSuccess / cudaFuncSetAttribute → HIP
# Flags: -O3, FMA contraction on (hipcc's default, as nvcc's), IEEE division / sqrt / exp.  The reference's release build adds
# --use_fast_math (gsplat/CMakeLists.txt:75) = ftz + approximate division / sqrt + fast transcendentals (no reassociation);
# GSX_REF_FAST_MATH=1 builds that flavour (clang's equivalents of the three) as gsplat_ref_hip_fast.so.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GSX_REFERENCE_ROOT:-/root/reference}"
if [ ! -f "$REF/gsplat/RasterizeToPixelsFromWorld3DGSFwd.cu" ]; then
  echo "reference not present at $REF — skipping oracle/_ref HIP build" >&2
  exit 0
fi
NAME=gsplat_ref_hip
EXTRA=""
if [ "${GSX_REF_FAST_MATH:-0}" = "1" ]; then NAME=gsplat_ref_hip_fast; EXTRA="-fgpu-flush-denormals-to-zero -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-approx-transcendentals"; fi
SHIM="$HERE/ref_hip"
OUTDIR="$HERE/_ref"
OBJ="$OUTDIR/obj_$NAME"
OUT="$OUTDIR/$NAME.so"
mkdir -p "$OBJ"
newest_dep=$(ls -t "$SHIM"/ref_bind.cpp "$SHIM"/prelude.h $(find "$SHIM/shim" -type f) "$REF"/gsplat/*.cu "$REF"/gsplat/*.cuh "$REF"/gsplat/*.cpp "$REF"/gsplat/*.h "$0" | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest_dep" ]; then echo "up to date: $OUT"; exit 0; fi
TP="$(python3 -c 'import torch,os;print(os.path.dirname(torch.__file__))')"
PYINC="$(python3 -c 'import sysconfig;print(sysconfig.get_paths()["include"])')"
PB11="$(python3 -c 'import pybind11;print(pybind11.get_include())')"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
COMMON="-std=c++20 -fPIC -I$SHIM/shim -I$REF/gsplat -I$TP/include -I$TP/include/torch/csrc/api/include -I/opt/rocm/include \
 -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=1 -DGLM_ENABLE_EXPERIMENTAL -Wno-unknown-pragmas -Wno-deprecated-declarations -w"
pids=()
for f in IntersectTile ProjectionUT3DGSFused RasterizeToPixelsFromWorld3DGSFwd RasterizeToPixelsFromWorld3DGSBwd SphericalHarmonicsCUDA QuatToRotmatCUDA RelocationCUDA; do
  ( "$HIPCC" -x hip --offload-arch=gfx950 -O3 $EXTRA -include "$SHIM/prelude.h" $COMMON -c "$REF/gsplat/$f.cu" -o "$OBJ/$f.o" ) & pids+=($!)
done
for f in Intersect Projection Rasterization SphericalHarmonics QuatToRotmat Relocation; do
  # -DNDEBUG: the reference's Release configuration (CMake adds it).  Rasterization.cpp:65 asserts channels == 3 in front of a dispatch over CDIM = 1, 2, 3, 4, 5, 8, ...
  # (:106-127): with the assert compiled out, as in the binaries the reference ships, its depth render modes (1 and 4 channels, rasterizer.cpp:272-294) reach the kernels
  ( g++ -O2 -DNDEBUG $COMMON -c "$REF/gsplat/$f.cpp" -o "$OBJ/$f.host.o" ) & pids+=($!)
done
( g++ -O2 $COMMON -I"$PYINC" -I"$PB11" -DTORCH_EXTENSION_NAME=$NAME -DPYBIND11_MODULE_NAME=$NAME -c "$SHIM/ref_bind.cpp" -o "$OBJ/ref_bind.o" ) & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o -Wl,-Bsymbolic \
  -L"$TP/lib" -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -ltorch_python -Wl,-rpath,"$TP/lib"
echo "built $OUT"
