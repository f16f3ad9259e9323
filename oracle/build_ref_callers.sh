#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Compiles the reference's OWN render call sites — src/training/rasterization/rasterizer.cpp and
# rasterizer_autograd.cpp, UNMODIFIED, from where they lie under /root/reference — twice, into two Python extensions under oracle/_ref/:
#
#   gsplat_ref_callers_gsx.so   against THIS repository's drop-in: -I compat/gsplat (Ops.h / Common.h / Cameras.h / Projection.h), linked with
#                               gaussian-splatting-cuda_amd/libgsx_gsplat_backend.so + libgsx.so — exactly the swap INTEGRATION.md describes
#   gsplat_ref_callers_ref.so   against the reference's gsplat/ headers and its own kernels compiled for gfx950 (the objects of
#                               oracle/build_ref_hip.sh, oracle/_ref/obj_gsplat_ref_hip)
#
# so that a -m gpu test (tests/test_gpu_reference_callers.py) runs gs::training::rasterize forward + backward through both and compares.
# The classes the call sites take (gs::Camera, gs::SplatData, gs::geometry::BoundingBox) come from the reference's own headers under
# include/, read where they lie; the handful of their members the call sites reference are defined in oracle/ref_callers/core_standins.cpp
# (the reference defines them next to its image / PLY / logging code: OpenImageIO, tinyply, spdlog — not in this image).  Stand-ins for
# toolchain pieces: oracle/ref_hip/shim (glm subset, c10/cuda spellings) + oracle/ref_callers/shim (<expected>, glm/gtc/matrix_transform.hpp,
# at::cuda::getStreamFromPool).  No reference source is copied; outputs only into oracle/_ref/ (git-ignored, travels to the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/.." && pwd)"
REF="${GSX_REFERENCE_ROOT:-/root/reference}"
SRC="$REF/src/training/rasterization"
if [ ! -f "$SRC/rasterizer.cpp" ]; then echo "reference not present at $REF — skipping gsplat_ref_callers" >&2; exit 0; fi
PKG="$ROOT/gaussian-splatting-cuda_amd"
OUTDIR="$HERE/_ref"
REFOBJ="$OUTDIR/obj_gsplat_ref_hip"
if [ ! -f "$PKG/libgsx_gsplat_backend.so" ]; then echo "build the package first (python __graft_entry__.py)" >&2; exit 1; fi
if [ ! -f "$REFOBJ/Rasterization.host.o" ]; then bash "$HERE/build_ref_hip.sh"; fi
TP="$(python3 -c 'import torch,os;print(os.path.dirname(torch.__file__))')"
PYINC="$(python3 -c 'import sysconfig;print(sysconfig.get_paths()["include"])')"
PB11="$(python3 -c 'import pybind11;print(pybind11.get_include())')"
SHIMS="-I$HERE/ref_callers/shim -I$HERE/ref_hip/shim"
BASE="-std=c++20 -O2 -fPIC $SHIMS -I$REF/include -I$REF/src/training -I$SRC -I$TP/include -I$TP/include/torch/csrc/api/include -I/opt/rocm/include \
 -I$PYINC -I$PB11 -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=1 -DGLM_ENABLE_EXPERIMENTAL -w"
LINK="-L$TP/lib -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -ltorch_python -Wl,-rpath,$TP/lib"
build_flavour() {   # $1 = gsx | ref, $2 = include dir of the gsplat headers, $3... = what to link the operators from
  local name="gsplat_ref_callers_$1" inc="$2"; shift 2
  local obj="$OUTDIR/obj_$name" out="$OUTDIR/$name.so"
  mkdir -p "$obj"
  local newest
  newest=$(ls -t "$SRC"/rasterizer.cpp "$SRC"/rasterizer_autograd.cpp "$SRC"/*.hpp "$REF"/include/core/camera.hpp "$REF"/include/core/splat_data.hpp \
    "$HERE"/ref_callers/*.cpp $(find "$HERE/ref_callers/shim" "$HERE/ref_hip/shim" -type f) "$inc"/*.h "$PKG"/libgsx_gsplat_backend.so "$REFOBJ"/*.o "$0" | head -1)
  if [ -f "$out" ] && [ "$out" -nt "$newest" ]; then echo "up to date: $out"; return; fi
  local pids=()
  ( g++ $BASE -I"$inc" -c "$SRC/rasterizer.cpp" -o "$obj/rasterizer.o" ) & pids+=($!)
  ( g++ $BASE -I"$inc" -c "$SRC/rasterizer_autograd.cpp" -o "$obj/rasterizer_autograd.o" ) & pids+=($!)
  ( g++ $BASE -I"$inc" -c "$HERE/ref_callers/core_standins.cpp" -o "$obj/core_standins.o" ) & pids+=($!)
  ( g++ $BASE -I"$inc" -DTORCH_EXTENSION_NAME=$name -c "$HERE/ref_callers/callers_bind.cpp" -o "$obj/callers_bind.o" ) & pids+=($!)
  for p in "${pids[@]}"; do wait "$p"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj"/*.o "$@" -Wl,-Bsymbolic $LINK
  echo "built $out"
}
build_flavour gsx "$ROOT/compat/gsplat" -L"$PKG" -lgsx_gsplat_backend -lgsx -Wl,-rpath,"$PKG"
build_flavour ref "$REF/gsplat" $(ls "$REFOBJ"/*.o | grep -v ref_bind.o)
