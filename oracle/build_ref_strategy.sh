#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Compiles the reference's OWN training host logic — src/training/strategies/mcmc.cpp, strategy_utils.cpp,
# src/training/optimizers/fused_adam.cpp, scheduler.cpp, UNMODIFIED, from where they lie under /root/reference — against THIS repository's
# drop-in (-I compat/gsplat for Ops.h, the reference's own fastgs/optimizer/include/adam_api.h for the Adam operator's declaration; linked with
# gaussian-splatting-cuda_amd/libgsx_gsplat_backend.so + libgsx.so, which define gsplat::relocation / add_noise and
# fast_gs::optimizer::adam_step_wrapper) into the Python extension oracle/_ref/gsplat_ref_strategy.so, so that a -m gpu test
# (tests/test_gpu_reference_strategy.py) can drive gs::training::MCMC next to gsx.strategy.MCMC / gsx.optim.FusedAdam through refine events.
# The classes the sources take come from the reference's own headers under include/ and src/training/, read where they lie; the members of
# gs::SplatData they reference are defined in oracle/ref_callers/core_standins.cpp + oracle/ref_strategy/core_standins_strategy.cpp (the reference
# defines them next to its PLY / SOG / image code: tinyply, TBB, OpenImageIO — not in this image).  Stand-ins for toolchain pieces:
# oracle/ref_hip/shim (glm subset), oracle/ref_callers/shim (<expected>), oracle/ref_colmap/shim (core/logger.hpp: spdlog + <format>),
# oracle/ref_strategy/shim (nlohmann/json_fwd.hpp: a forward declaration).  No reference source is copied; outputs only into oracle/_ref/.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/.." && pwd)"
REF="${GSX_REFERENCE_ROOT:-/root/reference}"
TR="$REF/src/training"
if [ ! -f "$TR/strategies/mcmc.cpp" ]; then echo "reference not present at $REF — skipping gsplat_ref_strategy" >&2; exit 0; fi
PKG="$ROOT/gaussian-splatting-cuda_amd"
OUTDIR="$HERE/_ref"
if [ ! -f "$PKG/libgsx_gsplat_backend.so" ]; then echo "build the package first (python __graft_entry__.py)" >&2; exit 1; fi
TP="$(python3 -c 'import torch,os;print(os.path.dirname(torch.__file__))')"
PYINC="$(python3 -c 'import sysconfig;print(sysconfig.get_paths()["include"])')"
PB11="$(python3 -c 'import pybind11;print(pybind11.get_include())')"
name=gsplat_ref_strategy
obj="$OUTDIR/obj_$name"; out="$OUTDIR/$name.so"
mkdir -p "$obj"
SHIMS="-I$HERE/ref_strategy/shim -I$HERE/ref_colmap/shim -I$HERE/ref_callers/shim -I$HERE/ref_hip/shim"
BASE="-std=c++20 -O2 -fPIC $SHIMS -I$ROOT/compat/gsplat -I$REF/include -I$TR -I$TR/rasterization -I$REF/fastgs/optimizer/include -I$TP/include \
 -I$TP/include/torch/csrc/api/include -I/opt/rocm/include -I$PYINC -I$PB11 -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=1 -DGLM_ENABLE_EXPERIMENTAL -w"
LINK="-L$TP/lib -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -ltorch_python -Wl,-rpath,$TP/lib"
newest=$(ls -t "$TR"/strategies/mcmc.cpp "$TR"/strategies/*.hpp "$TR"/strategies/strategy_utils.cpp "$TR"/optimizers/*.cpp "$TR"/optimizers/*.hpp \
  "$REF"/include/core/splat_data.hpp "$REF"/include/core/parameters.hpp "$HERE"/ref_strategy/*.cpp "$HERE"/ref_callers/core_standins.cpp \
  $(find "$HERE/ref_strategy/shim" "$HERE/ref_callers/shim" "$HERE/ref_colmap/shim" "$HERE/ref_hip/shim" -type f) "$ROOT"/compat/gsplat/*.h \
  "$PKG"/libgsx_gsplat_backend.so "$0" | head -1) || true
if [ -f "$out" ] && [ "$out" -nt "$newest" ]; then echo "up to date: $out"; exit 0; fi
pids=()
( g++ $BASE -c "$TR/strategies/mcmc.cpp" -o "$obj/mcmc.o" ) & pids+=($!)
( g++ $BASE -c "$TR/strategies/strategy_utils.cpp" -o "$obj/strategy_utils.o" ) & pids+=($!)
( g++ $BASE -c "$TR/optimizers/fused_adam.cpp" -o "$obj/fused_adam.o" ) & pids+=($!)
( g++ $BASE -c "$TR/optimizers/scheduler.cpp" -o "$obj/scheduler.o" ) & pids+=($!)
( g++ $BASE -c "$HERE/ref_callers/core_standins.cpp" -o "$obj/core_standins.o" ) & pids+=($!)
( g++ $BASE -c "$HERE/ref_strategy/core_standins_strategy.cpp" -o "$obj/core_standins_strategy.o" ) & pids+=($!)
( g++ $BASE -DTORCH_EXTENSION_NAME=$name -c "$HERE/ref_strategy/strategy_bind.cpp" -o "$obj/strategy_bind.o" ) & pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj"/*.o -L"$PKG" -lgsx_gsplat_backend -lgsx -Wl,-rpath,"$PKG" -Wl,-Bsymbolic $LINK
echo "built $out"
