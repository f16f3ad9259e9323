#!/usr/bin/env bash
# Builds tools/variants/libgsx_<name>.so from the working tree with extra -D flags on the blend translation unit (the other objects are the
# package's own):   bash tools/build_variant.sh <name> [-DGSX_...=...]     (CPU side; the A/B scripts under tools/ rotate the libraries on the GPU box)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/gaussian-splatting-cuda_amd/csrc"
name=$1; shift
mkdir -p "$ROOT/tools/variants" /tmp/gsx_var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I"$ROOT/include" "$@" -c "$CSRC/gsx_raster_fast.hip" -o /tmp/gsx_var_$name/gsx_raster_fast.hip.o 2>/dev/null
objs=$(ls "$CSRC"/*.hip.o | grep -v gsx_raster_fast)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/variants/libgsx_$name.so" $objs /tmp/gsx_var_$name/gsx_raster_fast.hip.o
echo "built tools/variants/libgsx_$name.so ($*)"
