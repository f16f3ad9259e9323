#!/usr/bin/env bash
# same-box A/B of the blend ops between the package's libgsx.so and a variant build (tools/build_variant.sh <name> <flags>): bash tools/ab_variant.sh <name> [1m|5m] [rounds]
cd "$GRAFT_REPO_ROOT"
name=$1; sc=${2:-1m}; rounds=${3:-3}
for r in $(seq $rounds); do
  unset GSX_VARIANT_LIB
  echo "default : $(python tools/blend_ab.py $sc 30 2>&1 | tail -1 | cut -c1-110)"
  export GSX_VARIANT_LIB=$GRAFT_REPO_ROOT/tools/variants/libgsx_$name.so
  echo "$name : $(python tools/blend_ab.py $sc 30 2>&1 | tail -1 | cut -c1-110)"
done
