#!/usr/bin/env bash
# A/B of the pair forward's alive-block mask (round 6; as measured: the mask was the default of the working tree then, -DGSX_PAIR_NO_ALIVE_MASK the variant —
# now it is -DGSX_PAIR_ALIVE_MASK against the default): timing (alternating libraries) and step counters on S-1M, S-5M @4K and a saturated frame.
# Variants: bash tools/build_variant.sh <name> <flags> on the CPU side (noalive, stats, stats_noalive).
cd "$GRAFT_REPO_ROOT"
V=$GRAFT_REPO_ROOT/tools/variants
for sc in 1m 5m dense; do
  for rep in 1 2; do
    for lib in cur noalive; do
      if [ $lib = cur ]; then unset GSX_VARIANT_LIB; else export GSX_VARIANT_LIB=$V/libgsx_$lib.so; fi
      echo "$sc $lib: $(python tools/blend_ab.py $sc 30 2>&1 | tail -1)"
    done
  done
done
unset GSX_VARIANT_LIB
for sc in 1m 5m; do
  echo "stats $sc alive-mask:    $(GSX_STATS_LIB=$V/libgsx_stats.so python tools/processed_isects.py $sc 2>&1 | tail -1 | cut -c1-330)"
  echo "stats $sc no alive-mask: $(GSX_STATS_LIB=$V/libgsx_stats_noalive.so python tools/processed_isects.py $sc 2>&1 | tail -1 | cut -c1-330)"
done
