#!/usr/bin/env bash
# The Gaussian-major backward with AoS staging, measured in round 5 and not kept (profiles/r05f_gq_aos_staging.md).
# CPU side:  git apply tools/patches/gq_aos_staging.patch
#            for v in "soa256" "aos256 -DGSX_GQ_AOS" "soa128 -DGSX_GS=128" "aos128 -DGSX_GQ_AOS -DGSX_GS=128"; do bash tools/build_variant.sh $v; done
#            git checkout gaussian-splatting-cuda_amd/csrc/gsx_raster_fast.hip      (profiles/pmc.json is stamped with the hash of the unpatched sources)
# GPU box:   bash tools/aos_experiment.sh       -> backward tests on both AoS builds, the four libraries in rotation in the training step, LDS counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PKG=gaussian-splatting-cuda_amd
cp $PKG/libgsx.so /tmp/keep.so
mkdir -p gpurun_out/aos
for v in aos256 aos128; do cp tools/variants/libgsx_$v.so $PKG/libgsx.so; echo "== tests with $v"; python -m pytest tests/test_gpu_ops.py -q -x -k "backward" 2>&1 | tail -1; done
cp /tmp/keep.so $PKG/libgsx.so
bash tools/variants_ab.sh soa256 aos256 soa128 aos128 2>&1 | grep -v amdgpu.ids
for v in soa128 aos128 aos256; do cp tools/variants/libgsx_$v.so $PKG/libgsx.so; timeout -k 5 60 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/aos/$v -o p --output-format csv -- python tools/run_fwd_bwd.py 2 all > gpurun_out/aos/$v.log 2>&1; echo "== $v"; python tools/pmc_summary.py gpurun_out/aos/$v raster_bwd_gq 2>/dev/null | head -5; done
cp /tmp/keep.so $PKG/libgsx.so
