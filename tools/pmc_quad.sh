#!/usr/bin/env bash
# PMC counters of the two forward blend kernels (tools/fwd_quad_ab.py runs both in one process), one small counter group per rocprofv3 pass.
# Usage (GPU box): bash tools/pmc_quad.sh   -> gpurun_out/pmc_quad/summary_<scene>.txt
export GSX_TEST_SWITCHES=1
out=gpurun_out/pmc_quad
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for sc in heavy 1m; do
mkdir -p "$out/$sc"
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
    i=$((i + 1))
    timeout -k 5 100 rocprofv3 --kernel-trace --pmc $grp -d "$out/$sc/p$i" -o p$i --output-format csv -- python tools/fwd_quad_ab.py $sc 2 p > "$out/$sc/p$i.log" 2>&1
    echo "pass $sc $i: rc=$?"
done
python tools/pmc_summary.py "$out/$sc" raster_fwd > "$out/summary_$sc.txt"
cat "$out/summary_$sc.txt"
done
