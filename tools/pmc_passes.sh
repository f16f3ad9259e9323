#!/usr/bin/env bash
# PMC counters of the bench step (fused render + loss + backward), one small counter group per rocprofv3 pass (a group the
# hardware cannot collect at once makes rocprofv3 abort and hang: every pass runs under `timeout`).
# Usage (GPU box): bash tools/pmc_passes.sh gpurun_out/pmc
out=${1:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout -k 5 ${PMC_TIMEOUT:-100} rocprofv3 --kernel-trace --pmc $grp -d "$out/p$i" -o p$i --output-format csv -- python tools/run_fwd_bwd.py 2 all > "$out/p$i.log" 2>&1
    echo "pass $i ($grp): rc=$?"
done
python tools/pmc_summary.py "$out" raster_ sh_ bin_ tile_sort projection pack_ gather loss_ activations adam frontend > "$out/summary.txt"
wc -l "$out/summary.txt"
