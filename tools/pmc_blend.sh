#!/usr/bin/env bash
export GSX_TEST_SWITCHES=1   # the A/B switches below are honoured only under this gate (include/gsx.h: gsx_test_switch)
# PMC counters of the two blend ops in isolation (tools/blend_ab.py), one small counter group per rocprofv3 pass, each under `timeout`.
# Usage (GPU box): [GSX_BWD=pm] bash tools/pmc_blend.sh gpurun_out/pmc_blend [1m|5m]
out=${1:-gpurun_out/pmc_blend}
scene=${2:-1m}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout -k 5 100 rocprofv3 --kernel-trace --pmc $grp -d "$out/p$i" -o p$i --output-format csv -- python tools/blend_ab.py $scene 2 > "$out/p$i.log" 2>&1
    echo "pass $i ($grp): rc=$?"
done
python tools/pmc_summary.py "$out" raster_ pack_ gather > "$out/summary.txt"
cat "$out/summary.txt"
