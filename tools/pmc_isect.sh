#!/usr/bin/env bash
# PMC counters of the binned intersection on a dense frame (tools/isect_bench.py: hidden scene, 1 M Gaussians, scales x2.2 -> 25 M keys),
# both fills (8-byte keys, ranked), one small counter group per rocprofv3 pass.  Usage (GPU box): bash tools/pmc_isect.sh gpurun_out/pmc_isect
out=${1:-gpurun_out/pmc_isect}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i + 1))
    timeout -k 5 100 rocprofv3 --kernel-trace --pmc $grp -d "$out/p$i" -o p$i --output-format csv -- python tools/isect_bench.py 1000000 2.2 5 > "$out/p$i.log" 2>&1
    echo "pass $i ($grp): rc=$?"
done
python tools/pmc_summary.py "$out" bin_ tile_sort giant rank trampoline > "$out/summary.txt"
cat "$out/summary.txt"
