#!/usr/bin/env bash
# kernel trace of the reference's own render call site on the gsx drop-in (tools/dropin_trace.py): bash tools/dropin_profile.sh <outdir>
out=${1:-gpurun_out/dropin}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
for f in gsx fused ref; do python tools/dropin_trace.py 20 $f >> "$out/timing.txt" 2>> "$out/timing.err"; done
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o dt --output-format csv -- python tools/dropin_trace.py 20 gsx > "$out/kt.log" 2>&1
python tools/dropin_trace.py --summarise "$out/kt" 23 > "$out/summary.txt" 2>&1
cat "$out/timing.txt"; head -40 "$out/summary.txt"
