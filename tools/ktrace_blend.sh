#!/usr/bin/env bash
export GSX_TEST_SWITCHES=1   # the A/B switches below are honoured only under this gate (include/gsx.h: gsx_test_switch)
# kernel-trace stats of the blend ops in isolation: bash tools/ktrace_blend.sh <outdir> <1m|5m>   (GSX_BWD selects the backward variant)
out=${1:-gpurun_out/kt}; scene=${2:-1m}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$out" -o kt --output-format csv -- python tools/blend_ab.py $scene 10 > "$out/run.log" 2>&1
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("raster", "gather", "pack")):
        print("%-70s calls %4s  avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
