#!/usr/bin/env bash
# kernel-trace stats of the intersection kernels inside the bench step: bash tools/ktrace_isect.sh <outdir> [bench args]   (GSX_WAVE_SORT etc. from the environment)
out=${1:-gpurun_out/kti}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$out" -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-order-ablation --no-camera-batch --no-s5m --no-fwd-bwd "$@" > "$out/bench.json" 2> "$out/err.txt"
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("tile_sort", "bin_", "rank", "rs_", "giant")):
        print("%-64s calls %4s  avg %8.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
