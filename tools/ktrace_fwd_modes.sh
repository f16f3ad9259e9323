#!/usr/bin/env bash
# In-step durations of the forward blend kernels on ONE box: the bench's training step under rocprofv3 --kernel-trace with each forward kernel forced
# (GSX_FWD, read per launch), alternating.   GPU box: bash tools/ktrace_fwd_modes.sh [modes...]      -> average us of the raster_fwd_* / raster_bwd_gq kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GSX_TEST_SWITCHES=1
modes=${@:-quad pair quad pair}
i=0
for m in $modes; do
  i=$((i + 1))
  out=gpurun_out/ktfwd/$i
  mkdir -p $out
  GSX_FWD=$m timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-camera-batch --no-s5m --no-order-ablation --no-fwd-bwd > $out/bench.json 2> $out/err
  python - $m $out <<'PY'
import csv, glob, json, sys
m, out = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0])))
pick = lambda sub: next((r for r in rows if sub in r["Name"]), None)
f = pick("raster_fwd_"); b = pick("raster_bwd_gq")
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("GSX_FWD=%-5s %-28s avg %.1f us (%s calls)   raster_bwd_gq avg %.1f us   step %.4f ms" % (m, f["Name"].split("(")[0][-28:], float(f["AverageNs"]) / 1e3, f["Calls"], float(b["AverageNs"]) / 1e3, d["ms_per_step"]))
PY
done
