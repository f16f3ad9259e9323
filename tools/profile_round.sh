#!/usr/bin/env bash
# One-stop measurement of a round on the GPU box (outputs under gpurun_out/<tag>/; copy the summaries into profiles/):
#   1. bench.py with its default flags (the driver's command)            -> bench.json
#   2. the same bench under rocprofv3 --kernel-trace --stats              -> kernel_stats.md + bench_under_rocprof.json
#   3. PMC counters of the bench step, one small group per pass           -> pmc/summary.txt
#   4. bench.py --scene 5m                                                -> bench_s5m.json
#   5. examples/train_garden_standin.py 4000 (twice)                      -> garden_standin.json
# Usage: bash tools/profile_round.sh r02
tag=${1:-r04}
out=gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
echo "bench rc=$?"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-camera-batch --no-s5m > "$out/bench_under_rocprof.json" 2> "$out/kt.err"
echo "kernel-trace rc=$?"
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
with open(out + "/kernel_stats.md", "w") as o:
    o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        n = r["Name"] if len(r["Name"]) < 110 else r["Name"][:107] + "..."
        o.write("| `%s` | %s | %.3f | %.1f | %.1f | %.1f | %.1f |\n" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print(open(out + "/kernel_stats.md").read()[:3000])
PY
if [ -z "$SKIP_PMC" ]; then   # SKIP_PMC=1 only when the blend sources still hash to profiles/pmc.json's blend_kernel_hash
  bash tools/pmc_passes.sh "$out/pmc" > "$out/pmc.log" 2>&1
  tail -3 "$out/pmc.log"
  # counter-derived figures of the blend ops, stamped with the hash of the blend sources they were measured on (bench.py drops them
  # as "stale" once the sources change); written to profiles/pmc.json on this box -> copied to $out/pmc.json for the merge back
  python tools/pmc_to_json.py "$out/pmc/summary.txt" s1m_1080p "profiles/${tag}_pmc_counters.md (rocprofv3 --pmc, separate passes, tools/pmc_passes.sh; bench step with the cfg2 camera)" 3360791 > "$out/pmc_entry.json" && cp profiles/pmc.json "$out/pmc.json"
fi
python bench.py --scene 5m --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_s5m.json" 2> "$out/bench_s5m.err"
echo "bench 5m rc=$?"
[ -n "$SKIP_GARDEN" ] && exit 0
# 5. the BASELINE configs[2] stand-in, twice (the first run on a fresh box is cold)                       -> garden_standin{_cold,}.json
timeout 150 python examples/train_garden_standin.py 4000 --json "$out/garden_standin_cold.json" > /dev/null 2> "$out/garden.err"
timeout 150 python examples/train_garden_standin.py 4000 --json "$out/garden_standin.json" > /dev/null 2>> "$out/garden.err"
echo "garden rc=$?"; cut -c 200-420 "$out/garden_standin.json"
