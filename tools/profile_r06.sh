#!/usr/bin/env bash
# Round 6's measurement batch on the GPU box (outputs under gpurun_out/r06/; the summaries are copied into profiles/ afterwards):
#   counters of the bench step at S-1M and at S-5M @4K, separate --pmc passes; processed intersections (-DGSX_STATS build);
#   kernel trace of the bench; the bench itself (the driver's command).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06; mkdir -p $out
bash tools/pmc_passes.sh $out/pmc > $out/pmc.log 2>&1; tail -1 $out/pmc.log
python tools/pmc_to_json.py $out/pmc/summary.txt s1m_1080p "profiles/r06_pmc_counters.md (rocprofv3 --pmc, separate passes, tools/pmc_passes.sh; bench step with the cfg2 camera)" 3360791 > $out/pmc_entry_s1m.json
GSX_SCENE=5m PMC_TIMEOUT=240 bash tools/pmc_passes.sh $out/pmc5m > $out/pmc5m.log 2>&1; tail -1 $out/pmc5m.log
python tools/pmc_to_json.py $out/pmc5m/summary.txt s5m_4k "profiles/r06_pmc_counters_s5m.md (rocprofv3 --pmc, separate passes, GSX_SCENE=5m tools/pmc_passes.sh; bench step with the scene's own camera)" 27430411 > $out/pmc_entry_s5m.json
python tools/processed_isects.py 1m --write > $out/processed_1m.json 2> $out/processed_1m.err
python tools/processed_isects.py 5m --write > $out/processed_5m.json 2> $out/processed_5m.err
cp profiles/pmc.json $out/pmc.json
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-camera-batch --no-s5m --no-order-ablation --repeats 1 --sustained-steps 0 > $out/bench_under_rocprof.json 2> $out/kt.err
python - $out <<'PY'
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
PY
python bench.py > $out/bench.json 2> $out/bench.err; echo bench rc=$?
python bench.py > $out/bench2.json 2> $out/bench2.err; echo bench2 rc=$?
ls $out
