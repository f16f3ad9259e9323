cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03e_extra; mkdir -p $out
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$out/kt5m" -o kt --output-format csv -- python bench.py --scene 5m --steps 10 --warmup 3 --no-cpu-baseline --no-order-ablation > "$out/bench_5m_under_rocprof.json" 2> "$out/kt5m.err"
echo "5m rc=$?"
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$out/ktgarden" -o kt --output-format csv -- python examples/train_garden_standin.py 4000 --json "$out/garden_under_rocprof.json" > /dev/null 2> "$out/ktgarden.err"
echo "garden rc=$?"
python - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for tag in ("kt5m", "ktgarden"):
    f = glob.glob(out + "/" + tag + "/**/*kernel_stats.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    with open(out + "/" + tag + "_kernel_stats.md", "w") as o:
        o.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
            n = r["Name"] if len(r["Name"]) < 110 else r["Name"][:107] + "..."
            o.write("| `%s` | %s | %.3f | %.1f | %.1f | %.1f | %.1f |\n" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print(tag, "total GPU ms", tot / 1e6)
PY
