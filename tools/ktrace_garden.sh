cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ktg; mkdir -p $out
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$out" -o kt --output-format csv -- python examples/train_garden_standin.py 4000 --json "$out/garden.json" > /dev/null 2> "$out/err.txt"
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-60s calls %5s avg %8.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
