cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
for v in cur frametr; do
  out=gpurun_out/ab_frame/$v$round; mkdir -p $out
  if [ $v = frametr ]; then export GSX_VARIANT_LIB=$GRAFT_REPO_ROOT/tools/variants/libgsx_frametr.so; else unset GSX_VARIANT_LIB; fi
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python tools/run_fwd_bwd.py 30 all > $out/run.log 2>&1
  python - $out $v$round <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(k in r["Name"] for k in ("frontend_kernel", "gather", "raster_fwd_pair", "raster_bwd_gq")):
        print("%-8s %-60s calls %4s avg %9.1f us" % (sys.argv[2], r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done
