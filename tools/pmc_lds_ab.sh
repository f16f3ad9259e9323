#!/usr/bin/env bash
# LDS counters of the bench step's kernels with the package's libgsx.so and with a variant build: bash tools/pmc_lds_ab.sh <variant name> [kernel substring]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
name=$1; pat=${2:-raster_fwd}
out=gpurun_out/pmc_lds_$name; mkdir -p $out
for flav in default $name; do
  if [ $flav = default ]; then unset GSX_VARIANT_LIB; else export GSX_VARIANT_LIB=$GRAFT_REPO_ROOT/tools/variants/libgsx_$name.so; fi
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/$flav -o p --output-format csv -- python tools/run_fwd_bwd.py 2 all > $out/$flav.log 2>&1
  echo "$flav rc=$?"
  python - $out/$flav "$pat" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = {}
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = agg.setdefault((r["Kernel_Name"][:60], r["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print("   %-62s %-22s %12.0f per dispatch (%d)" % (k, c, v / n, n))
PY
done
