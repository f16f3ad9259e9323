"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name substring filter), average per dispatch.
Usage: python tools/pmc_summary.py <dir-or-csv> [substr ...]"""
import collections
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    subs = sys.argv[2:] or ["raster_", "sh_", "isect", "projection"]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    ndisp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(s in name for s in subs):
                continue
            short = name.split("(")[0].replace("void ", "")[:60]
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[(short, r["Counter_Name"])].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(k)
        for c, val in sorted(v.items()):
            n = max(1, len(ndisp[(k, c)]))
            print("    %-32s %16.0f per dispatch (%d dispatches)" % (c, val / n, n))


if __name__ == "__main__":
    main()
