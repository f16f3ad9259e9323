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
    dur = collections.defaultdict(dict)   # kernel -> {(file, dispatch): ns} of the pass that collected GRBM_GUI_ACTIVE (the effective clock)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(s in name for s in subs):
                continue
            short = name.split("(")[0].replace("void ", "")[:60]
            agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
            ndisp[(short, r["Counter_Name"])].add(r["Dispatch_Id"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):
                dur[short][(f, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in agg.items():
        print(k)
        for c, val in sorted(v.items()):
            n = max(1, len(ndisp[(k, c)]))
            print("    %-32s %16.0f per dispatch (%d dispatches)" % (c, val / n, n))
        if dur[k]:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; / 8 / the kernel's duration = the clock the chip sustained under this kernel
            # (MI355X_MICROARCH.md "DVFS give-back"; short kernels read high: the counter also runs a little before and after the kernel)
            d = sum(dur[k].values()) / len(dur[k])
            print("    %-32s %16.3f GHz over %.1f us (GRBM_GUI_ACTIVE / 8 / duration, same pass)" % ("effective_clock", v["GRBM_GUI_ACTIVE"] / max(1, len(ndisp[(k, "GRBM_GUI_ACTIVE")])) / 8.0 / d, d / 1e3))


if __name__ == "__main__":
    main()
