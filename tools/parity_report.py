"""gpurun_out/parity.jsonl (written by tests/helpers.parity_record during `pytest -m gpu`) -> a tracked markdown summary.
Usage: python tools/parity_report.py gpurun_out/parity.jsonl > profiles/parity_r03.md"""
import json
import sys


def fmt(v):
    if isinstance(v, float):
        return "%.3g" % v
    return str(v)


def main(path):
    recs = {}
    for line in open(path):
        line = line.strip()
        if line:
            r = json.loads(line)
            recs[r["test"]] = r   # the last run of a test wins
    print("# Parity numbers of the `-m gpu` tests on the MI355X, as recorded by tests/helpers.parity_record\n")
    print("Two families of records.  (1) `... vs reference kernel`: the HIP operators (and the CPU oracle) against the reference's OWN kernels,")
    print("gsplat/*.cu compiled unmodified for gfx950 (oracle/build_ref_hip.sh) and run on the same GPU, stage by stage on identical inputs")
    print("(tests/test_gpu_reference_hip.py); tolerances asserted: small cases 1e-4 RGB L-inf on every pixel, 1e-3 gradient rel-L2, integers exact;")
    print("full frames: every pixel within one Gaussian's threshold contribution, <= 4e-4 of the pixels beyond 1e-4, gradients 1e-3.  The")
    print("`reference kernel built with --use_fast_math vs reference kernel` records are the reference against ITSELF (its release flags vs IEEE).")
    print("(2) `... vs oracle`: the HIP path against the CPU oracle on the BASELINE configs (tests/test_gpu_fullsize.py): forward 1e-4 L-inf on")
    print("pixels without a threshold-ambiguous decision (window 4e-4), every pixel within max colour / 255 + 1e-4; backward 1e-3 rel-L2;")
    print("projection relative to the float64 evaluation of the same formulas; binning bit-exact.  `wX_` = ambiguity window X.\n")
    for name, r in recs.items():
        print("## " + name + "\n")
        print("| quantity | value |\n|---|---|")
        for k, v in r.items():
            if k != "test":
                print("| %s | %s |" % (k, fmt(v)))
        print()


if __name__ == "__main__":
    main(sys.argv[1])
