"""gpurun_out/parity.jsonl (written by tests/helpers.parity_record during `pytest -m gpu`) -> a tracked markdown summary.
Usage: python tools/parity_report.py gpurun_out/parity.jsonl > profiles/parity_r02.md"""
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
    print("# Parity numbers of the `-m gpu` tests (MI355X vs the CPU oracle), as recorded by tests/helpers.parity_record\n")
    print("Tolerances asserted: forward 1e-4 L-inf on pixels without a threshold-ambiguous decision (window 4e-4), every pixel within")
    print("max colour / 255 + 1e-4, every pixel beyond 1e-4 explained by an ambiguous decision (window 1e-3); backward 1e-3 rel-L2;")
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
