"""Per-kernel averages over the launches of the bench's TIMED REGION only, from a rocprofv3 --kernel-trace CSV of
`bench.py --steps K --warmup W --repeats 1 --sustained-steps 0 --no-cpu-baseline --no-camera-batch --no-s5m --no-order-ablation`.
The --stats summary averages every launch of the process: the first launches at a cold clock, the fwd+bwd leg, the per-op pass (every op
bracketed by event records: ~6 us bubbles each, a lower clock).  The order of the steps in that invocation is fixed (bench.py): W + K steps of
the fwd+bwd leg, 1 step that creates the Adam moments, W warm-up steps, the K timed steps, min(K, 8) steps of the per-op pass — the window of
the timed region is found from the launches of the front-end kernel (one per step).
Usage: python tools/trace_region_stats.py <kernel_trace.csv> K W"""
import csv
import sys


def main(path, K, W):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    fe = [r for r in rows if "frontend_kernel" in r["Kernel_Name"]]
    first = (W + K) + 1 + W
    assert len(fe) == first + K + min(K, 8), "unexpected number of steps in the trace: %d front-end launches" % len(fe)
    ws, nxt = fe[first]["s"], fe[first + K]["s"]
    reg = [r for r in rows if ws <= r["s"] < nxt]
    we = max(r["e"] for r in reg)   # (the host reads its events between the region and the per-op pass: the window ends with the region's last kernel)
    agg = {}
    for r in reg:
        a = agg.setdefault(r["Kernel_Name"], [0, 0])
        a[0] += 1
        a[1] += r["e"] - r["s"]
    busy = sum(a[1] for a in agg.values())
    print("timed region: %d steps, window %.1f us per step, kernels %.1f us per step (%.1f %% of the window), %d launches per step\n"
          % (K, (we - ws) / K / 1e3, busy / K / 1e3, 100.0 * busy / (we - ws), len(reg) // K))
    print("| kernel | launches / step | avg us | us / step |\n|---|---|---|---|")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        nm = n if len(n) < 110 else n[:107] + "..."
        print("| `%s` | %.1f | %.1f | %.1f |" % (nm, c / K, t / c / 1e3, t / K / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
