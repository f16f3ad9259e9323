"""Summarise a rocprofv3 rocpd SQLite database (…_results.db) into a per-kernel stats table
(calls, total / average / min / max duration) — the same content as `--stats` CSV output.
Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/xxx_kernel_stats.md"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        f"max(vgpr_count) if_v, max(lds_size) from kernels group by {name_col} order by sum(end-start) desc"
        if "vgpr_count" in cols else
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), 0, 0 "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        nm = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
        print(f"| `{nm}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | {100*r[2]/total:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1])
