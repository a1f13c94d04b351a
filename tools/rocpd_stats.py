#!/usr/bin/env python
"""Per-kernel summary (calls, total / mean / min / max duration) of a rocprofv3 rocpd database -> CSV in the column layout of
rocprofv3's own kernel_stats.csv.    python tools/rocpd_stats.py gpurun_out/prof/x_results.db profiles/out.csv"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100.0 * r[2] / tot:.2f}", r[4], r[5]])
    for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
        print(f"{r[0][:70]:70s} {r[1]:6d} avg {r[3] / 1e3:9.1f} us  {100.0 * r[2] / tot:5.1f} %")
    print(f"total {tot / 1e6:.2f} ms over all launches")


if __name__ == "__main__":
    main()
