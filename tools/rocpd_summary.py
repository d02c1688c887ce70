"""Summarise rocprofv3's rocpd databases (ROCm 7.2 default output) into the CSVs kept under profiles/.

  python tools/rocpd_summary.py stats  <trace_results.db> <out.csv>      per-kernel calls / total / avg / min / max (ns)
  python tools/rocpd_summary.py pmc    <pmc_results.db>   <out.csv>      per-kernel sum and per-launch mean of every counter
"""
import csv
import sqlite3
import sys


def cols(c, table):
    return [r[1] for r in c.execute(f"pragma table_info({table})")]


def stats(db, out):
    c = sqlite3.connect(db)
    k = cols(c, "kernels")
    name = "name" if "name" in k else "kernel_name"
    dur = "duration" if "duration" in k else '("end" - start)'
    rows = c.execute(f"select {name}, count(*), sum({dur}), avg({dur}), min({dur}), max({dur}) from kernels group by {name} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100.0 * r[2] / tot:.2f}", r[4], r[5]])
    return rows


def pmc(db, out):
    c = sqlite3.connect(db)
    k = cols(c, "counters_collection")
    name = "kernel_name" if "kernel_name" in k else "name"
    rows = c.execute(f"select {name}, counter_name, count(distinct dispatch_id), sum(value) from counters_collection group by 1, 2 order by 4 desc").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Dispatches", "Sum", "MeanPerDispatch"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], r[3] / max(r[2], 1)])
    return rows


if __name__ == "__main__":
    mode, db, out = sys.argv[1:4]
    for r in (stats if mode == "stats" else pmc)(db, out):
        print(r)
