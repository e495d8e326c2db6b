#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.

usage: tools/rocprof_summary.py gpurun_out/<dir>/<name>_results.db profiles/<name>_kernel_stats.csv
Columns: kernel, calls, total_us, avg_us, pct  (durations converted from ns when needed).
"""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, "%.3f" % tot, "%.3f" % avg, "%.3f" % pct])
    print("wrote %s (%d kernels)" % (out, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
