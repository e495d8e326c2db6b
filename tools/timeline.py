#!/usr/bin/env python3
"""Steady-state timeline of a rocprofv3 --kernel-trace rocpd database: per-kernel start/end of a window of steps, the share
of wall time with >= 1 kernel running (GPU busy union), idle gaps, and how many kernels run concurrently.

usage: tools/timeline.py <results.db> [skip_fraction=0.5] [window_ms=5]
"""
import sqlite3
import sys


def main(db, skip=0.5, window_ms=5.0):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    rows = [(n.split("(")[0].replace("orbx::", ""), a, b) for n, a, b in rows if "k_" in n]
    t0, t1 = rows[0][1], rows[-1][2]
    w0 = t0 + (t1 - t0) * float(skip)
    w1 = w0 + float(window_ms) * 1e6
    win = [r for r in rows if r[1] >= w0 and r[2] <= w1]
    ev = sorted([(a, 1) for _, a, b in win] + [(b, -1) for _, a, b in win])
    busy, depth, last, hist = 0, 0, ev[0][0], {}
    for t, d in ev:
        if depth > 0:
            busy += t - last
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    span = ev[-1][0] - ev[0][0]
    print("window %.3f ms, %d kernels; busy union %.1f %%; concurrency histogram (share of time with k kernels running): %s"
          % (span / 1e6, len(win), 100.0 * busy / span, {k: round(v / span, 3) for k, v in sorted(hist.items())}))
    per = {}
    for n, a, b in win:
        per.setdefault(n, []).append((b - a) / 1e3)
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("  %-18s n=%3d  avg %7.1f us  total %8.1f us" % (n, len(v), sum(v) / len(v), sum(v)))
    print("  first 40 kernels of the window (start us, duration us):")
    for n, a, b in win[:40]:
        print("    %9.1f %7.1f  %s" % ((a - win[0][1]) / 1e3, (b - a) / 1e3, n))


if __name__ == "__main__":
    main(*sys.argv[1:])
