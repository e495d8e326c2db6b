#!/usr/bin/env python3
"""Memory-copy + kernel timeline of a rocprofv3 --kernel-trace --memory-copy-trace rocpd database: copies (direction, bytes, GB/s)
and kernels of a window, link busy share per direction.   usage: tools/copy_timeline.py <results.db> [skip_fraction] [window_ms]"""
import sqlite3
import sys


def main(db, skip=0.8, window_ms=6.0):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    mc = [t for t in tabs if t.startswith("rocpd_memory_copy")][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({mc})")]
    print("# copy table columns:", cols)
    rows = c.execute(f"select * from {mc} order by start").fetchall()
    ix = {n: i for i, n in enumerate(cols)}
    cp = [(r[ix["start"]], r[ix["end"]], r[ix["size"]], str(r[ix.get("name_id", ix.get("name", 0))])) for r in rows]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    kr = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    # window: starts at the upload (largest copy size) that lies `skip` of the way through the uploads
    top = max(x[2] for x in cp)
    ups = [x for x in cp if x[2] == top]
    w0 = ups[int(len(ups) * float(skip))][0] - 1000
    w1 = w0 + float(window_ms) * 1e6
    print("# %d uploads of %.2f MB; window starts at upload %d" % (len(ups), top / 1e6, int(len(ups) * float(skip))))
    win = [x for x in cp if x[0] >= w0 and x[1] <= w1]
    big = [x for x in win if x[2] > (1 << 20)]
    for kind, sel in (("copies > 1 MiB", big), ("all copies", win)):
        ev = sorted([(a, 1) for a, b, _, _ in sel] + [(b, -1) for a, b, _, _ in sel])
        busy, depth, last = 0, 0, ev[0][0]
        for t, d in ev:
            if depth > 0:
                busy += t - last
            depth += d
            last = t
        print("%s: %d in %.3f ms, union busy %.1f %%, bytes %.1f MB" % (kind, len(sel), (ev[-1][0] - ev[0][0]) / 1e6,
                                                                        100.0 * busy / (ev[-1][0] - ev[0][0]), sum(x[2] for x in sel) / 1e6))
    print("start_us  dur_us  MB  GB/s  kind")
    for a, b, sz, nm in win[:80]:
        if sz > 100000:
            print("%9.1f %8.1f %7.2f %6.1f  %s" % ((a - w0) / 1e3, (b - a) / 1e3, sz / 1e6, sz / max(1, b - a), nm))
    kw = [(n.split("(")[0].replace("orbx::", "").replace("void ", ""), a, b) for n, a, b in kr if a >= w0 and b <= w1 and "k_" in n]
    print("kernels in the window:", len(kw), " busy-sum %.1f us" % (sum(b - a for _, a, b in kw) / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:])
