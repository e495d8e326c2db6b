#!/usr/bin/env python3
"""Timeline of ONE single-frame call of the drop-in host API from a rocprofv3 rocpd database collected with
`--kernel-trace --memory-copy-trace --hip-runtime-trace` (tools/frame_trace.sh): every HIP runtime call of the host thread,
every kernel and every copy between the start of the frame's first upload and the return of its synchronisation.

usage: tools/frame_trace.py <results.db> [frame_index_from_end=3]
"""
import sqlite3
import sys


def tables(c):
    return [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]


def cols(c, t):
    return [r[1] for r in c.execute(f"pragma table_info({t})")]


def main(db, back=3):
    back = int(back)
    c = sqlite3.connect(db)
    tabs = tables(c)
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    kern = c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    kern = [(n.split("(")[0].replace("void ", "").replace("orbx::", ""), a, b) for n, a, b in kern]
    mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
    copies = []
    if mc:
        copies = c.execute(f"select start, end, size from {mc[0]} order by start").fetchall()
    # HIP API regions: view `regions` (name, start, end) when present, else rocpd_region + rocpd_string
    api = []
    if "regions" in tabs:
        cc = cols(c, "regions")
        if "name" in cc:
            api = c.execute("select name, start, end from regions order by start").fetchall()
    if not api:
        rg = [t for t in tabs if t.startswith("rocpd_region")]
        st = [t for t in tabs if t.startswith("rocpd_string")]
        if rg and st:
            api = c.execute(f"select s.string, r.start, r.end from {rg[0]} r join {st[0]} s on r.name_id = s.id order by r.start").fetchall()
    # frames are delimited by the result-pack kernel (last kernel of a frame)
    packs = [k for k in kern if "k_result_pack" in k[0] or "k_stereo_filter_pack" in k[0]]
    if len(packs) < back + 1:
        print("not enough frames in the trace")
        return
    end_prev = packs[-back - 1][2]
    end_this = packs[-back][2]
    # the frame begins at the first API call after the previous frame's synchronisation returned
    syncs = [a for a in api if "StreamSynchronize" in a[0] and a[1] < end_prev + 200000 and a[2] >= end_prev]
    t0 = syncs[0][2] if syncs else end_prev
    syncs2 = [a for a in api if "StreamSynchronize" in a[0] and a[2] >= end_this]
    t1 = syncs2[0][2] if syncs2 else end_this
    ev = []
    for n, a, b in api:
        if t0 <= a <= t1:
            ev.append((a, b, "  api  " + n))
    for n, a, b in kern:
        if t0 <= a <= t1:
            ev.append((a, b, "KERNEL " + n))
    for a, b, sz in copies:
        if t0 <= a <= t1:
            ev.append((a, b, "COPY   %d bytes" % sz))
    print("frame: %.1f us from the return of the previous synchronisation to the return of this one" % ((t1 - t0) / 1e3))
    kbusy = sum(b - a for n, a, b in kern if t0 <= a <= t1)
    print("sum of kernel durations %.1f us, sum of copy durations %.1f us" % (kbusy / 1e3, sum(b - a for a, b, s in copies if t0 <= a <= t1) / 1e3))
    print("%10s %9s  what" % ("start us", "dur us"))
    for a, b, s in sorted(ev):
        print("%10.1f %9.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, s))


if __name__ == "__main__":
    main(*sys.argv[1:])
