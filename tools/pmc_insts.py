#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database, normalised per wave when SQ_WAVES is there.

usage: tools/pmc_insts.py <results.db> [kernel substring] [out.json]
With out.json: {kernel: {counter: average per launch, "launches_per_batch": dispatches per bench step}} -- bench.py reads
profiles/pmc_insts.json for roofline.valu (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x duration)).
"""
import json
import sqlite3
import sys


def _kname(k):
    """'void orbx::k_detect<false, 52>(orbx::Geom, ...)' -> 'k_detect'"""
    import re
    k = k.replace("(anonymous namespace)::", "")
    return re.sub(r"<.*$", "", re.sub(r"^void\s+", "", k).split("(")[0].replace("orbx::", ""))


def main(db, pat="", out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like ? "
                     "group by 1, 2", ("%" + pat + "%",)).fetchall()
    by = {}
    for k, n, v in rows:
        by.setdefault(_kname(k), {})[n] = v
    if out:
        n = dict(c.execute("select kernel_name, count(*) from counters_collection where kernel_name like ? and counter_name = "
                           "'SQ_INSTS_VALU' group by 1", ("%" + pat + "%",)).fetchall())
        n = {_kname(k): v for k, v in n.items()}
        steps = [v for k, v in n.items() if k in ("k_detect", "k_describe", "k_octree")]   # one launch per step each (not the clock probe)
        base = min(steps) if steps else (min(n.values()) if n else 1)
        res = {k: dict({c_: round(v, 1) for c_, v in d.items()}, launches_per_batch=max(1, round(n.get(k, base) / base)))
               for k, d in by.items()}
        res["_note"] = ("per-launch averages of one rocprofv3 --pmc pass over the default bench workload (64 images 1280x720 per "
                        "launch); launches_per_batch = dispatches of the kernel per step")
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, d in sorted(by.items()):
        w = d.get("SQ_WAVES")
        print(k, " ".join("%s=%.0f%s" % (n, v, (" (%.1f/wave)" % (v / w)) if w and n != "SQ_WAVES" else "")
                          for n, v in sorted(d.items())))


if __name__ == "__main__":
    main(*sys.argv[1:])
