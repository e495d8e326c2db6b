#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database, normalised per wave when SQ_WAVES is there.

usage: tools/pmc_insts.py <results.db> [kernel substring]
"""
import sqlite3
import sys


def main(db, pat=""):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like ? "
                     "group by 1, 2", ("%" + pat + "%",)).fetchall()
    by = {}
    for k, n, v in rows:
        by.setdefault(k.split("(")[0].replace("orbx::", ""), {})[n] = v
    for k, d in sorted(by.items()):
        w = d.get("SQ_WAVES")
        print(k, " ".join("%s=%.0f%s" % (n, v, (" (%.1f/wave)" % (v / w)) if w and n != "SQ_WAVES" else "")
                          for n, v in sorted(d.items())))


if __name__ == "__main__":
    main(*sys.argv[1:])
