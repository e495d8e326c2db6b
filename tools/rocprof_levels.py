#!/usr/bin/env python3
"""Per-grid-size breakdown of one kernel's launches in a rocprofv3 rocpd database (e.g. k_resize per pyramid level).

usage: tools/rocprof_levels.py <results.db> <kernel substring>
"""
import sqlite3
import sys


def main(db, pat):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else "grid_size_x"
    gy = "grid_y" if "grid_y" in cols else "grid_size_y"
    gz = "grid_z" if "grid_z" in cols else "grid_size_z"
    q = ("select %s, %s, %s, count(*), avg(end - start), min(end - start), max(end - start) from kernels "
         "where name like ? group by 1, 2, 3 order by 1 desc" % (gx, gy, gz))
    for row in c.execute(q, ("%" + pat + "%",)):
        print("grid %6d x %4d x %4d  n=%4d  avg %.1f us  min %.1f  max %.1f" %
              (row[0], row[1], row[2], row[3], row[4] / 1e3, row[5] / 1e3, row[6] / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
