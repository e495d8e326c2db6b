#!/usr/bin/env python3
"""Join tools/ubench/traffic_calib's known byte counts with two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: tools/pmc_calib.py <known.json> <fetch.db> <write.db> <out.json>
known.json = the list traffic_calib prints ([[label, bytes], ...] in launch order); the databases hold one row per dispatch
in the same order.  Output: per access shape, counter bytes / known bytes -> the correction factors tools/pmc_summary.py
applies (profiles/pmc_calibration.json).
"""
import json
import sqlite3
import sys


def per_dispatch(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)").fetchall()]
    key = "dispatch_id" if "dispatch_id" in cols else "rowid"
    rows = c.execute("select %s, kernel_name, sum(value) from counters_collection where counter_name=? and kernel_name like "
                     "'%%k_%%' group by %s order by %s" % (key, key, key), (counter,)).fetchall()
    return [(r[1].split("(")[0], r[2]) for r in rows]


def main(known, fdb, wdb, out):
    known = json.load(open(known))
    f, w = per_dispatch(fdb, "FETCH_SIZE"), per_dispatch(wdb, "WRITE_SIZE")
    assert len(f) == len(known) == len(w), (len(f), len(w), len(known))
    res = {"_note": "FETCH_SIZE / WRITE_SIZE (KiB per dispatch in rocprofv3's output, x1024 here) against the known byte "
                    "count of each access shape; 'ratio' = counter bytes / known bytes of the direction the kernel "
                    "exercises.  1 GiB ranges (4x the Infinity Cache) unless the label says otherwise.", "shapes": []}
    for (label, nbytes), (kf, vf), (kw, vw) in zip(known, f, w):
        assert label.split("(")[0] in kf, (label, kf)
        rd = "rd" in label
        cnt = (vf if rd else vw) * 1024.0
        res["shapes"].append({"shape": label, "known_bytes": nbytes, "fetch_bytes": int(vf * 1024), "write_bytes": int(vw * 1024),
                              "ratio": round(cnt / nbytes, 4)})
        print("%-24s known %12d  FETCH %12d  WRITE %12d  ratio %.4f" % (label, nbytes, vf * 1024, vw * 1024, cnt / nbytes))
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
