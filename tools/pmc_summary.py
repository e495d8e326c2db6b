#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite) into profiles/pmc_traffic.json.

usage: tools/pmc_summary.py <fetch_results.db> <write_results.db> <out.json> [round-tag]
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Calibration in this access pattern (dword-per-lane
loads, dword stores): k_blur writes exactly P bytes per image and WRITE_SIZE reads 1.02x that, so no correction
factor is applied (the 2x under-count MI355X_MICROARCH.md describes is for 16 B/lane streaming reads).
"""
import json
import sqlite3
import sys


def _kname(k):
    """'void orbx::k_detect<false, 52>(orbx::Geom, ...)' -> 'k_detect'"""
    import re
    return re.sub(r"<.*$", "", re.sub(r"^void\s+", "", k).split("(")[0].replace("orbx::", ""))


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? "
                     "group by kernel_name", (counter,)).fetchall()
    return {_kname(r[0]): (r[1], r[2]) for r in rows}


def main(fdb, wdb, out, tag="r1"):
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    res = {"_note": "HBM-side bytes per launch = (FETCH_SIZE + WRITE_SIZE) KiB * 1024, averaged over dispatches; "
                    "bench config: 64 images (32 stereo pairs) 1280x720 per launch; round " + tag}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fk, wk = f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]
        res[k] = {"fetch_KiB": round(fk, 1), "write_KiB": round(wk, 1), "hbm_bytes_per_launch": int((fk + wk) * 1024),
                  "dispatches": f.get(k, (0, 0))[1]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
