#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite) into profiles/pmc_traffic.json.

usage: tools/pmc_summary.py <fetch_results.db> <write_results.db> <out.json> [round-tag]
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Calibration (tools/pmc_calib.sh -> profiles/r3_pmc_calibration.json,
profiles/r3_tcc_probe.txt): on gfx950 every L2 -> fabric read request is a 128-byte line (TCC_EA0_RDREQ_32B stays 0 for
4 / 8 / 16 B-per-lane streaming reads AND for k_detect's 8 B/lane row loads) while FETCH_SIZE tallies it at 64 bytes, so
FETCH_SIZE reads exactly 0.5x the bytes fetched in every read shape; WRITE_SIZE (64-byte write requests) is exact (1.000
for 4 / 8 / 16 B-per-lane stores).  Reads are therefore DOUBLED here (MI355X_MICROARCH.md "HBM" prescribes the same x2
for wide streaming reads; the calibration extends it to the narrow shapes); rounds 1-2 reported the raw counter.
"""
FETCH_FACTOR = 2.0   # profiles/r3_pmc_calibration.json: ratio 0.5000 for every read shape
WRITE_FACTOR = 1.0   # ratio 1.0000
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
    res = {"_note": "HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB * 1024 (gfx950: FETCH_SIZE counts 128-byte "
                    "requests at 64 bytes, profiles/r3_pmc_calibration.json), averaged over dispatches; "
                    "bench config: 64 images (32 stereo pairs) 1280x720 per launch; round " + tag}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fk, wk = f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]
        res[k] = {"fetch_KiB_raw": round(fk, 1), "write_KiB_raw": round(wk, 1),
                  "fetch_bytes": int(fk * 1024 * FETCH_FACTOR), "write_bytes": int(wk * 1024 * WRITE_FACTOR),
                  "hbm_bytes_per_launch": int((fk * FETCH_FACTOR + wk * WRITE_FACTOR) * 1024),
                  "dispatches": f.get(k, (0, 0))[1]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:])
