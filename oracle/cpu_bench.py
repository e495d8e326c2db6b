"""CPU baseline for bench.py: the oracle (single-thread port of the reference's serial semantics) run
frame-parallel over P worker processes, one stereo pair at a time per worker — the `cpu_mt` baseline of
BASELINE.md §2.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

usage: python -m oracle.cpu_bench <pairs.npy [D,2,H,W] u8> <nfeatures> <bf> <b> <procs> <pairs_per_proc> [--extract-only]
Prints one JSON object: aggregate pairs/s (all workers start together; wall time of the slowest worker),
single-worker pairs/s, procs.  --extract-only skips ComputeStereoMatches (mono configs: a "pair" is two frames).
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np


def _worker(args):
    path, nf, bf, b, n_pairs, wid, start_at, extract_only = args
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as oracle
    pairs = np.load(path, mmap_mode="r")
    oL, oR = oracle.OracleExtractor(nf), oracle.OracleExtractor(nf)
    L0, R0 = np.ascontiguousarray(pairs[0, 0]), np.ascontiguousarray(pairs[0, 1])
    oL.extract(L0)  # warm the instruction / data paths before the common start
    while time.time() < start_at:
        time.sleep(0.001)
    t0 = time.perf_counter()
    for i in range(n_pairs):
        p = pairs[(wid + i) % len(pairs)]
        L, R = np.ascontiguousarray(p[0]), np.ascontiguousarray(p[1])
        _, kL, dL = oL.extract(L)
        _, kR, dR = oR.extract(R)
        if not extract_only:
            oracle.stereo_match(oL, oR, kL, dL, kR, dR, bf, b)
    return time.perf_counter() - t0


def run(path, nf, bf, b, procs, pairs_per_proc, extract_only=False):
    start_at = time.time() + 3.0 + 0.01 * procs
    with mp.get_context("fork").Pool(procs) as pool:
        times = pool.map(_worker, [(path, nf, bf, b, pairs_per_proc, w, start_at, extract_only) for w in range(procs)])
    return {"procs": procs, "pairs": procs * pairs_per_proc, "wall_s": max(times),
            "pairs_per_s": procs * pairs_per_proc / max(times), "per_worker_pairs_per_s": pairs_per_proc / (sum(times) / len(times))}


if __name__ == "__main__":
    a = sys.argv
    eo = "--extract-only" in a
    a = [x for x in a if x != "--extract-only"]
    res = run(a[1], int(a[2]), float(a[3]), float(a[4]), int(a[5]), int(a[6]), eo)
    if eo:
        res["frames_per_s"] = 2 * res["pairs_per_s"]
    print(json.dumps(res))
