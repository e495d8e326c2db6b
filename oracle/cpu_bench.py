"""CPU baseline for bench.py: the oracle (single-thread port of the reference's serial semantics) run
frame-parallel over P worker processes, one stereo pair at a time per worker — the `cpu_mt` baseline of
BASELINE.md §2.  TEST / MEASUREMENT INFRASTRUCTURE ONLY.

usage: python -m oracle.cpu_bench <pairs.npy [D,2,H,W] u8> <nfeatures> <bf> <b> <procs> <pairs_per_proc> [--extract-only]
       python -m oracle.cpu_bench --mt <pairs.npy> <nfeatures> <bf> <b> <frames>     (threaded single pipeline, see run_mt)
Prints one JSON object: aggregate pairs/s (all workers start together; wall time of the slowest worker),
single-worker pairs/s, procs.  --extract-only skips ComputeStereoMatches (mono configs: a "pair" is two frames).
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np


def _pin(wid):
    """Pin worker `wid` to ONE core of the process's affinity set (round robin): the baseline's core count is then the number
    of distinct cores in use, and workers do not migrate.  Returns the core or -1 (no sched_setaffinity / pinning failed)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        core = cores[wid % len(cores)]
        os.sched_setaffinity(0, {core})
        return core
    except Exception:
        return -1


def _worker(args):
    path, nf, bf, b, n_pairs, wid, start_at, extract_only = args
    if os.environ.get("ORB_ORACLE_PIN", "1") != "0":
        _pin(wid)
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


def _pair_job(args):
    L, R, nf, bf, b = args
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as oracle
    oL, oR = oracle.OracleExtractor(nf), oracle.OracleExtractor(nf)
    _, kL, dL = oL.extract(L)
    _, kR, dR = oR.extract(R)
    u, dep = oracle.stereo_match(oL, oR, kL, dL, kR, dR, bf, b)
    return kL, dL, kR, dR, u, dep


def oracle_pairs(lefts, rights, nf, bf, b, workers=None):
    """The oracle on many stereo pairs in parallel (checker for the batched parity tests): per pair
    (kpsL, descL, kpsR, descR, uRight, depth).  Spawned workers: safe next to an initialised HIP runtime."""
    import concurrent.futures as cf
    if workers is None:
        workers = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    jobs = [(np.ascontiguousarray(l), np.ascontiguousarray(r), nf, bf, b) for l, r in zip(lefts, rights)]
    workers = max(1, min(workers, len(jobs)))
    if workers == 1:
        return [_pair_job(j) for j in jobs]
    with cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(_pair_job, jobs))


def run(path, nf, bf, b, procs, pairs_per_proc, extract_only=False):
    start_at = time.time() + 3.0 + 0.01 * procs
    with mp.get_context("fork").Pool(procs) as pool:
        times = pool.map(_worker, [(path, nf, bf, b, pairs_per_proc, w, start_at, extract_only) for w in range(procs)])
    return {"procs": procs, "pairs": procs * pairs_per_proc, "wall_s": max(times),
            "pairs_per_s": procs * pairs_per_proc / max(times), "per_worker_pairs_per_s": pairs_per_proc / (sum(times) / len(times))}


def run_mt(path, nf, bf, b, frames):
    """`cpu_mt`: ONE pipeline with the reference's thread structure (2 eye threads x 8 per-level tasks = 16 threads,
    src/Frame.cc:200-203 + src/ORBextractor.cc:764-846) on consecutive stereo frames; timers placed like REGISTER_TIMES
    (src/Frame.cc:196-232): wall time of the both-eye extraction and of ComputeStereoMatches, mean +- std over the frames."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as oracle
    pairs = np.load(path, mmap_mode="r")
    oL, oR = oracle.OracleExtractor(nf), oracle.OracleExtractor(nf)
    ext, ste = [], []
    t0 = time.perf_counter()
    for i in range(frames + 2):
        p = pairs[i % len(pairs)]
        r = oracle.stereo_frame_mt(oL, oR, np.ascontiguousarray(p[0]), np.ascontiguousarray(p[1]), bf, b)
        if i == 1:
            t0 = time.perf_counter()  # two warm-up frames
        if i >= 2:
            ext.append(r[6])
            ste.append(r[7])
    wall = time.perf_counter() - t0
    ext, ste = np.array(ext), np.array(ste)
    tot = ext + ste
    return {"frames": frames, "threads": 2 * oL.nlevels, "wall_s": wall, "pairs_per_s": frames / wall,
            "extract_ms_mean": float(ext.mean()), "extract_ms_std": float(ext.std()),
            "stereo_ms_mean": float(ste.mean()), "stereo_ms_std": float(ste.std()),
            "frame_ms_mean": float(tot.mean()), "frame_ms_std": float(tot.std())}


def digest(path, nf, bf, b):
    """sha256 over everything the oracle returns for the first pair: two builds of the oracle (ORB_ORACLE_LIB) must agree."""
    import hashlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as oracle
    p = np.load(path, mmap_mode="r")[0]
    oL, oR = oracle.OracleExtractor(nf), oracle.OracleExtractor(nf)
    _, kL, dL = oL.extract(np.ascontiguousarray(p[0]))
    _, kR, dR = oR.extract(np.ascontiguousarray(p[1]))
    u, dep = oracle.stereo_match(oL, oR, kL, dL, kR, dR, bf, b)
    h = hashlib.sha256()
    for x in (kL, dL, kR, dR, u, dep):
        h.update(np.ascontiguousarray(x).tobytes())
    return {"digest": h.hexdigest(), "keypoints": int(len(kL) + len(kR))}


if __name__ == "__main__":
    if "--digest" in sys.argv:
        a = [x for x in sys.argv if x != "--digest"]
        print(json.dumps(digest(a[1], int(a[2]), float(a[3]), float(a[4]))))
        sys.exit(0)
    if "--mt" in sys.argv:
        a = [x for x in sys.argv if x != "--mt"]
        print(json.dumps(run_mt(a[1], int(a[2]), float(a[3]), float(a[4]), int(a[5]))))
        sys.exit(0)
    a = sys.argv
    eo = "--extract-only" in a
    a = [x for x in a if x != "--extract-only"]
    res = run(a[1], int(a[2]), float(a[3]), float(a[4]), int(a[5]), int(a[6]), eo)
    if eo:
        res["frames_per_s"] = 2 * res["pairs_per_s"]
    print(json.dumps(res))
