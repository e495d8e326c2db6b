import os, sys, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
w, h, nf = 1280, 720, 1500
bf, b = 0.12 * 532.03, 0.12
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
base = [synth.stereo_pair(w, h, 5 + i) for i in range(8)]
for nfr in (8, 96, 8, 96):
    ring = np.empty((nfr, 2, h, w), np.uint8)
    for i in range(nfr):
        ring[i, 0] = base[i % 8][0]; ring[i, 1] = base[i % 8][1]
    gc.collect(); gc.disable()
    for i in range(20): ex.extract_stereo(ring[i % nfr, 0], ring[i % nfr, 1], bf=bf, b=b)
    ts = []
    for i in range(600):
        L, R = ring[i % nfr, 0], ring[i % nfr, 1]
        t0 = time.perf_counter(); ex.extract_stereo(L, R, bf=bf, b=b); ts.append((time.perf_counter() - t0) * 1e3)
    gc.enable()
    ts = np.array(ts)
    print("%3d distinct frames in one ring: mean %.3f p50 %.3f p90 %.3f ms" % (nfr, ts.mean(), np.percentile(ts, 50), np.percentile(ts, 90)))

# the C ABI call alone (what a C++ caller sees): prebuilt ctypes arguments, (a) NULL output arrays (results stay in the handle's
# page-locked block), (b) caller arrays (the library copies the results out, like ORBextractor::operator()'s output arguments)
import ctypes as C
lib = orbx.lib()
cap = ex.capacity
nfr = 96
ring = np.empty((nfr, 2, h, w), np.uint8)
for i in range(nfr):
    ring[i, 0] = base[i % 8][0]; ring[i, 1] = base[i % 8][1]
lap = (C.c_int32 * 2)(0, 0)
n = [C.c_int() for _ in range(4)]
kL, kR = np.empty((cap, 28), np.uint8), np.empty((cap, 28), np.uint8)
dL, dR = np.empty((cap, 32), np.uint8), np.empty((cap, 32), np.uint8)
ur, dp = np.empty(cap, np.float32), np.empty(cap, np.float32)
for mode in ("NULL output arrays", "caller output arrays", "NULL output arrays", "caller output arrays"):
    outs = mode.startswith("caller")
    args = []
    for i in range(nfr):
        args.append((ex._h, ring[i, 0].ctypes.data, ring[i, 1].ctypes.data, w, h, w, w, lap, lap,
                     kL.ctypes.data if outs else None, dL.ctypes.data if outs else None, cap, C.byref(n[0]), C.byref(n[1]),
                     kR.ctypes.data if outs else None, dR.ctypes.data if outs else None, cap, C.byref(n[2]), C.byref(n[3]),
                     C.c_float(bf), C.c_float(b), ur.ctypes.data if outs else None, dp.ctypes.data if outs else None))
    f = lib.orbx_extract_stereo
    gc.collect(); gc.disable()
    for i in range(20): f(*args[i % nfr])
    ts = []
    for i in range(600):
        a = args[i % nfr]
        t0 = time.perf_counter(); f(*a); ts.append((time.perf_counter() - t0) * 1e3)
    gc.enable()
    ts = np.array(ts)
    print("C ABI call, %s: mean %.3f p50 %.3f p90 %.3f ms" % (mode, ts.mean(), np.percentile(ts, 50), np.percentile(ts, 90)))
