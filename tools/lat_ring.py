#!/usr/bin/env python3
"""orbx_extract_stereo (C ABI, caller arrays) against the number of DISTINCT host frames the calls cycle through (pageable memory:
the runtime pins the pages of every upload; it keeps recently pinned ranges).  usage: python tools/lat_ring.py"""
import ctypes as C, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
w, h, nf = 1280, 720, 1500
bf, b = 0.12 * 532.03, 0.12
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
base = [synth.stereo_pair(w, h, 5 + i) for i in range(8)]
cap = ex.capacity
lap = (C.c_int32 * 2)(0, 0)
n = [C.c_int() for _ in range(4)]
kL, kR = np.empty((cap, 28), np.uint8), np.empty((cap, 28), np.uint8)
dL, dR = np.empty((cap, 32), np.uint8), np.empty((cap, 32), np.uint8)
ur, dp = np.empty(cap, np.float32), np.empty(cap, np.float32)
f = orbx.lib().orbx_extract_stereo
for layout in ("one array [n][2][h][w]", "two arrays left[n][h][w], right[n][h][w]"):
    for nfr in (4, 16, 32, 48, 64, 96, 192):
        if layout.startswith("one"):
            ring = np.empty((nfr, 2, h, w), np.uint8)
            fl = [(ring[i, 0], ring[i, 1]) for i in range(nfr)]
        else:
            left, right = np.empty((nfr, h, w), np.uint8), np.empty((nfr, h, w), np.uint8)
            fl = [(left[i], right[i]) for i in range(nfr)]
        for i, (L, R) in enumerate(fl):
            L[:] = base[i % 8][0]; R[:] = base[i % 8][1]
        args = [(ex._h, L.ctypes.data, R.ctypes.data, w, h, w, w, lap, lap, kL.ctypes.data, dL.ctypes.data, cap, C.byref(n[0]),
                 C.byref(n[1]), kR.ctypes.data, dR.ctypes.data, cap, C.byref(n[2]), C.byref(n[3]), C.c_float(bf), C.c_float(b), ur.ctypes.data,
                 dp.ctypes.data) for L, R in fl]
        gc.collect(); gc.disable()
        for i in range(2 * nfr): f(*args[i % nfr])
        ts = []
        for i in range(400):
            a = args[i % nfr]
            t0 = time.perf_counter(); f(*a); ts.append((time.perf_counter() - t0) * 1e3)
        gc.enable()
        ts = np.array(ts)
        print("%-42s %3d distinct frames (%5.1f MB): mean %.4f p50 %.4f ms" % (layout, nfr, nfr * 2 * w * h / 1e6, ts.mean(), np.percentile(ts, 50)))
