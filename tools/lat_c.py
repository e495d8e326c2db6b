#!/usr/bin/env python3
"""The C ABI call orbx_extract_stereo alone (caller output arrays, prebuilt ctypes arguments), 1280x720 N=1500, 96 distinct
frames: mean / p50 / p90 over N calls.  usage: python tools/lat_c.py [calls] ; ORBX_PROF_LIB=liborbx_prof.so selects a build."""
import ctypes as C, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_fast_amd as orbx
if os.environ.get("ORBX_PROF_LIB"):
    orbx.LIB_PATH = os.path.join(os.path.dirname(orbx.__file__), os.environ["ORBX_PROF_LIB"])
from orb_slam3_fast_amd import synth
w, h, nf = 1280, 720, 1500
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
bf, b = 0.12 * 532.03, 0.12
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
base = [synth.stereo_pair(w, h, 5 + i) for i in range(8)]
nfr = 96
ring = np.empty((nfr, 2, h, w), np.uint8)
for i in range(nfr):
    ring[i, 0] = base[i % 8][0]; ring[i, 1] = base[i % 8][1]
cap = ex.capacity
lap = (C.c_int32 * 2)(0, 0)
n = [C.c_int() for _ in range(4)]
kL, kR = np.empty((cap, 28), np.uint8), np.empty((cap, 28), np.uint8)
dL, dR = np.empty((cap, 32), np.uint8), np.empty((cap, 32), np.uint8)
ur, dp = np.empty(cap, np.float32), np.empty(cap, np.float32)
args = [(ex._h, ring[i, 0].ctypes.data, ring[i, 1].ctypes.data, w, h, w, w, lap, lap, kL.ctypes.data, dL.ctypes.data, cap, C.byref(n[0]),
         C.byref(n[1]), kR.ctypes.data, dR.ctypes.data, cap, C.byref(n[2]), C.byref(n[3]), C.c_float(bf), C.c_float(b), ur.ctypes.data,
         dp.ctypes.data) for i in range(nfr)]
f = orbx.lib().orbx_extract_stereo
gc.collect(); gc.disable()
for i in range(30): f(*args[i % nfr])
ts = []
for i in range(calls):
    a = args[i % nfr]
    t0 = time.perf_counter(); f(*a); ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("orbx_extract_stereo (C ABI, caller arrays): mean %.4f p50 %.4f p90 %.4f ms over %d calls  %s" % (ts.mean(), np.percentile(ts, 50), np.percentile(ts, 90), calls, os.environ.get("LAT_TAG", "")))
if os.environ.get("LAT_EXTRA_HANDLES"):
    # the same with other handles (and their streams) alive, idle: bench.py's latency leg runs beside the headline's four handles
    extra = [orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=64) for _ in range(int(os.environ["LAT_EXTRA_HANDLES"]))]
    for e in extra:
        e(base[0][0])
    gc.collect(); gc.disable()
    for i in range(30): f(*args[i % nfr])
    ts = []
    for i in range(calls):
        a = args[i % nfr]
        t0 = time.perf_counter(); f(*a); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print("  with %d idle handles alive: mean %.4f p50 %.4f p90 %.4f ms" % (len(extra), ts.mean(), np.percentile(ts, 50), np.percentile(ts, 90)))
if os.environ.get("LAT_HOST_PYRAMID"):
    ex.set_host_pyramid(True)
    for i in range(30): f(*args[i % nfr])
    ts = []
    for i in range(calls):
        a = args[i % nfr]
        t0 = time.perf_counter(); f(*a); ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    print("  with the host pyramid kept: mean %.4f p50 %.4f p90 %.4f ms" % (ts.mean(), np.percentile(ts, 50), np.percentile(ts, 90)))
    ex.set_host_pyramid(False)
