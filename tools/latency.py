#!/usr/bin/env python3
"""Single-frame latency of the drop-in host API (one stereo pair at a time, like Frame's constructor):
two extractor handles called from two threads (src/Frame.cc:200-203) + ComputeStereoMatches, host images in,
host keypoints / descriptors / depths out.   usage: python tools/latency.py [w h nfeatures reps]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
h = int(sys.argv[2]) if len(sys.argv) > 2 else 720
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
L, R = synth.stereo_pair(w, h, 5)
exL = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
exR = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
bf, b = 0.12 * 532.03, 0.12


def frame():
    out = [None, None]

    def run(i, ex, im):
        out[i] = ex(im)

    t0 = time.perf_counter()
    ts = [threading.Thread(target=run, args=(0, exL, L)), threading.Thread(target=run, args=(1, exR, R))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    t1 = time.perf_counter()
    orbx.ComputeStereoMatches(exL, exR, bf, b)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1


import gc
gc.collect()
gc.disable()  # a generation-2 collection inside a millisecond timing window would dominate it
for _ in range(5):
    frame()
ext, st = zip(*[frame() for _ in range(reps)])
ext, st = np.array(ext) * 1e3, np.array(st) * 1e3
t0 = time.perf_counter()
for _ in range(reps):
    exL(L)
seq = (time.perf_counter() - t0) / reps * 1e3
exP = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
frames = [synth.stereo_pair(w, h, 5 + i) for i in range(8)]
for i in range(20):
    exP.extract_stereo(*frames[i % 8], bf=bf, b=b)
ts = []
for i in range(4 * reps):
    Li, Ri = frames[i % 8]
    t0 = time.perf_counter()
    exP.extract_stereo(Li, Ri, bf=bf, b=b)
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
pair = ts.mean()
print("%dx%d N=%d  orbx_extract_stereo (both eyes + ComputeStereoMatches, one batched pipeline, one thread): %.3f ms (%.0f fps)  "
      "[p10 %.3f p50 %.3f p90 %.3f over %d frames]" % (w, h, nf, pair, 1e3 / pair, np.percentile(ts, 10), np.percentile(ts, 50),
                                                        np.percentile(ts, 90), len(ts)))
exP.set_host_pyramid(True)
for i in range(10):
    exP.extract_stereo(*frames[i % 8], bf=bf, b=b)
ts = []
for i in range(2 * reps):
    Li, Ri = frames[i % 8]
    t0 = time.perf_counter()
    exP.extract_stereo(Li, Ri, bf=bf, b=b)
    pl, pr = exP.host_pyramid(0), exP.host_pyramid(1)
    ts.append((time.perf_counter() - t0) * 1e3)
exP.set_host_pyramid(False)
ts = np.array(ts)
print("%dx%d N=%d  the same with the host copy of both pyramids kept current (orbx_set_host_pyramid, the C++ mirror's default): "
      "%.3f ms  [p10 %.3f p50 %.3f p90 %.3f]" % (w, h, nf, ts.mean(), np.percentile(ts, 10), np.percentile(ts, 50), np.percentile(ts, 90)))
from orb_slam3_fast_amd.hipmem import pinned_like
Lp, Rp = pinned_like(L), pinned_like(R)
for _ in range(5):
    exP.extract_stereo(Lp, Rp, bf=bf, b=b)
t0 = time.perf_counter()
for _ in range(reps):
    exP.extract_stereo(Lp, Rp, bf=bf, b=b)
pairp = (time.perf_counter() - t0) / reps * 1e3
print("%dx%d N=%d  the same with the frames in page-locked host memory (hipHostMalloc / registered capture buffers): %.3f ms (%.0f fps)"
      % (w, h, nf, pairp, 1e3 / pairp))
print("%dx%d N=%d  both-eye extraction (2 threads) %.3f +- %.3f ms   stereo match %.3f +- %.3f ms   total %.3f ms (%.0f fps); "
      "one eye alone %.3f ms" % (w, h, nf, ext.mean(), ext.std(), st.mean(), st.std(), (ext + st).mean(),
                                 1e3 / (ext + st).mean(), seq))
