import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
from orb_slam3_fast_amd.hipmem import DeviceBuffer
B = 64
for (w, h, dw, dh) in ((752, 480, 600, 350), (1280, 720, 640, 360)):
    L, R = synth.stereo_pair(w, h, 6)
    f = DeviceBuffer.from_numpy(np.stack([L, R] * (B // 2)))
    pp = orbx.Preproc(w, h, out_size=(dw, dh), max_batch=B)
    for _ in range(3): pp.run_device(f.ptr.value, B, w, w * h)
    t0 = time.perf_counter()
    for _ in range(20): pp.run_device(f.ptr.value, B, w, w * h)
    t = (time.perf_counter() - t0) / 20
    print("resize %dx%d -> %dx%d x %d: %.1f us wall, %.1f GB/s algorithmic" % (w, h, dw, dh, B, t * 1e6, B * (w * h + dw * dh) / t / 1e9))
