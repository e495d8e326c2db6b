#!/usr/bin/env python
"""Throughput of the device-resident pre-processing chain on synthetic frames resident in HBM:
  rectify : 64 x 1280x720 raw frames -> cv::remap with two float maps (left / right)        [C3-sized]
  clahe   : 64 x 512x512 frames -> CLAHE(3.0, 8x8)                                          [C4 / TUM-VI-sized]
  gray    : 64 x 1280x720 RGB frames -> cv::cvtColor(RGB2GRAY)                              [colour cameras, Tracking::GrabImage*]
  chain   : CLAHE + remap + extraction of 32 stereo pairs 752x480 through orbx_extract_batch_raw_device
Wall clock around synchronised runs (kernel durations: run under rocprofv3 --kernel-trace --stats)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import orb_slam3_fast_amd as orbx  # noqa: E402
from orb_slam3_fast_amd import synth  # noqa: E402
from orb_slam3_fast_amd.hipmem import DeviceBuffer  # noqa: E402


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters


def main():
    import gc
    gc.collect()
    gc.disable()  # a generation-2 collection inside a 2 ms timing window would dominate it
    only = sys.argv[1] if len(sys.argv) > 1 else None  # 'rectify' | 'clahe' | 'gray': just that stage (for kernel traces)
    out = {}
    B = 64
    w, h = 1280, 720
    L, R = synth.stereo_pair(w, h, 5)
    frames = DeviceBuffer.from_numpy(np.stack([L, R] * (B // 2)))
    ml, mr = synth.rectify_maps(w, h, seed=1), synth.rectify_maps(w, h, seed=2, rot_deg=(-0.3, 0.5, -0.2))
    pp = orbx.Preproc(w, h, maps=(np.stack([ml[0], mr[0]]), np.stack([ml[1], mr[1]])), max_batch=B)
    t = timed(lambda: pp.run_device(frames.ptr.value, B, w, w * h)) if only in (None, 'rectify') else 1.0
    out["rectify_1280x720"] = {"frames_per_s": B / t, "us_per_batch": t * 1e6, "batch": B,
                               "algorithmic_GBps": B * (2 * w * h) / t / 1e9, "with_maps_GBps": (B * 2 * w * h + 2 * 8 * w * h) / t / 1e9}
    w2 = h2 = 512
    f2 = DeviceBuffer.from_numpy(np.stack([synth.mono_frame(w2, h2, i) for i in range(4)] * (B // 4)))
    pc = orbx.Preproc(w2, h2, clahe=(3.0, (8, 8)), max_batch=B)
    t = timed(lambda: pc.run_device(f2.ptr.value, B, w2, w2 * h2)) if only in (None, 'clahe') else 1.0
    tg = 1.0
    if only in (None, 'gray'):
        rgbf = DeviceBuffer.from_numpy(np.stack([np.stack([L, R, L], 2), np.stack([R, L, R], 2)] * (B // 2)))
        pg = orbx.Preproc(w, h, channels=3, rgb=True, max_batch=B)
        tg = timed(lambda: pg.run_device(rgbf.ptr.value, B, 3 * w, 3 * w * h))
    if only:
        return
    out["gray_1280x720x3"] = {"frames_per_s": B / tg, "us_per_batch": tg * 1e6, "batch": B, "algorithmic_GBps": B * (4 * w * h) / tg / 1e9}
    out["clahe_512x512"] = {"frames_per_s": B / t, "us_per_batch": t * 1e6, "batch": B, "algorithmic_GBps": B * (3 * w2 * h2) / t / 1e9}
    w3, h3 = 752, 480
    L3, R3 = synth.stereo_pair(w3, h3, 6)
    f3 = DeviceBuffer.from_numpy(np.stack([L3, R3] * (B // 2)))
    m3l, m3r = synth.rectify_maps(w3, h3, seed=3), synth.rectify_maps(w3, h3, seed=4, rot_deg=(-0.3, 0.5, -0.2))
    p3 = orbx.Preproc(w3, h3, maps=(np.stack([m3l[0], m3r[0]]), np.stack([m3l[1], m3r[1]])), clahe=(3.0, (8, 8)), max_batch=B)
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w3, max_height=h3, max_batch=B)

    def chain():
        ex.extract_batch_raw_device(p3, f3.ptr.value, B, w3, w3 * h3)
        ex.sync()

    def plain():
        ex.extract_batch_device(f3.ptr.value, B, w3, h3, w3, w3 * h3)
        ex.sync()

    tc, tp = timed(chain), timed(plain)
    out["chain_752x480"] = {"frames_per_s_with_preproc": B / tc, "frames_per_s_extract_only": B / tp, "preproc_us_per_batch": (tc - tp) * 1e6}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
