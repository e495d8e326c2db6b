#!/usr/bin/env python3
"""Throughput on NATURAL texture at the headline size: 1280x720 stereo pairs cut out of mosaics of the committed natural images
(tests/golden/natural_images.npz: skimage camera / astronaut / the Middlebury motorcycle pair), right eye = left shifted by a
per-pair disparity.  Same loop as tools/kbench.py (device-resident frames, three handles, HIP-event stage table).  The synthetic
streams of bench.py are the denser case (~15 k FAST candidates at level 0 against 4-6 k here).
usage: python tools/bench_natural.py [pairs=32] [steps=40]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd.hipmem import DeviceBuffer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H = 1280, 720
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "natural_images.npz"))
tiles = [z["camera"], z["astronaut"], z["moto_left"], z["moto_right"]]
rng = np.random.default_rng(7)


def mosaic(seed):
    r = np.random.default_rng(seed)
    canvas = np.zeros((H + 64, W + 256), np.uint8)
    y = 0
    while y < canvas.shape[0]:
        x, rowh = 0, 0
        while x < canvas.shape[1]:
            t = tiles[int(r.integers(0, 4))]
            if r.random() < 0.5:
                t = t[:, ::-1]
            h, w = min(t.shape[0], canvas.shape[0] - y), min(t.shape[1], canvas.shape[1] - x)
            canvas[y:y + h, x:x + w] = t[:h, :w]
            x += w
            rowh = max(rowh, h)
        y += rowh
    return canvas


lefts, rights = [], []
for i in range(B):
    c = mosaic(100 + i)
    d = int(rng.integers(8, 96))
    lefts.append(c[32:32 + H, 128:128 + W])
    rights.append(c[32:32 + H, 128 + d:128 + d + W])   # a fronto-parallel scene at disparity d
imgs = np.ascontiguousarray(np.concatenate([np.stack(lefts), np.stack(rights)]))
dbuf = DeviceBuffer.from_numpy(imgs)
exs = [orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * B) for _ in range(3)]
bf, b = 0.12 * 532.03, 0.12
it = [0]


def step():
    e = exs[it[0] % 3]
    it[0] += 1
    e.extract_batch_device(dbuf.ptr.value, 2 * B, W, H, W, W * H)
    orbx.stereo_match_async(e, e, bf, b, 0, B, B)


for _ in range(30):
    step()
for e in exs:
    e.sync()
t0 = time.perf_counter()
for _ in range(K):
    step()
for e in exs:
    e.sync()
dt = time.perf_counter() - t0
ex = exs[0]
ex.profile_enable(True)
ex.profile_collect()
for _ in range(5):
    ex.extract_batch_device(dbuf.ptr.value, 2 * B, W, H, W, W * H)
    orbx.stereo_match_async(ex, ex, bf, b, 0, B, B)
    ex.sync()
prof = ex.profile_collect()
w_, h_, nc, ns = ex.level_stats(0)
print("natural mosaics 1280x720, %d pairs per step, 3 handles: %.0f pairs/s  %.3f ms/step | single handle (us): %s"
      % (B, B * K / dt, 1e3 * dt / K, "  ".join("%s %.0f" % (k[2:], 1e3 * v[0] / max(v[1], 1)) for k, v in prof.items() if v[1])))
print("image 0: FAST candidates per level %s (sum %d), selected keypoints per level %s (sum %d)" % (nc.tolist(), int(nc.sum()), ns.tolist(), int(ns.sum())))
