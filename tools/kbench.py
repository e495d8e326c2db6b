#!/usr/bin/env python3
"""Per-kernel micro-bench (HIP events inside liborbx) without torch: B synthetic pairs, K steps.
usage: python tools/kbench.py [pairs] [steps] [w] [h]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
from orb_slam3_fast_amd.hipmem import DeviceBuffer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1280
H = int(sys.argv[4]) if len(sys.argv) > 4 else 720
cache = "/tmp/kbench_%dx%d.npy" % (W, H)
if os.path.exists(cache):
    base = np.load(cache)
else:
    prs = [synth.stereo_pair(W, H, stream=i) for i in range(2)]
    base = np.stack([prs[0][0], prs[1][0], prs[0][1], prs[1][1]])
    np.save(cache, base)
imgs = np.concatenate([np.stack([base[i % 2] for i in range(B)]), np.stack([base[2 + i % 2] for i in range(B)])])
d = DeviceBuffer.from_numpy(imgs)
NH = int(os.environ.get("KB_HANDLES", "1"))
exs = [orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * B) for _ in range(NH)]
ex = exs[0]
bf, b = 0.12 * 532.03, 0.12
_it = [0]


def step():
    e = exs[_it[0] % NH]
    _it[0] += 1
    e.extract_batch_device(d.ptr.value, 2 * B, W, H, W, W * H)
    orbx.stereo_match_async(e, e, bf, b, 0, B, B)


for _ in range(2 * NH):
    step()
for e in exs:
    e.sync()
    e.profile_enable(os.environ.get("KB_NOPROF") is None)
    e.profile_collect()
t0 = time.perf_counter()
for _ in range(K):
    step()
for e in exs:
    e.sync()
dt = time.perf_counter() - t0
prof = ex.profile_collect()
tag = os.environ.get("KB_TAG", "")
print("%s pairs/s %.0f  ms/step %.3f | " % (tag, B * K / dt, 1e3 * dt / K) +
      "  ".join("%s %.1f" % (k[2:], 1e3 * v[0] / max(v[1], 1)) for k, v in prof.items() if v[1]))
