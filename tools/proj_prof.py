#!/usr/bin/env python
"""30 SearchByProjection(F, MapPoints) calls for a kernel trace (rocprofv3 --kernel-trace --stats): how much of the 0.19 ms per
call is kernel time, how much launch gaps / copies / synchronisations."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import orb_slam3_fast_amd as orbx  # noqa: E402
from orb_slam3_fast_amd import synth  # noqa: E402

w, h, nf = 1280, 720, 1500
L0, _ = synth.stereo_pair(w, h, 300, 0)
L1, R1 = synth.stereo_pair(w, h, 300, 1)
eP, eL = (orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h) for _ in range(2))
_, kp, dp = eP(L0)
_, kc, dc = eL(L1)
sf = eL.GetScaleFactors()
rng = np.random.default_rng(2024)
n = len(kp)
mps = np.zeros(n, orbx.MP_DTYPE)
mps["proj_x"], mps["proj_y"] = kp["x"] - 4 + rng.normal(0, 3.0, n), kp["y"] - 2 + rng.normal(0, 3.0, n)
mps["proj_xr"] = mps["proj_x"] - 10
mps["view_cos"], mps["track_depth"] = 0.9985, rng.uniform(1, 80, n)
mps["predicted_level"] = np.clip(kp["octave"] + rng.integers(-1, 2, n), 0, 7)
mps["in_view"], mps["bad"], mps["has_observations"] = 1, 0, 1
mps["desc"] = dp
occ = np.zeros(len(kc), np.uint8)
m = orbx.ORBmatcher(0.8, True)
uR = np.full(len(kc), -1, np.float32)
for _ in range(5):
    m.SearchByProjection(kc, dc, uR, (0.0, 0.0, float(w), float(h)), sf, mps, occ, 3.0, True, 60.0)
t0 = time.perf_counter()
for _ in range(30):
    m.SearchByProjection(kc, dc, uR, (0.0, 0.0, float(w), float(h)), sf, mps, occ, 3.0, True, 60.0)
print("ms per call", (time.perf_counter() - t0) / 30 * 1e3)
