#!/usr/bin/env python
"""Kernel-level look at the bag-of-words path (run under rocprofv3 --kernel-trace --stats): 30 one-shot ComputeBoW calls,
30 SearchByBoW calls, 10 batched ComputeBoW of 64 extracted images, on a 10^6-word synthetic tree."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import orb_slam3_fast_amd as orbx  # noqa: E402
from orb_slam3_fast_amd import synth  # noqa: E402
from orb_slam3_fast_amd.hipmem import DeviceBuffer  # noqa: E402

w, h, nf = 1280, 720, 1500
L1, R1 = synth.stereo_pair(w, h, 300, 1)
cols = synth.make_vocabulary_bfs(10, 6, seed=1)
voc = orbx.ORBVocabulary(10, 6, *cols)
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=64)
_, kc, dc = ex(L1)
_, kr, dr = ex(R1)
fa, fb = voc.transform(dc, 4)[1], voc.transform(dr, 4)[1]
valid = np.ones(len(kc), np.uint8)
for _ in range(30):
    voc.transform(dc, 4)
for _ in range(30):
    orbx.SearchByBoW(fa, kc, dc, valid, fb, kr, dr, -1, 0.7, True)
imgs = DeviceBuffer.from_numpy(np.stack([L1, R1] * 32))
ex.extract_batch_device(imgs.ptr.value, 64, w, h, w, w * h)
ex.sync()
for _ in range(10):
    voc.transform_batch(ex, 4)
    ex.sync()
print("done")
