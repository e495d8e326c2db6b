#!/usr/bin/env python3
"""Section timing of k_octree's histogram variant (the product path); build liborbx_prof.so with -DOCT_PROF (see
tools/octree_prof.py) and run with ORBX_OCTREE_PROF_LEVEL=<level> (level 0 cannot be selected: use -1 via the env
ORBX_OCTREE_PROF_LEVEL0=1)."""
import sys, os
sys.path.insert(0, '.')
import orb_slam3_fast_amd as orbx
orbx.LIB_PATH = os.path.join(os.path.dirname(orbx.__file__), os.environ.get("ORBX_PROF_LIB", "liborbx_prof.so"))
import numpy as np
from orb_slam3_fast_amd import synth
from orb_slam3_fast_amd.hipmem import DeviceBuffer
prs = [synth.stereo_pair(1280, 720, stream=i) for i in range(4)]
imgs = np.stack([prs[i % 4][0] for i in range(32)] + [prs[i % 4][1] for i in range(32)])
d = DeviceBuffer.from_numpy(imgs)
ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=1280, max_height=720, max_batch=64)
for i in range(3):
    ex.extract_batch_device(d.ptr.value, 64, 1280, 720, 1280, 1280 * 720)
    ex.sync()
    print("----", flush=True)
