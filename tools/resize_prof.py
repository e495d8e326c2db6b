#!/usr/bin/env python3
"""Phase timing inside k_resize (profiling aid, not part of the product).  Build the instrumented library with
-DRS_PROF (same command as tools/octree_prof.py, output ../liborbx_prof.so), then run this script on the GPU box: a few
blocks of image 7 print the 10 ns ticks spent in {footprint load, barrier, horizontal pass, barrier, vertical pass}.
"""
import sys, os
sys.path.insert(0, '.')
import orb_slam3_fast_amd as orbx
orbx.LIB_PATH = os.path.join(os.path.dirname(orbx.__file__), "liborbx_prof.so")
import numpy as np
from orb_slam3_fast_amd import synth
from orb_slam3_fast_amd.hipmem import DeviceBuffer
L, R = synth.stereo_pair(1280, 720, stream=0)
imgs = np.stack([L, R] * 32)
d = DeviceBuffer.from_numpy(imgs)
ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=1280, max_height=720, max_batch=64)
for i in range(2):
    ex.extract_batch_device(d.ptr.value, 64, 1280, 720, 1280, 1280 * 720)
    ex.sync()
    print("----", flush=True)
