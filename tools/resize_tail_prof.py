#!/usr/bin/env python3
"""Phase timing inside k_resize_tail (profiling aid, not part of the product).  Build the instrumented library:
  cd orb_slam3_fast_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DRT_PROF -shared \
      -o ../liborbx_prof.so *.hip -ldl -Wl,-rpath,/opt/rocm/lib
then run this on the GPU box: a few workgroups of image 7 print the 10 ns ticks of {tile load, per level: coefficient
loads, barrier, rows, barrier}."""
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
for i in range(3):
    ex.extract_batch_device(d.ptr.value, 64, 1280, 720, 1280, 1280 * 720)
    ex.sync()
    print("----", flush=True)
