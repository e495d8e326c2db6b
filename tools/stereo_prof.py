#!/usr/bin/env python3
"""Phase timing inside k_stereo_band (profiling aid, not part of the product).  Build the instrumented library:
  cd orb_slam3_fast_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DST_PROF -shared \
      -o ../liborbx_prof.so *.hip -ldl -Wl,-rpath,/opt/rocm/lib
then run this on the GPU box: a few workgroups of pair 5 print the 10 ns ticks of {row-table reads, record staging, barrier,
match, barrier, SAD}."""
import sys, os
sys.path.insert(0, '.')
import orb_slam3_fast_amd as orbx
orbx.LIB_PATH = os.path.join(os.path.dirname(orbx.__file__), "liborbx_prof.so")
import numpy as np
from orb_slam3_fast_amd import synth
from orb_slam3_fast_amd.hipmem import DeviceBuffer
prs = [synth.stereo_pair(1280, 720, stream=i) for i in range(4)]
imgs = np.stack([prs[i % 4][0] for i in range(32)] + [prs[i % 4][1] for i in range(32)])
d = DeviceBuffer.from_numpy(imgs)
ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=1280, max_height=720, max_batch=64)
for i in range(3):
    ex.extract_batch_device(d.ptr.value, 64, 1280, 720, 1280, 1280 * 720)
    orbx.stereo_match_async(ex, ex, 0.12 * 532.03, 0.12, 0, 32, 32)
    ex.sync()
    print("----", flush=True)
