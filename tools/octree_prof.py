#!/usr/bin/env python3
"""Section timing inside k_octree (profiling aid, not part of the product).

Build the instrumented library first:
  cd orb_slam3_fast_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DOCT_PROF \
      -shared -o ../liborbx_prof.so orbx_kernels.hip orbx_stereo.hip orbx_guided.hip orbx_preproc.hip orbx_bow.hip orbx_api.hip \
      -Wl,-rpath,/opt/rocm/lib
then   ORBX_OCTREE_PROF_LEVEL=0 python tools/octree_prof.py      (pyramid level)
prints, for image 0, the 10 ns ticks between the MK() markers of the per-pass variant (octree_body; this script forces it
with the test hook, the product path is the histogram variant): gather, roots, then per phase-1 pass {count sweep, node
loop, scan, node loop, relabel sweep}, per phase-2 round {sort, rest}, final selection.
"""
import sys, os
sys.path.insert(0, '.')
import orb_slam3_fast_amd as orbx
orbx.LIB_PATH = os.path.join(os.path.dirname(orbx.__file__), os.environ.get("ORBX_PROF_LIB", "liborbx_prof.so"))
import numpy as np
from orb_slam3_fast_amd import synth
L, R = synth.stereo_pair(1280, 720, stream=0)
orbx.lib().orbx_debug_set_octree_global(2)
ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=1280, max_height=720)
for i in range(2):
    ex(L)
    print("----", flush=True)
