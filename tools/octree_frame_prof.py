#!/usr/bin/env python3
"""Section timing of k_octree's product path (histogram variant) inside a SINGLE stereo frame (orbx_extract_stereo):
build `make -C orb_slam3_fast_amd/csrc prof`, run with ORBX_OCTREE_PROF_LEVEL=<level>."""
import sys, os
sys.path.insert(0, '.')
import orb_slam3_fast_amd as orbx
orbx.LIB_PATH = os.path.join(os.path.dirname(orbx.__file__), os.environ.get("ORBX_PROF_LIB", "liborbx_prof.so"))
from orb_slam3_fast_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
H = int(sys.argv[2]) if len(sys.argv) > 2 else 720
NF = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
L, R = synth.stereo_pair(W, H, 5)
ex = orbx.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2)
for i in range(int(os.environ.get("PROF_FRAMES", "3"))):
    ex.extract_stereo(L, R, bf=63.8, b=0.12)
    print("----", flush=True)
