#!/usr/bin/env python3
"""One 1280x720 stereo frame at a time through the C++ drop-in class (csrc/ORBextractor.h: ExtractStereo = both eyes +
ComputeStereoMatches, results in std::vector<cv::KeyPoint> / cv::Mat / std::vector<float>), timed inside the C++ program
(tests/cpp/frame_like latency): what the reference's stereo Frame constructor (src/Frame.cc:196-232) would see.
usage: python tools/lat_cpp.py [calls=600] [distinct_frames=32]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam3_fast_amd import synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_cpp_mirror
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w, h, nf = 1280, 720, 1500
exe = test_cpp_mirror.build_exe(False)
base = [synth.stereo_pair(w, h, 5 + i) for i in range(8)]
ring = np.empty((nfr, 2, h, w), np.uint8)
for i in range(nfr):
    ring[i, 0], ring[i, 1] = base[i % 8]
path = os.path.join(tempfile.gettempdir(), "orbx_lat_frames.raw")
ring.tofile(path)
r = subprocess.run([exe, "latency", str(w), str(h), str(nf), path, str(nfr), str(calls)], capture_output=True, text=True)
print(r.stdout.strip() or r.stderr.strip(), "(rc %d)" % r.returncode)
os.remove(path)
