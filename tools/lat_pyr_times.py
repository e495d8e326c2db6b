#!/usr/bin/env python3
"""Host-side phases of orbx_extract_stereo with the host pyramid kept (ORBX_LAT_TIMES=1 prints them on stderr)."""
import os, sys
os.environ["ORBX_LAT_TIMES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
w, h, nf = 1280, 720, 1500
ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
frames = [synth.stereo_pair(w, h, 5 + i) for i in range(4)]
ex.set_host_pyramid(True)
for i in range(40):
    ex.extract_stereo(*frames[i % 4], bf=63.8, b=0.12)
