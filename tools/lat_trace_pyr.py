import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
L, R = synth.stereo_pair(1280, 720, 5)
ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=1280, max_height=720, max_batch=2)
ex.set_host_pyramid(True)
for _ in range(40):
    ex.extract_stereo(L, R, bf=63.8, b=0.12)
    ex.host_pyramid(0), ex.host_pyramid(1)
