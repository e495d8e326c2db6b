import numpy as np, sys
sys.path.insert(0, '.')
import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
from oracle import oracle_py as o
mx = 0; mp = 0; dec = 0; tot = 0
for seed in range(20):
    sc = synth.fisheye_stereo_scene(seed)
    args = (sc["kL"], sc["dL"], sc["mono_left"], sc["kR"], sc["dR"], sc["mono_right"])
    rig = o.kb8_rig(sc["cam1"], sc["cam2"], sc["R12"], sc["t12"])
    ora = o.fisheye_stereo_match(*args, rig, sc["level_sigma2"])
    hip = orbx.ComputeStereoFishEyeMatches(*args, rig, sc["level_sigma2"])
    both = (ora[2] >= 0) & (hip[2] >= 0)
    dec += int(((ora[2] >= 0) != (hip[2] >= 0)).sum()); tot += int(both.sum())
    mx = max(mx, float((np.abs(hip[4][both] - ora[4][both]) / ora[4][both]).max()))
    mp = max(mp, float((np.abs(hip[5][both] - ora[5][both]).max(1) / np.linalg.norm(ora[5][both], axis=1)).max()))
print("accepted pairs", tot, "decision flips", dec, "max rel depth err %.3g" % mx, "max rel point err %.3g" % mp)
