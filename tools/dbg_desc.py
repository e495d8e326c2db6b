#!/usr/bin/env python3
"""Debug aid: where do keypoints / descriptors of the HIP path differ from the oracle (by level, by position inside a task)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
from oracle import oracle_py as oracle

w, h, nf, nl, stream = 640, 480, 1000, 8, 12
img = synth.mono_frame(w, h, stream)
ex = orbx.ORBextractor(nf, 1.2, nl, 20, 7, max_width=w, max_height=h)
oe = oracle.OracleExtractor(nf, 1.2, nl, 20, 7)
mono, k, d = ex(img, (0, 0))
omono, ok_, od = oe.extract(img, (0, 0))
print("n", len(k), len(ok_), "mono", mono, omono)
n = min(len(k), len(ok_))
bad_kp = [i for i in range(n) if k[i].tobytes() != ok_[i].tobytes()]
bad_d = [i for i in range(n) if not np.array_equal(d[i], od[i])]
print("bad kp", len(bad_kp), "bad desc", len(bad_d))
# position inside level
lv = ok_["octave"][:n]
starts = {l: int(np.argmax(lv == l)) for l in range(nl)}
for name, bad in (("kp", bad_kp), ("desc", bad_d)):
    pos = [(int(lv[i]), i - starts[int(lv[i])]) for i in bad]
    print(name, "first", pos[:24])
    print(name, "mod4 histogram", np.bincount([p[1] % 4 for p in pos], minlength=4))
for i in bad_kp[:6]:
    print(i, k[i], ok_[i])
for i in bad_d[:3]:
    print(i, "hamming", int(np.unpackbits(d[i] ^ od[i]).sum()))
