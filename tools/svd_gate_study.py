#!/usr/bin/env python3
"""How much does the choice of SVD matter in KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:341-432)?

The reference takes the null vector of the 4x4 triangulation system from Eigen::JacobiSVD<Matrix4f> (float, two-sided
Jacobi); Eigen is not in this image, so oracle and device use a one-sided Jacobi in DOUBLE.  This study bounds the
effect with an independent float32 SVD (LAPACK sgesdd, scipy.linalg.lapack -- another algorithm of the reference's
precision class; numpy.linalg.svd would silently compute in double): N random fisheye-stereo matches -- well conditioned (wide parallax) to ill conditioned (parallax at
the 0.9998 gate), exact to several pixels of measurement noise so that the chi-square gates are straddled -- are pushed
through the oracle's routine twice, once with each null vector, and compared:
  * decision flips (accept <-> reject, or a different rejecting gate) and whether the oracle's gated quantity sits within
    the tolerance tests/test_fisheye.py grants (GATE_TOL) -- a flip outside it would be a real disagreement;
  * histogram of the relative depth difference of the matches both accept.

usage: python tools/svd_gate_study.py [N=100000] [out.json]
"""
import json
import os
import sys

import numpy as np
from scipy.linalg import lapack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from orb_slam3_fast_amd import synth  # noqa: E402

GATE_TOL = 1e-3  # tests/test_fisheye.py


def borderline(g):
    cosp, z1, z2, e1, e2 = (float(v) for v in g)
    near = [abs(cosp - 0.9998) < GATE_TOL * 1e-2]
    for z in (z1, z2):
        if not np.isnan(z):
            near.append(abs(z) < 1e-3)
    for e in (e1, e2):
        if not np.isnan(e):
            near.append(abs(e - 1.0) < 10 * GATE_TOL)
    return any(near)


def study(n, seed=20220131):
    rng = np.random.default_rng(seed)
    R12 = np.eye(3)
    t12 = np.array([0.101, 0.002, 0.001])
    rig = O.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM2, R12, t12)
    sig2 = [np.float32(1.2) ** (2 * l) for l in range(8)]
    res = dict(n=0, both_accept=0, both_reject_same_gate=0, flips=0, flips_borderline=0, flips_outside_tolerance=0,
               cond_max=0.0)
    rel = []
    conds = []
    worst = []
    for i in range(n):
        # depth: log-uniform 0.15 .. 120 m (parallax from wide to far beyond the 0.9998 gate), anywhere in the lapping field
        z = float(np.exp(rng.uniform(np.log(0.15), np.log(120.0))))
        X1 = np.array([rng.uniform(-0.8, 0.8) * z, rng.uniform(-0.8, 0.8) * z, z])
        X2 = R12.T @ (X1 - t12)
        if X2[2] <= 0:
            continue
        uv1 = synth.kb8_project_np(synth.TUMVI_CAM1, X1) + rng.normal(0, rng.choice([0.0, 0.3, 1.0, 2.5]), 2)
        uv2 = synth.kb8_project_np(synth.TUMVI_CAM2, X2) + rng.normal(0, rng.choice([0.0, 0.3, 1.0, 2.5]), 2)
        s1, s2 = float(sig2[rng.integers(0, 8)]), float(sig2[rng.integers(0, 8)])
        d0, p0, g0, A = O.kb8_triangulate_ex(rig, uv1, uv2, s1, s2)
        if d0 == -1.0:            # parallax gate: the SVD is never reached
            continue
        _, sv, vt, info = lapack.sgesdd(np.asfortranarray(A, np.float32))
        assert info == 0 and vt.dtype == np.float32
        d1, p1, g1, _ = O.kb8_triangulate_ex(rig, uv1, uv2, s1, s2, xh=vt[3])
        res["n"] += 1
        c = float(sv[0] / max(sv[3], 1e-30))
        conds.append(c)
        a0, a1 = d0 > 1e-4, d1 > 1e-4
        if a0 and a1:
            res["both_accept"] += 1
            rel.append(abs(d1 - d0) / d0)
        elif not a0 and not a1 and d0 == d1:
            res["both_reject_same_gate"] += 1
        else:
            res["flips"] += 1
            if borderline(g0) or borderline(g1):
                res["flips_borderline"] += 1
            else:
                res["flips_outside_tolerance"] += 1
                worst.append(dict(d_double=d0, d_float=d1, gates_double=[float(v) for v in g0], gates_float=[float(v) for v in g1],
                                  cond=c))
    rel = np.array(rel)
    edges = [0, 1e-7, 1e-6, 1e-5, 1e-4, 2e-4, 1e-3, 1e-2, 1.0]
    hist = np.histogram(rel, bins=edges)[0].tolist() if len(rel) else []
    res.update(rel_depth_hist={"edges": edges, "counts": hist}, rel_depth_max=float(rel.max()) if len(rel) else 0.0,
               rel_depth_p999=float(np.percentile(rel, 99.9)) if len(rel) else 0.0,
               cond_percentiles={str(q): float(np.percentile(conds, q)) for q in (50, 90, 99, 100)},
               examples_outside_tolerance=worst[:10])
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    r = study(n)
    print(json.dumps(r, indent=1))
    if len(sys.argv) > 2:
        json.dump(r, open(sys.argv[2], "w"), indent=1)
