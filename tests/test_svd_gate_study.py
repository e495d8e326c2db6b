"""KannalaBrandt8::TriangulateMatches takes the null vector of its 4x4 system from Eigen::JacobiSVD<Matrix4f> (float);
oracle and device use a one-sided Jacobi in double (Eigen is not in this image).  tools/svd_gate_study.py bounds what the
choice of SVD can change by pushing an independent float32 SVD (LAPACK sgesdd) through the oracle's routine; the full run
(1e5 samples, profiles/r2_svd_gate_study.json) found 0 decision flips.  This is the small always-on version."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_float32_svd_never_flips_a_gate_outside_tolerance(oracle):
    import svd_gate_study
    r = svd_gate_study.study(4000, seed=7)
    assert r["n"] > 1500 and r["both_accept"] > 1000
    assert r["flips_outside_tolerance"] == 0, r["examples_outside_tolerance"]
    assert r["rel_depth_p999"] < 2e-4      # float32-SVD noise on the accepted depths (ill-conditioned tail: max below)
    assert r["rel_depth_max"] < 5e-3


def test_committed_study_result():
    r = json.load(open(os.path.join(ROOT, "profiles", "r2_svd_gate_study.json")))
    assert r["n"] > 50000 and r["flips"] == 0 and r["flips_outside_tolerance"] == 0
