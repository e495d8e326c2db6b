"""Oracle-independent pins of three OpenCV rules the oracle restates from memory.  None of them is an
OpenCV fixture -- parity stays "partial" -- but each holds a piece of the restatement against something that is NOT the oracle:

* cv::fastAtan2 (src/ORBextractor.cc:98) against numpy.arctan2 in double precision with OpenCV's documented accuracy (0.3 deg);
* the coefficient tables of cv::resize's linear path (src/ORBextractor.cc:1122) against an exact-rational model
  (fractions.Fraction) of the sampling position (d + 1/2) * (src / dst) - 1/2;
* cv::FAST's non-maximum suppression rule and output order (src/ORBextractor.cc:810-826) against a brute-force numpy 3x3 strict
  maximum over a score map that tests/test_pin_skimage.py pins to scikit-image.
"""
import ctypes as C
from fractions import Fraction

import numpy as np
import pytest


def _atan2_n(oracle, y, x):
    y, x = np.ascontiguousarray(y, np.float32), np.ascontiguousarray(x, np.float32)
    out = np.zeros(len(y), np.float32)
    f = oracle.lib().oro_fast_atan2_n
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
    f.restype = None
    f(y.ctypes.data, x.ctypes.data, out.ctypes.data, len(y))
    return out


def test_fast_atan2_within_documented_bound_of_numpy_arctan2(oracle):
    """IC_Angle hands fastAtan2 two INTEGER moments as floats: |m| <= 255 * sum |u| over the radius-15 disc (749 px, sum |u|
    = 4 * sum_{v} ... < 5000): every pair with |m| <= 400 exhaustively, 4e6 random pairs over the whole reachable range, and the
    neighbourhoods of the axes and diagonals at large magnitudes.  OpenCV documents an accuracy of about 0.3 degrees."""
    rng = np.random.default_rng(20220131)
    g = np.arange(-400, 401, dtype=np.int32)
    gy, gx = np.meshgrid(g, g, indexing="ij")
    M = 255 * 5000
    ry, rx = rng.integers(-M, M + 1, 4_000_000), rng.integers(-M, M + 1, 4_000_000)
    big = rng.integers(1, M + 1, 200_000)
    eps = rng.integers(-3, 4, 200_000)
    ys = np.concatenate([gy.ravel(), ry, eps, big, big + eps, -big + eps, big])
    xs = np.concatenate([gx.ravel(), rx, big, eps, big, big, -big + eps])
    got = _atan2_n(oracle, ys, xs).astype(np.float64)
    want = np.degrees(np.arctan2(ys.astype(np.float64), xs.astype(np.float64))) % 360.0
    zero = (ys == 0) & (xs == 0)
    assert (got[zero] == 0).all()                      # fastAtan2(0, 0) = 0
    d = np.abs(got - want)
    d = np.minimum(d, 360.0 - d)[~zero]
    assert d.max() < 0.3, d.max()
    assert ((got >= 0) & (got <= 360.0)).all()
    assert d.mean() < 0.02                             # (the polynomial's typical error is far below its bound)


@pytest.mark.parametrize("s,d", [(1280, 1067), (1067, 889), (889, 741), (741, 617), (617, 514), (514, 429), (429, 357),
                                 (720, 600), (600, 500), (640, 533), (480, 400), (752, 627), (512, 427), (333, 278), (1600, 400),
                                 (100, 250)])
def test_resize_tables_against_exact_rational_positions(oracle, s, d):
    """cv::resize samples destination index i at p = (i + 1/2) * (s / d) - 1/2, takes floor(p) and an 11-bit weight of the
    fraction.  Exact rationals give p; the table may differ from them only by the float rounding of p (relative 2^-24 of a value
    < 2^11: < 2^-12 absolute) and the rounding of the weight to 1/2048."""
    f = oracle.lib().oro_resize_coefs
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f.restype = None
    for clamp in (1, 0):
        ofs, ab = np.zeros(d, np.int32), np.zeros(2 * d, np.int16)
        f(s, d, clamp, ofs.ctypes.data, ab.ctypes.data)
        a0, a1 = ab[0::2].astype(int), ab[1::2].astype(int)
        for i in range(d):
            p = (Fraction(2 * i + 1, 2)) * Fraction(s, d) - Fraction(1, 2)
            fl = p.numerator // p.denominator
            frac = p - fl
            near_int = min(frac, 1 - frac) < Fraction(1, 2048)       # float rounding may land on the other side of an integer
            sx, w1 = int(ofs[i]), int(a1[i])
            if clamp and fl < 0:
                assert (sx, w1, int(a0[i])) == (0, 0, 2048), i
                continue
            if clamp and (fl >= s - 1 or sx >= s - 1):
                assert sx == s - 1 and w1 == 0 and a0[i] == 2048, i
                continue
            if not near_int:
                assert sx == fl, (i, sx, fl)
            pos_table = Fraction(sx) + Fraction(w1, 2048)
            assert abs(pos_table - p) <= Fraction(1, 4096) + Fraction(1, 4096), (i, float(pos_table), float(p))
            assert a0[i] + w1 in (2047, 2048, 2049), i               # two independently rounded weights
            assert 0 <= w1 <= 2048 and 0 <= a0[i] <= 2048


def test_nms_rule_and_output_order_against_bruteforce_strict_maximum(oracle):
    """cv::FAST with nonmaxSuppression keeps a corner iff its score is STRICTLY greater than the scores of its 8 neighbours (a
    non-corner scores 0) and emits the kept corners row by row, left to right -- the order the reference's per-cell lists and
    therefore its quadtree ties depend on.  Score map: the oracle's per-pixel cornerScore, which tests/test_pin_skimage.py holds
    against scikit-image's FAST at every threshold."""
    from orb_slam3_fast_amd import synth
    for stream, th in ((3, 20), (4, 7), (5, 40)):
        img = synth.mono_frame(320, 240, stream)
        img[60:120, 40:200] = np.where(np.random.default_rng(stream).random((60, 160)) < 0.5, 60, 170).astype(np.uint8)   # ties
        S = oracle.fast_score_map(img, th).astype(np.int32)        # 0 = not a corner at th
        h, w = S.shape
        P = np.pad(S, 1)
        nb = np.stack([P[1 + j:1 + j + h, 1 + i:1 + i + w] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i, j) != (0, 0)])
        keep = (S > 0) & (S > nb.max(0))
        ys, xs = np.nonzero(keep)                                  # numpy's nonzero IS row-major order
        want = np.stack([xs, ys, S[ys, xs]], 1)
        got = oracle.fast(img, th, nms=True)
        assert len(want) > 200
        assert np.array_equal(np.asarray(got, np.int64), want.astype(np.int64))
        # equal neighbours suppress each other: among the pre-NMS corners some have an equal-score neighbour and none of them is kept
        eq = (S > 0) & (S == nb.max(0))
        assert eq.any() and not (eq & keep).any()
