"""Oracle-independent pins of three OpenCV rules the oracle restates from memory.  None of them is an
OpenCV fixture -- parity stays "partial" -- but each holds a piece of the restatement against something that is NOT the oracle:

* cv::fastAtan2 (src/ORBextractor.cc:98) against numpy.arctan2 in double precision with OpenCV's documented accuracy (0.3 deg);
* the coefficient tables of cv::resize's linear path (src/ORBextractor.cc:1122) against an exact-rational model
  (fractions.Fraction) of the sampling position (d + 1/2) * (src / dst) - 1/2;
* cv::FAST's non-maximum suppression rule and output order (src/ORBextractor.cc:810-826) against a brute-force numpy 3x3 strict
  maximum over a score map that tests/test_pin_skimage.py pins to scikit-image;
* (second half of round 5) the OUTPUTS of cv::resize (src/ORBextractor.cc:1122), GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101
  (:1074-1076, both tap generations) and cv::remap INTER_LINEAR (src/System.cc:294) against real-valued models in double precision
  (numpy, scipy.ndimage.correlate1d / map_coordinates): within the fixed-point quantisation and unbiased;
* (round 6) the vector body of OpenCV 4.0 .. 4.5.0's vertical Gaussian pass on the 257-sum taps (a third blur mode: floor in the
  body, round in the scalar tail), cv::remap's 1/32 bilinear table and cv::cvtColor's fixed-point gray weights against
  exact-rational models.
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


# ---- second set (round 5): the OUTPUTS of the three image filters the oracle restates from memory, against float models that share
# no code with it (numpy / scipy.ndimage in double precision).  A fixed-point filter may differ from the exact real-valued filter
# only by its documented quantisation, so the bounds below are tight enough to catch a wrong sampling position, a wrong border rule,
# a wrong tap or a shifted kernel -- the failure modes of a restatement -- though not a different rounding of the last bit.

def _bilinear_float(src, dw, dh):
    """cv::resize INTER_LINEAR in real arithmetic: sample at ((i + 1/2) * s / d - 1/2), clamp coordinates to the image."""
    sh, sw = src.shape
    x = (np.arange(dw) + 0.5) * (sw / dw) - 0.5
    y = (np.arange(dh) + 0.5) * (sh / dh) - 0.5
    x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
    fx, fy = x - x0, y - y0
    xa, xb = np.clip(x0, 0, sw - 1), np.clip(x0 + 1, 0, sw - 1)
    ya, yb = np.clip(y0, 0, sh - 1), np.clip(y0 + 1, 0, sh - 1)
    s = src.astype(np.float64)
    top = s[ya][:, xa] * (1 - fx) + s[ya][:, xb] * fx
    bot = s[yb][:, xa] * (1 - fx) + s[yb][:, xb] * fx
    return top * (1 - fy)[:, None] + bot * fy[:, None]


@pytest.mark.parametrize("sw,sh,dw,dh", [(1280, 720, 1067, 600), (640, 480, 533, 400), (357, 201, 298, 167), (333, 517, 278, 431),
                                         (400, 300, 200, 150), (300, 200, 75, 50)])
def test_resize_output_within_quantisation_of_real_bilinear(oracle, sw, sh, dw, dh):
    """ComputePyramid's cv::resize (src/ORBextractor.cc:1122): 11-bit weights per axis, intermediate >> 4, two >> 16 products and a
    final (+2) >> 2.  Against the real-valued bilinear sample the result may be off by the weight rounding (255 * 2 * 2^-12), the
    truncations (< 1 gray level in total) and the final rounding (1/2): |oracle - real| < 1.25 everywhere, and unbiased."""
    rng = np.random.default_rng(sw * 31 + dw)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    src[: sh // 3] = np.clip(np.add.outer(np.arange(sh // 3), np.arange(sw)) % 256, 0, 255).astype(np.uint8)  # a smooth part too
    got = oracle.resize(src, dw, dh).astype(np.float64)
    want = _bilinear_float(src, dw, dh)
    d = got - want
    assert np.abs(d).max() < 1.25, np.abs(d).max()
    assert abs(d.mean()) < 0.2, d.mean()


def test_gaussian_blur_against_real_gaussian_and_reflect101(oracle):
    """GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of src/ORBextractor.cc:1074-1076: cv::getGaussianKernel(7, 2) = exp(-x^2 / 8)
    normalised; the 8-bit path uses 8.8 fixed-point taps per axis.  scipy.ndimage.correlate1d with the REAL kernel and
    mode='mirror' (= REFLECT_101: the edge pixel is not repeated) is the reference.  The current taps (18 34 48 56 48 34 18, sum
    256: OpenCV >= 4.5.1) stay within the taps' rounding and are unbiased; the taps of OpenCV 4.0 .. 4.5.0 (18 34 49 55 49 34 18)
    sum to 257 -- a constant image of 100 comes out as 101 -- i.e. they follow the real filter scaled by (257 / 256)^2."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (97, 123), dtype=np.uint8)
    img[40:60, 30:90] = 255
    img[:5, :] = 0
    k = np.exp(-(np.arange(-3, 4) ** 2) / 8.0)
    k /= k.sum()
    f64 = img.astype(np.float64)
    want = ndi.correlate1d(ndi.correlate1d(f64, k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    wrong = ndi.correlate1d(ndi.correlate1d(f64, k, axis=1, mode="reflect"), k, axis=0, mode="reflect")   # BORDER_REFLECT: edge repeated
    edge = np.zeros(img.shape, bool)
    edge[:3, :] = edge[-3:, :] = edge[:, :3] = edge[:, -3:] = True
    for variant, scale in ((451, 1.0), (440, (257.0 / 256.0) ** 2)):
        got = oracle.blur(img, variant).astype(np.float64)
        d = got - np.minimum(want * scale, 255.0)
        assert np.abs(d).max() < 1.5, (variant, np.abs(d).max())
        assert np.abs(d).mean() < 0.35 and abs(d.mean()) < 0.02, (variant, np.abs(d).mean(), d.mean())
        # the border rule alone: the model with the edge pixel repeated is ten times further away along the border
        assert np.abs(d)[edge].mean() * 5 < np.abs(got - np.minimum(wrong * scale, 255.0))[edge].mean()
    assert (oracle.blur(np.full((40, 40), 100, np.uint8), 451) == 100).all()
    assert (oracle.blur(np.full((40, 40), 100, np.uint8), 440) == 101).all()


def test_remap_within_quantisation_of_real_bilinear(oracle):
    """cv::remap INTER_LINEAR with CV_32FC1 maps (initUndistortRectifyMap + remap of the stereo examples, src/System.cc:294): the
    fractional position is quantised to 1/32, weights to 15 bits.  Against scipy.ndimage.map_coordinates(order=1) on the same float
    maps, inside the image: the 1/32 quantisation moves a sample by at most 1/64 px per axis, i.e. by at most (1/64 + 1/64) * the
    local gradient -- on a smooth image (gradient <= 4 per px) well under 1 gray level."""
    ndi = pytest.importorskip("scipy.ndimage")
    h, w = 120, 160
    yy, xx = np.mgrid[0:h, 0:w]
    img = (2.0 * xx + 1.5 * yy + 20 * np.sin(xx / 9.0) + 15 * np.cos(yy / 7.0)) % 256
    img = np.clip(np.where(np.abs(np.diff(img, axis=1, append=img[:, -1:])) > 100, 128, img), 0, 255).astype(np.uint8)
    rng = np.random.default_rng(3)
    mapx = (xx + 3.0 * np.sin(yy / 17.0) + rng.uniform(-0.5, 0.5, (h, w))).astype(np.float32)
    mapy = (yy + 2.0 * np.cos(xx / 23.0) + rng.uniform(-0.5, 0.5, (h, w))).astype(np.float32)
    got = oracle.remap(img, mapx, mapy).astype(np.float64)
    want = ndi.map_coordinates(img.astype(np.float64), [mapy.astype(np.float64), mapx.astype(np.float64)], order=1, mode="nearest")
    inside = (mapx >= 1) & (mapx <= w - 2) & (mapy >= 1) & (mapy <= h - 2)
    g = np.maximum(np.abs(np.diff(img.astype(float), axis=1, append=0)), np.abs(np.diff(img.astype(float), axis=0, append=0)))
    iy, ix = np.clip(np.rint(mapy).astype(int), 0, h - 1), np.clip(np.rint(mapx).astype(int), 0, w - 1)
    smooth = inside & (ndi.maximum_filter(g, 5)[iy, ix] <= 8)        # (the gradient around the SOURCE position of the sample)
    d = (got - want)[smooth]
    assert smooth.sum() > 8000
    assert np.abs(d).max() < 1.0, np.abs(d).max()
    assert abs(d.mean()) < 0.2


# ----------------------------------------------------------------------------------------------------------------- round 6
def _vline_simd_body(hp, taps):
    """The arithmetic of smooth.simd.hpp's vlineSmoothONa_yzy_a<uint8_t, ufixedpoint16> body as published (OpenCV 4.0 .. 4.5.0),
    lane by lane in numpy: 8.8 rows re-biased into int16 (v_add_wrap with 0x8000), int32 dot products with the int16 taps
    (v_dotprod), + the constant 128 << 16, v_rshr_pack<16> (add 1 << 15, arithmetic shift, saturate to int16), then the
    unsigned 16 -> 8 pack (saturate to 0 .. 255).  hp: [7][n] uint16 rows, taps: 7 ints."""
    s16 = (hp.astype(np.int64) - 32768)                         # the wrapped add as a signed value
    assert (s16 >= -32768).all() and (s16 <= 32767).all()
    acc = (s16 * np.asarray(taps, np.int64)[:, None]).sum(0) + (128 << 16)
    assert (np.abs(acc) < 2 ** 31).all()                        # int32 lanes do not overflow
    r = np.clip((acc + (1 << 15)) >> 16, -32768, 32767)         # v_rshr_pack<16>: int32 -> int16 with rounding and saturation
    return np.clip(r, 0, 255).astype(np.uint8)                  # v_pack_u: int16 -> uint8 with saturation


def test_gaussian_blur_440_vector_body_floors_and_scalar_tail_rounds(oracle):
    """VERDICT (round 5, item 7): OpenCV 4.0 .. 4.5.0's vectorised vertical pass gives the -32768 bias of its int16 rows back as
    the constant 128 << 16 = 32768 * 256; the 4.0 .. 4.5.0 taps sum to 257, so the result is 32768 short -- the rounding half.
    The oracle models this as variants 44016 / 44032 (body = the first (w / lanes) * lanes columns, orb_oracle.cpp
    gaussian_blur7).  Pinned here: the known answer flat 100 -> 100 in the body and 101 in the tail; the body columns equal the
    lane-level emulation of the published vector arithmetic (_vline_simd_body), the tail columns the rounded scalar sum; an image
    narrower than the vector is all tail; the >= 4.5.1 taps (sum 256) give the same bytes through the body arithmetic as through
    the rounded sum, i.e. the mode exists only for the 257-sum taps."""
    t440, t451 = np.array([18, 34, 49, 55, 49, 34, 18]), np.array([18, 34, 48, 56, 48, 34, 18])
    flat = np.full((12, 75), 100, np.uint8)
    for variant, lanes in ((44016, 16), (44032, 32)):
        out = oracle.blur(flat, variant)
        body = (75 // lanes) * lanes
        assert (out[:, :body] == 100).all() and (out[:, body:] == 101).all(), variant
    assert (oracle.blur(flat, 440) == 101).all() and (oracle.blur(flat, 451) == 100).all()
    assert (oracle.blur(flat[:, :20], 44032) == 101).all()       # w < lanes: the vector loop never runs
    assert (oracle.blur(flat[:, :20], 44016)[:, :16] == 100).all()
    rng = np.random.default_rng(606)
    img = rng.integers(0, 256, (40, 117), dtype=np.uint8)
    img[5:20, 30:80] = np.where(rng.random((15, 50)) < 0.5, 255, 252)          # saturation: sums reach 256.9
    h, w = img.shape

    def rows_h(taps):   # the horizontal pass in exact integers with BORDER_REFLECT_101 and the ufixedpoint16 saturation
        xi = np.abs(np.arange(-3, w + 3))
        xi = np.where(xi >= w, 2 * w - 2 - xi, xi)
        pad = img[:, xi].astype(np.int64)
        hp = sum(int(taps[k]) * pad[:, k:k + w] for k in range(7))
        return np.minimum(hp, 65535)

    for taps, variants in ((t440, (44016, 44032)), (t451, ())):
        hp = rows_h(taps)
        yi = np.abs(np.arange(-3, h + 3))
        yi = np.where(yi >= h, 2 * h - 2 - yi, yi)
        hpp = hp[yi]
        simd = np.stack([_vline_simd_body(hpp[y:y + 7], taps) for y in range(h)])
        exact = sum(int(taps[k]) * hpp[k:k + h] for k in range(7))
        rounded = np.minimum((exact + 32768) >> 16, 255).astype(np.uint8)
        floored = np.minimum(exact >> 16, 255).astype(np.uint8)
        if taps is t451:
            assert np.array_equal(simd, rounded)                 # sum 256: the bias constant is exact, body == scalar
            assert np.array_equal(oracle.blur(img, 451), rounded)
            continue
        assert np.array_equal(simd, floored)                     # sum 257: the body floors
        assert (floored != rounded).mean() > 0.3                 # (and that is not a rare event: about half of the pixels)
        assert np.array_equal(oracle.blur(img, 440), rounded)
        for variant in variants:
            lanes = variant - 44000
            body = (w // lanes) * lanes
            out = oracle.blur(img, variant)
            assert np.array_equal(out[:, :body], simd[:, :body]) and np.array_equal(out[:, body:], rounded[:, body:]), variant


def test_remap_table_against_exact_rational_bilinear(oracle):
    """cv::remap INTER_LINEAR on CV_8U (src/System.cc:294): coordinates are quantised to 1/32 (cvRound(x * 32), INTER_BITS = 5),
    the four weights of a cell come from a 32 x 32 table of int16 values scaled by 2^15 whose entries are exact --
    (32 - fx)(32 - fy) * 32 etc. -- and the result is (sum + 2^14) >> 15.  With exact weights the output must equal the
    exact-rational bilinear value at the QUANTISED position rounded half up, for every one of the 1024 fractional cells and for
    every pixel configuration -- computed here with fractions.Fraction, no floating point and no oracle code."""
    rng = np.random.default_rng(607)
    src = rng.integers(0, 256, (9, 11), dtype=np.uint8)
    fy, fx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    for (iy, ix) in ((0, 0), (3, 5), (7, 9)):
        mx = (ix + fx / 32.0).astype(np.float32)
        my = (iy + fy / 32.0).astype(np.float32)
        out = oracle.remap(src, mx, my)
        for a in range(32):
            for b in range(32):
                p00, p01, p10, p11 = (int(src[iy, ix]), int(src[iy, ix + 1]), int(src[iy + 1, ix]), int(src[iy + 1, ix + 1]))
                u, v = Fraction(b, 32), Fraction(a, 32)
                exact = (1 - v) * ((1 - u) * p00 + u * p01) + v * ((1 - u) * p10 + u * p11)
                want = (exact + Fraction(1, 2)).__floor__()
                assert int(out[a, b]) == want, (iy, ix, a, b)
    # the quantisation rule itself: cvRound is round-half-to-even on x * 32 evaluated in float
    xs = np.array([2.0 + k / 64.0 for k in range(64)], np.float32)            # every half step between two table entries
    out = oracle.remap(src, xs[None, :], np.full((1, 64), 4.0, np.float32))
    for k in range(64):
        q = int(np.rint(np.float32(xs[k]) * np.float32(32.0)))                # rint = half to even
        ixq, fq = q >> 5, q & 31
        exact = (1 - Fraction(fq, 32)) * int(src[4, ixq]) + Fraction(fq, 32) * int(src[4, ixq + 1])
        assert int(out[0, k]) == (exact + Fraction(1, 2)).__floor__(), k


def test_gray_weights_against_exact_rational_coefficients(oracle):
    """cv::cvtColor(.., COLOR_RGB2GRAY / BGR2GRAY) on CV_8U (src/Tracking.cc:1394-1412): Y = 0.299 R + 0.587 G + 0.114 B in fixed
    point.  OpenCV >= 3.4.2 / 4.x: 15 fractional bits, R and G weights rounded to nearest, B = 2^15 - R - G so that the weights
    sum to exactly one (9798, 19235, 3735); older releases: 14 bits, all three rounded (4899, 9617, 1868 -- they happen to sum to
    2^14).  Pinned: the constants against the exact rationals; white stays white; and on random pixels the output is within
    1/2 + (the weights' quantisation error) of the exact-rational luma -- in integers, no floating point."""
    cR, cG, cB = Fraction(299, 1000), Fraction(587, 1000), Fraction(114, 1000)
    for bits, (wr, wg, wb) in ((15, (9798, 19235, 3735)), (14, (4899, 9617, 1868))):
        one = 1 << bits
        assert wr == round(cR * one) and wg == round(cG * one) and wr + wg + wb == one
        assert abs(Fraction(wb, one) - cB) < Fraction(1, one)
        rng = np.random.default_rng(608 + bits)
        px = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
        px[0, :4] = [[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 0, 255]]
        got = oracle.cvt_gray(px, True, bits).astype(np.int64)
        r, g, b = (px[..., i].astype(np.int64) for i in range(3))
        assert np.array_equal(got, (r * wr + g * wg + b * wb + one // 2) >> bits)      # the fixed-point formula itself
        exact1000 = 299 * r + 587 * g + 114 * b                                         # 1000 x the exact-rational luma
        slack = 500 + int(1000 * 255 * (abs(Fraction(wr, one) - cR) + abs(Fraction(wg, one) - cG) + abs(Fraction(wb, one) - cB))) + 1
        assert (np.abs(1000 * got - exact1000) <= slack).all()
        assert got[0, 0] == 255 and got[0, 1] == 0
