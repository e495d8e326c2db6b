"""Known-answer and property tests that pin the oracle's OpenCV-kernel restatements.

The reference holds no tests for this path and OpenCV is not installable here, so these KATs are the
hand-derivable answers of Appendix B plus independent numpy re-derivations of the same published
formulas (PARITY UNPINNED against real OpenCV — see oracle/orb_oracle.h).
"""
import ctypes
import ctypes.util

import numpy as np
import pytest

from orb_slam3_fast_amd import synth

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2),
        (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


# ------------------------------------------------------------------------------------------- B2 resize
def np_resize(src, dw, dh):
    """independent vectorised restatement of the 11-bit fixed-point bilinear resize"""
    sh, sw = src.shape
    def coef(dn, sn, clamp_x):
        scale = 1.0 / (float(dn) / sn)
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_x:
            lo = s < 0
            f[lo], s[lo] = 0, 0
            hi = s >= sn - 1
            f[hi], s[hi] = 0, sn - 1
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, c0, c1
    sx, a0, a1 = coef(dw, sw, True)
    sy, b0, b1 = coef(dh, sh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    H = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]
    r0 = H[np.clip(sy, 0, sh - 1)]
    r1 = H[np.clip(sy + 1, 0, sh - 1)]
    out = ((((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2)
    return out.astype(np.uint8)


def test_resize_constant_and_identity(oracle):
    c = np.full((100, 120), 77, np.uint8)
    assert np.all(oracle.resize(c, 100, 83) == 77)
    rng = np.random.default_rng(1)
    im = rng.integers(0, 256, (50, 64), dtype=np.uint8)
    assert np.array_equal(oracle.resize(im, 64, 50), im)


def test_resize_exact_half_equals_area_average(oracle):
    rng = np.random.default_rng(8)
    im = rng.integers(0, 256, (96, 128), dtype=np.uint8)
    a = im.astype(np.int32)
    box = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(oracle.resize(im, 64, 48), box)


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 533, 400), (357, 201, 298, 168), (64, 48, 53, 40),
                                         (1280, 720, 1067, 600), (100, 100, 250, 130)])
def test_resize_vs_numpy_restatement(oracle, sw, sh, dw, dh):
    rng = np.random.default_rng(sw * 7 + dh)
    im = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    assert np.array_equal(oracle.resize(im, dw, dh), np_resize(im, dw, dh))


# ------------------------------------------------------------------------------------------- B3 FAST
def ring_image(center, ring_vals, size=15, bg=None):
    im = np.full((size, size), center if bg is None else bg, np.uint8)
    c = size // 2
    im[c, c] = center
    for (dx, dy), v in zip(RING, ring_vals):
        im[c + dy, c + dx] = v
    return im, c


def test_fast_kats(oracle):
    assert len(oracle.fast(np.full((40, 40), 100, np.uint8), 20)) == 0
    im, c = ring_image(100, [130] * 9 + [100] * 7)
    r = oracle.fast(im, 20, nms=False)
    assert [c, c, 0] in r.tolist() or any((p[0] == c and p[1] == c) for p in r)
    r = oracle.fast(im, 20, nms=True)
    hit = [p for p in r if p[0] == c and p[1] == c]
    assert len(hit) == 1 and hit[0][2] == 29          # score = M - 1 with M = 30
    im8, c = ring_image(100, [130] * 8 + [100] * 8)
    assert not any(p[0] == c and p[1] == c for p in oracle.fast(im8, 20, nms=False))
    # dark arc, wrapping around index 15 -> 0
    vals = [100] * 16
    for k in (13, 14, 15, 0, 1, 2, 3, 4, 5):
        vals[k] = 40
    imw, c = ring_image(100, vals)
    hit = [p for p in oracle.fast(imw, 20) if p[0] == c and p[1] == c]
    assert len(hit) == 1 and hit[0][2] == 59


def test_fast_equal_neighbours_both_suppressed(oracle):
    # strict '>' in the NMS: two adjacent corners with identical scores kill each other.
    # mirror-symmetric image => mirrored pixels get equal scores
    rng = np.random.default_rng(5)
    half = rng.integers(0, 256, (40, 20), dtype=np.uint8)
    sym = np.hstack([half, half[:, ::-1]])          # columns 19 | 20 are mirror images
    allc = oracle.fast(sym, 10, nms=False)
    kept = oracle.fast(sym, 10, nms=True)
    s_all = {(x, y) for x, y, _ in allc.tolist()}
    pairs = [y for y in range(3, 37) if (19, y) in s_all and (20, y) in s_all]
    assert pairs
    for y in pairs:
        assert not any((p[0] in (19, 20)) and p[1] == y for p in kept.tolist())


def fast_score_map(img, tmin):
    """level-wide model: S(p) = M-1 if M > tmin else 0, M = max over 9-arcs of min |diff| of one sign"""
    h, w = img.shape
    I = img.astype(np.int32)
    S = np.zeros((h, w), np.int32)
    d = np.zeros((25, h - 6, w - 6), np.int32)
    c = I[3:h - 3, 3:w - 3]
    for k in range(25):
        dx, dy = RING[k % 16]
        d[k] = c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx]
    M = np.full((h - 6, w - 6), -999, np.int32)
    for k in range(16):
        arc = d[k:k + 9]
        M = np.maximum(M, arc.min(0))
        M = np.maximum(M, (-arc).min(0))
    S[3:h - 3, 3:w - 3] = np.where(M > tmin, M - 1, 0)
    return S


def levelwide_candidates(img, ini_th, min_th):
    """The level-wide / cell-masked formulation the HIP path uses, in the reference's order."""
    h, w = img.shape
    S = fast_score_map(img, min_th)
    minB, maxBX, maxBY = 16, w - 16, h - 16
    width, height = np.float32(maxBX - minB), np.float32(maxBY - minB)
    nC, nR = int(width / np.float32(35)), int(height / np.float32(35))
    wC, hC = int(np.ceil(width / np.float32(nC))), int(np.ceil(height / np.float32(nR)))
    out = []
    for i in range(nR):
        iniY = minB + i * hC
        maxY = iniY + hC + 6
        if iniY >= maxBY - 3:
            continue
        maxY = min(maxY, maxBY)
        for j in range(nC):
            iniX = minB + j * wC
            maxX = iniX + wC + 6
            if iniX >= maxBX - 6:
                continue
            maxX = min(maxX, maxBX)
            y0, y1, x0, x1 = iniY + 3, maxY - 3, iniX + 3, maxX - 3      # detectable window of the cell
            if y1 <= y0 or x1 <= x0:
                continue
            T = np.zeros((y1 - y0 + 2, x1 - x0 + 2), np.int32)
            T[1:-1, 1:-1] = S[y0:y1, x0:x1]
            cen = T[1:-1, 1:-1]
            keep = cen > 0
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if dx or dy:
                        keep &= cen > T[1 + dy:T.shape[0] - 1 + dy, 1 + dx:T.shape[1] - 1 + dx]
            k20 = keep & (cen >= ini_th)
            sel = k20 if k20.any() else keep
            ys, xs = np.nonzero(sel)
            for y, x in zip(ys, xs):
                out.append((x + x0 - 16, y + y0 - 16, cen[y, x]))
    return np.array(out, np.int32).reshape(-1, 3)


@pytest.mark.parametrize("w,h,stream", [(320, 240, 3), (389, 277, 4), (241, 250, 5)])
def test_cellwise_fast_equals_levelwide_formulation(oracle, w, h, stream):
    img = synth.mono_frame(w, h, stream)
    img[: h // 3, : w // 2] = (img[: h // 3, : w // 2] // 8) + 100     # low-contrast zone -> min-threshold cells
    ex = oracle.OracleExtractor(500, 1.2, 1, 20, 7)
    ex.compute_pyramid(img)
    c = ex.detect_candidates(0)
    lit = np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)
    mod = levelwide_candidates(img, 20, 7)
    assert len(lit) > 50 and (lit[:, 2] < 20).any() and (lit[:, 2] >= 20).any()
    assert np.array_equal(lit, mod)


# ------------------------------------------------------------------------------------------- B4 blur
def test_blur_kats(oracle):
    assert np.all(oracle.blur(np.full((30, 40), 93, np.uint8)) == 93)
    im = np.zeros((21, 21), np.uint8)
    im[10, 10] = 255
    out = oracle.blur(im)
    assert out[10, 10] == 12 and out[10, 9] == (56 * 48 * 255 + 32768) >> 16
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)
    rng = np.random.default_rng(2)
    im = rng.integers(0, 256, (37, 45), dtype=np.uint8)
    pad = np.pad(im.astype(np.int64), 3, mode="reflect")          # numpy 'reflect' == BORDER_REFLECT_101
    acc = np.zeros(im.shape, np.int64)
    for i in range(7):
        for j in range(7):
            acc += k[i] * k[j] * pad[i:i + 37, j:j + 45]
    assert np.array_equal(oracle.blur(im), ((acc + 32768) >> 16).astype(np.uint8))
    assert oracle.blur(im, 440).shape == im.shape


# ------------------------------------------------------------------------------------------- B5 / B8
def test_fast_atan2_kats(oracle):
    assert oracle.fast_atan2(0, 0) == 0
    assert oracle.fast_atan2(0, 1) == 0
    assert abs(oracle.fast_atan2(1, 0) - 90) < 1e-4
    assert abs(oracle.fast_atan2(0, -1) - 180) < 1e-4
    assert abs(oracle.fast_atan2(-1, 0) - 270) < 1e-4
    assert abs(oracle.fast_atan2(1, 1) - 45) < 0.02
    rng = np.random.default_rng(3)
    for y, x in rng.integers(-30000, 30000, (2000, 2)):
        a = oracle.fast_atan2(y, x)
        ref = np.degrees(np.arctan2(float(y), float(x))) % 360
        assert 0 <= a <= 360 and min(abs(a - ref), 360 - abs(a - ref)) < 0.35


def test_cv_round_half_even(oracle):
    f = oracle.lib().oro_cv_round_f
    assert [f(0.5), f(1.5), f(2.5), f(-0.5), f(-1.5), f(2.4999), f(-2.5)] == [0, 2, 2, 0, -2, 2, -2]


def test_sincos_vs_libm(oracle):
    """The reference calls libm cosf / sinf (src/ORBextractor.cc:106-107); the oracle's default is its restatement of
    glibc's two ifunc variants (the definition the device runs, csrc/orbx_sincos.h), which must EQUAL the host libm bit
    for bit on a glibc >= 2.28 x86-64 host (skipped on any other libm: there the restatement, not the host, is the oracle): 1.2e7 angles drawn the way the path forms them (fastAtan2 of integer moments, times factorPI),
    plus a strided sweep of all floats in [0, 2 pi] (tools/sincos_sweep.cpp is the exhaustive version: 1.09e9
    arguments, 0 mismatches for either variant on glibc 2.35)."""
    if oracle.host_libm_variant() not in (1, 2):
        pytest.skip("host libm is not a glibc >= 2.28 x86-64 sinf/cosf: nothing to compare the restatement with")
    for fused in (1, 0):
        bad, first = oracle.sincos_check(20220131, 12_000_000, fused)
        assert bad == 0, "glibc model (fused=%d) differs from the host libm, first at angle %r" % (fused, first)
    # spot values: exact quadrant angles as the path forms them, and agreement with float64 to half an ulp
    fpi = np.float32(np.pi / np.float32(180.0))
    for deg in (0, 45, 90, 135, 180, 225, 270, 315, 360):
        a = float(np.float32(deg) * fpi)
        s, c = oracle.sincosf(a)
        assert (s, c) == oracle.sincos_model(a, 1) == oracle.sincos_model(a, 0)
        assert abs(s - np.sin(np.float64(a))) < 6e-8 and abs(c - np.cos(np.float64(a))) < 6e-8
    ang = oracle.reachable_angles(7, 100000)
    ls, lc = oracle.libm_sincos(ang)
    assert np.all(np.abs(ls - np.sin(ang.astype(np.float64))) <= 0.57 * np.spacing(np.abs(ls)) + 1e-45)  # glibc: < 0.56 ulp
    assert np.all(np.abs(lc - np.cos(ang.astype(np.float64))) <= 0.57 * np.spacing(np.abs(lc)) + 1e-45)


def test_sincos_modes_agree_in_descriptors(oracle):
    """Descriptor bytes are the same whichever sin/cos definition is selected (host libm, FMA model, SSE2 model)."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    try:
        out = []
        for mode in (0, 1, 2):
            oracle.set_sincos_mode(mode)
            out.append([oracle.descriptor(img, 32, 32, float(a)).tobytes() for a in np.linspace(0, 359.9, 720, dtype=np.float32)])
        assert out[1] == out[2]
        if oracle.host_libm_variant() in (1, 2):
            assert out[0] == out[1]
    finally:
        oracle.set_sincos_mode(1)   # the default: glibc model = the device's definition


# ------------------------------------------------------------------------------------------- A5 / A7
def test_ic_angle_symmetry(oracle):
    im = np.zeros((41, 41), np.uint8)
    im[:, 21:] = 200                                # brighter to the right -> angle 0
    assert oracle.ic_angle(im, 20, 20) == 0.0
    assert abs(oracle.ic_angle(im.T.copy(), 20, 20) - 90) < 1e-3
    assert abs(oracle.ic_angle(im[:, ::-1].copy(), 20, 20) - 180) < 1e-3


def test_descriptor_bit_order(oracle):
    # angle 0: a=1,b=0 -> sample (y, x) offsets straight from the pattern; bit i of byte k = test 8k+i
    rng = np.random.default_rng(6)
    im = rng.integers(0, 256, (61, 61), dtype=np.uint8)
    d = oracle.descriptor(im, 30.0, 30.0, 0.0)
    ptr = oracle.lib().oro_pattern()
    pat = np.frombuffer((ctypes.c_int8 * 1024).from_address(ptr), np.int8).reshape(256, 4).astype(int)
    exp = np.zeros(32, np.uint8)
    for t, (x0, y0, x1, y1) in enumerate(pat):
        if im[30 + y0, 30 + x0] < im[30 + y1, 30 + x1]:
            exp[t // 8] |= 1 << (t % 8)
    assert np.array_equal(d, exp)


def test_hamming(oracle):
    rng = np.random.default_rng(7)
    a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
    assert oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
    assert oracle.hamming(a, a) == 0 and oracle.hamming(a, ~a) == 256


def test_ic_angle_vs_numpy_restatement(oracle):
    """IC_Angle (src/ORBextractor.cc:75-99) transcribed with numpy: integer moments over the circular patch, then
    cv::fastAtan2 of (m01, m10)."""
    rng = np.random.default_rng(21)
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    for _ in range(40):
        im = rng.integers(0, 256, (50, 60), dtype=np.uint8)
        cx, cy = int(rng.integers(16, 44)), int(rng.integers(16, 34))
        c = im.astype(np.int64)
        m10 = sum(u * c[cy, cx + u] for u in range(-15, 16))
        m01 = 0
        for v in range(1, 16):
            v_sum = 0
            for u in range(-umax[v], umax[v] + 1):
                plus, minus = c[cy + v, cx + u], c[cy - v, cx + u]
                v_sum += plus - minus
                m10 += u * (plus + minus)
            m01 += v * v_sum
        assert oracle.ic_angle(im, cx, cy) == oracle.fast_atan2(float(np.float32(m01)), float(np.float32(m10)))


def test_rotated_descriptor_vs_numpy_restatement(oracle):
    """computeOrbDescriptor (src/ORBextractor.cc:102-147) transcribed with numpy float32: a = cos, b = sin of
    angle * (float)(CV_PI / 180.f) (through the oracle's defined sin/cos), sample rows
    cvRound(x b + y a), columns cvRound(x a - y b) with separately rounded float products, bit i of byte k = test 8k + i."""
    f32 = np.float32
    rng = np.random.default_rng(22)
    ptr = oracle.lib().oro_pattern()
    pat = np.frombuffer((ctypes.c_int8 * 1024).from_address(ptr), np.int8).reshape(512, 2).astype(np.int64)
    factor_pi = f32(np.pi / f32(180.0))

    def cv_round(v):  # cvRound(float): round half to even
        return int(np.rint(np.float64(v)))

    for _ in range(25):
        im = rng.integers(0, 256, (64, 64), dtype=np.uint8)
        px, py = float(rng.integers(20, 44)), float(rng.integers(20, 44))
        angle = f32(rng.uniform(0, 360))
        s, c = oracle.sincosf(float(angle * factor_pi))
        a, b = f32(c), f32(s)
        exp = np.zeros(32, np.uint8)
        for i in range(256):
            vals = []
            for x, y in (pat[2 * i], pat[2 * i + 1]):
                row = cv_round(f32(f32(x) * b) + f32(f32(y) * a))
                col = cv_round(f32(f32(x) * a) - f32(f32(y) * b))
                vals.append(int(im[int(py) + row, int(px) + col]))
            if vals[0] < vals[1]:
                exp[i // 8] |= 1 << (i % 8)
        assert np.array_equal(oracle.descriptor(im, px, py, float(angle)), exp)


def test_resize_geometry_vs_torch_float_bilinear(oracle):
    """cv::resize(INTER_LINEAR) samples at half-pixel centres, like torch's bilinear interpolate with
    align_corners=False (no antialiasing); the 11-bit fixed-point arithmetic may differ from the float result by one
    gray level.  An independent check of the coefficient geometry (sx, fx) of B2."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(31)
    for (sh, sw, dh, dw) in ((720, 1280, 600, 1067), (480, 752, 400, 627), (240, 320, 201, 267), (100, 90, 84, 75)):
        img = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float64))[None, None], size=(dh, dw), mode="bilinear",
                                              align_corners=False)[0, 0].numpy()
        got = oracle.resize(img, dw, dh).astype(np.float64)
        assert np.abs(got - ref).max() <= 1.0 + 1e-9
        assert np.abs(got - np.rint(ref)).mean() < 0.2  # equal to the rounded float value in ~87 % of the pixels (the
        # intermediate >> 4 and >> 16 truncations bias the fixed-point result down by a fraction of a level)


def test_blur_vs_float_gaussian(oracle):
    """The fixed-point 7x7 blur against scipy's float convolution with the normalised sigma = 2 Gaussian (mirror border =
    BORDER_REFLECT_101): the integer taps {18, 34, 48, 56, ...} / 256 are that kernel to 8 bits, so results agree within
    two gray levels."""
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(32)
    img = rng.integers(0, 256, (90, 120), dtype=np.uint8)
    x = np.arange(-3, 4)
    g = np.exp(-x * x / (2 * 2.0 * 2.0))
    g /= g.sum()
    assert np.abs(g * 256 - np.array([18, 34, 48, 56, 48, 34, 18])).max() < 0.9
    ref = ndi.convolve1d(ndi.convolve1d(img.astype(np.float64), g, axis=0, mode="mirror"), g, axis=1, mode="mirror")
    assert np.abs(oracle.blur(img).astype(np.float64) - ref).max() < 2.0


def test_fast_against_the_definition(oracle):
    """FAST-9-16 from its definition, by brute force: p is a corner at threshold t iff 9 contiguous ring pixels are all
    > I(p) + t or all < I(p) - t; cornerScore = the largest t at which p is still a corner (OpenCV's documented
    meaning); non-maximum suppression keeps p iff its score is strictly greater than the 8 neighbours' scores."""
    rng = np.random.default_rng(41)
    base = rng.integers(0, 256, (6, 7), dtype=np.uint8)
    img = np.kron(base, np.ones((6, 6), np.uint8))
    img = (img.astype(np.int16) + rng.integers(-25, 26, img.shape)).clip(0, 255).astype(np.uint8)
    h, w = img.shape
    I = img.astype(np.int32)

    def is_corner(x, y, t):
        ring = [int(I[y + dy, x + dx]) for dx, dy in RING]
        c = int(I[y, x])
        for s in range(16):
            arc = [ring[(s + k) % 16] for k in range(9)]
            if all(v > c + t for v in arc) or all(v < c - t for v in arc):
                return True
        return False

    th = 20
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if is_corner(x, y, th):
                t = th
                while t + 1 <= 255 and is_corner(x, y, t + 1):
                    t += 1
                score[y, x] = t
    exp_all = {(x, y, int(score[y, x])) for y in range(h) for x in range(w) if score[y, x] > 0}
    got_all = {tuple(p) for p in oracle.fast(img, th, nms=False).tolist()}
    assert {(x, y) for x, y, _ in got_all} == {(x, y) for x, y, _ in exp_all} and len(exp_all) > 20
    exp_nms = set()
    for (x, y, s) in exp_all:
        nb = [score[y + j, x + i] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i, j) != (0, 0)]
        if all(s > v for v in nb):
            exp_nms.add((x, y, s))
    assert {tuple(p) for p in oracle.fast(img, th, nms=True).tolist()} == exp_nms and len(exp_nms) > 5
