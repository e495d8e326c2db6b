"""Stereo rectification and CLAHE in front of the extractor: cv::remap(im, out, M1, M2, INTER_LINEAR)
of System::TrackStereo (src/System.cc:294-295, maps from src/Settings.cc:557-572) and cv::createCLAHE(3.0, (8, 8))->apply of
the TUM-VI front ends (Examples/Stereo/stereo_tum_vi.cc:100,142-143).

The oracle restates OpenCV's arithmetic from its published algorithm (no OpenCV here: "parity unpinned", like every OpenCV
kernel of the oracle); these tests check it against an independently written numpy model and against properties that
follow from the definition, and the HIP kernels against the oracle bit for bit."""
import os

import numpy as np
import pytest

from orb_slam3_fast_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "rectify_clahe.npz")


# ---- independent numpy models ---------------------------------------------------------------------------------------------------
def np_remap(img, mx, my):
    img = img.astype(np.int64)
    h, w = img.shape

    def rnd(m):
        t = m.astype(np.float32) * np.float32(32)
        ok = np.abs(t) < 2147483648.0  # False for NaN
        return np.where(ok, np.rint(np.where(ok, t, 0)), -2147483648).astype(np.int64)

    fsx, fsy = rnd(mx), rnd(my)
    sx, sy = np.clip(fsx >> 5, -32768, 32767), np.clip(fsy >> 5, -32768, 32767)
    fx, fy = fsx & 31, fsy & 31
    wt = [(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    zero = (fx == 0) & (fy == 0)
    wt[0] = np.where(zero, 32767, wt[0])
    wt[3] = np.where(zero, 1, wt[3])
    acc = np.zeros(mx.shape, np.int64)
    for k, (ox, oy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        xx, yy = sx + ox, sy + oy
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        acc += np.where(ok, img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0) * wt[k]
    return ((acc + 16384) >> 15).astype(np.uint8)


def np_clahe(img, clip_limit=3.0, tiles=(8, 8)):
    h, w = img.shape
    tx_n, ty_n = tiles
    ext = img
    if w % tx_n or h % ty_n:
        ext = np.pad(img, ((0, ty_n - h % ty_n), (0, tx_n - w % tx_n)), mode="reflect")  # numpy "reflect" == BORDER_REFLECT_101
    eh, ew = ext.shape
    tw, th = ew // tx_n, eh // ty_n
    area = tw * th
    scale = np.float32(255) / np.float32(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    lut = np.zeros((ty_n, tx_n, 256), np.float32)
    for ty in range(ty_n):
        for tx in range(tx_n):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                hist += clipped // 256
                residual = clipped % 256
                if residual:
                    step = max(256 // residual, 1)
                    idx = np.arange(0, 256, step)[:residual]
                    hist[idx] += 1
            cs = np.cumsum(hist).astype(np.float32)
            lut[ty, tx] = np.clip(np.rint(cs * scale), 0, 255)
    inv_tw, inv_th = np.float32(1.0) / np.float32(tw), np.float32(1.0) / np.float32(th)
    txf = np.arange(w, dtype=np.float32) * inv_tw - np.float32(0.5)
    tyf = np.arange(h, dtype=np.float32) * inv_th - np.float32(0.5)
    tx1, ty1 = np.floor(txf).astype(np.int64), np.floor(tyf).astype(np.int64)
    xa, ya = (txf - tx1.astype(np.float32))[None, :], (tyf - ty1.astype(np.float32))[:, None]
    xa1, ya1 = np.float32(1) - xa, np.float32(1) - ya
    tx2, ty2 = np.minimum(tx1 + 1, tx_n - 1)[None, :], np.minimum(ty1 + 1, ty_n - 1)[:, None]
    tx1, ty1 = np.maximum(tx1, 0)[None, :], np.maximum(ty1, 0)[:, None]
    v = img.astype(np.int64)
    res = (lut[ty1, tx1, v] * xa1 + lut[ty1, tx2, v] * xa) * ya1 + (lut[ty2, tx1, v] * xa1 + lut[ty2, tx2, v] * xa) * ya
    assert res.dtype == np.float32
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def _maps(rng, dw, dh, sw, sh, kind):
    u, v = np.meshgrid(np.arange(dw, dtype=np.float32), np.arange(dh, dtype=np.float32))
    if kind == "rectify":
        return synth.rectify_maps(dw, dh, sw, sh, seed=int(rng.integers(1 << 20)))
    if kind == "random":  # anywhere, including far outside the source
        return (rng.uniform(-8, sw + 8, (dh, dw)).astype(np.float32), rng.uniform(-8, sh + 8, (dh, dw)).astype(np.float32))
    if kind == "grid32":  # every 1/32 fraction, exactly representable
        return ((u * (sw - 1) / max(dw - 1, 1)).astype(np.float32) + (rng.integers(0, 32, (dh, dw)) / 32).astype(np.float32),
                (v * (sh - 1) / max(dh - 1, 1)).astype(np.float32) + (rng.integers(0, 32, (dh, dw)) / 32).astype(np.float32))
    if kind == "special":
        mx = rng.uniform(-2, sw + 2, (dh, dw)).astype(np.float32)
        my = rng.uniform(-2, sh + 2, (dh, dw)).astype(np.float32)
        mx[0, :6] = [np.nan, np.inf, -np.inf, 1e12, -1e12, 3e9]
        my[1, :6] = [np.nan, np.inf, -np.inf, 1e12, -1e12, 7e7]
        mx[2, :4] = [-1.0, -0.5, sw - 1.0, sw - 0.5]
        my[3, :4] = [-1.0, -0.5, sh - 1.0, sh - 0.5]
        return mx, my
    raise ValueError(kind)


# ---- oracle vs model --------------------------------------------------------------------------------------------------------------
def test_oracle_remap_known_answers(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 56), dtype=np.uint8)
    h, w = img.shape
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    assert np.array_equal(oracle.remap(img, u, v), img)  # weights {32767, 0, 0, 1} still give the source pixel
    out = oracle.remap(img, u + 3, v - 2)  # integer shift, zero border
    assert np.array_equal(out[2:, :w - 3], img[:h - 2, 3:]) and not out[:2].any() and not out[:, w - 3:].any()
    half = oracle.remap(img, u + 0.5, v)  # half-pixel: (a + b + 1) >> 1, last column blends with the 0 border
    a = img.astype(np.int32)
    b = np.concatenate([a[:, 1:], np.zeros((h, 1), np.int32)], 1)
    assert np.array_equal(half, ((a + b + 1) >> 1).astype(np.uint8))
    q = oracle.remap(img, u + 0.25, v + 0.75)  # weights 6144 / 2048 / 18432 / 6144
    c = np.concatenate([a[1:], np.zeros((1, w), np.int32)], 0)
    d = np.concatenate([b[1:], np.zeros((1, w), np.int32)], 0)
    assert np.array_equal(q, ((a * 6144 + b * 2048 + c * 18432 + d * 6144 + 16384) >> 15).astype(np.uint8))


@pytest.mark.parametrize("kind", ["rectify", "random", "grid32", "special"])
def test_oracle_remap_matches_numpy_model(oracle, kind):
    rng = np.random.default_rng({"rectify": 1, "random": 2, "grid32": 3, "special": 4}[kind])
    for (sw, sh, dw, dh) in ((64, 48, 64, 48), (97, 61, 80, 50), (33, 35, 70, 41)):
        img = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        mx, my = _maps(rng, dw, dh, sw, sh, kind)
        assert np.array_equal(oracle.remap(img, mx, my), np_remap(img, mx, my)), (kind, sw, sh)


def test_oracle_clahe_matches_numpy_model(oracle):
    rng = np.random.default_rng(6)
    cases = [(synth.mono_frame(512, 512, 3), 3.0, (8, 8)), (synth.mono_frame(376, 240, 4), 3.0, (8, 8)),
             (synth.mono_frame(333, 247, 5), 3.0, (8, 8)),  # neither axis divides: reflect-101 extension
             (synth.mono_frame(320, 243, 6), 2.0, (4, 6)),  # only the rows do not divide (both axes are extended)
             (rng.integers(0, 256, (96, 128), dtype=np.uint8), 40.0, (8, 8)),
             (rng.integers(100, 110, (64, 64), dtype=np.uint8), 0.0, (2, 2)),  # clip 0: plain tile equalisation
             ((synth.mono_frame(256, 256, 7) // 8 + 100).astype(np.uint8), 1.0, (8, 8))]  # low contrast: heavy clipping
    for img, clip, tiles in cases:
        assert np.array_equal(oracle.clahe(img, clip, tiles), np_clahe(img, clip, tiles)), (img.shape, clip, tiles)


def test_oracle_clahe_properties(oracle):
    flat = np.full((128, 160), 77, np.uint8)
    out = oracle.clahe(flat, 3.0, (8, 8))
    assert (out == out[0, 0]).all()  # one occupied bin in every tile: every lut maps 77 to the same value
    # no clipping, one tile: global histogram equalisation lut = rne(cdf * 255 / area)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (32, 48), dtype=np.uint8)
    cdf = np.cumsum(np.bincount(img.ravel(), minlength=256)).astype(np.float32)
    lut = np.rint(cdf * (np.float32(255) / np.float32(img.size))).astype(np.uint8)
    assert np.array_equal(oracle.clahe(img, 0.0, (1, 1)), lut[img])
    # monotone: a brighter pixel at the same place never comes out darker (all luts are non-decreasing)
    a = synth.mono_frame(256, 192, 8)
    o = oracle.clahe(a, 3.0, (8, 8))
    b = a.copy()
    b[50, 60] = min(int(a[50, 60]) + 40, 255)
    # the histogram of the tile changes slightly, so compare through the model instead of assuming the same lut
    assert np.array_equal(oracle.clahe(b, 3.0, (8, 8)), np_clahe(b, 3.0, (8, 8)))
    assert o.std() > a.std() * 0.9  # contrast is not destroyed


def test_oracle_reproduces_rectify_clahe_golden(oracle):
    g = np.load(GOLDEN)
    assert np.array_equal(oracle.clahe(g["img"], 3.0, (8, 8)), g["clahe"])
    assert np.array_equal(oracle.clahe(g["img"], 2.0, (4, 3)), g["clahe_4x3_clip2"])
    assert np.array_equal(oracle.remap(g["img"], g["map_x"], g["map_y"]), g["remap"])
    assert np.array_equal(oracle.remap(g["clahe"], g["map_x"], g["map_y"]), g["chain"])
    assert np.array_equal(np_remap(g["img"], g["map_x"], g["map_y"]), g["remap"])  # and the independent model


@pytest.mark.gpu
def test_hip_reproduces_rectify_clahe_golden():
    import orb_slam3_fast_amd as orbx
    g = np.load(GOLDEN)
    assert np.array_equal(orbx.CLAHE(3.0, (8, 8)).apply(g["img"]), g["clahe"])
    assert np.array_equal(orbx.CLAHE(2.0, (4, 3)).apply(g["img"]), g["clahe_4x3_clip2"])
    assert np.array_equal(orbx.remap(g["img"], g["map_x"], g["map_y"]), g["remap"])
    pp = orbx.Preproc(136, 100, maps=(g["map_x"], g["map_y"]), clahe=(3.0, (8, 8)), max_batch=1)
    assert np.array_equal(pp.run(g["img"]), g["chain"])


# ---- HIP vs oracle ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rectify", "random", "grid32", "special"])
def test_hip_remap_matches_oracle(oracle, kind):
    import orb_slam3_fast_amd as orbx
    rng = np.random.default_rng(11)
    for (sw, sh, dw, dh) in ((752, 480, 752, 480), (640, 480, 600, 350), (97, 61, 83, 50), (1280, 720, 1241, 703)):
        img = synth.mono_frame(sw, sh, 9) if sw >= 512 else rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        mx, my = _maps(rng, dw, dh, sw, sh, kind)
        assert np.array_equal(orbx.remap(img, mx, my), oracle.remap(img, mx, my)), (kind, sw, sh, dw, dh)
    # interleaved channels: cv::remap works per channel
    col = rng.integers(0, 256, (61, 97, 3), dtype=np.uint8)
    mx, my = _maps(rng, 83, 50, 97, 61, kind)
    got = orbx.remap(col, mx, my)
    for c in range(3):
        assert np.array_equal(got[..., c], oracle.remap(np.ascontiguousarray(col[..., c]), mx, my))


@pytest.mark.gpu
def test_hip_preproc_remap_forms_match_oracle(oracle):
    """The plans' two cv::remap forms (round 6: source footprints staged through LDS, k_remap_lds / per-thread window gathers,
    k_remap1) against the oracle: rectification maps of several sizes (tile edges: widths that are no multiple of 128, heights
    that are no multiple of 8), two maps, batches whose per-map image count is odd, even, below and above one group of eight;
    maps with seams and taps outside the source (the byte-read branch, zero weights); maps whose footprints do not fit (random,
    a strong rotation: the plan keeps k_remap1 whatever the hook says) -- every result equal to the oracle's."""
    import ctypes as C
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd import hipmem
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    rng = np.random.default_rng(66)

    def seam_maps(dw, dh, sw, sh):   # a rectification map with a vertical seam, a shifted band and a region far outside the source
        mx, my = synth.rectify_maps(dw, dh, sw, sh, seed=5)
        mx, my = mx.copy(), my.copy()
        mx[:, dw // 2:] += 9.25
        my[dh // 3:dh // 3 + 5, :] += 3.5
        mx[:7, :40] = -20.0
        my[-6:, -50:] = sh + 30.0
        mx[10:14, 60:64] = sw - 1.25       # the pair straddles the right edge
        my[20:23, 10:20] = -0.75           # and the top edge
        return mx, my

    cases = [((752, 480), (720, 460), "rectify", 2, (1, 2, 3, 4, 9, 17, 18)), ((1280, 720), (1280, 720), "rectify", 2, (2, 5)),
             ((640, 480), (601, 353), "rectify", 1, (1, 8, 9)), ((512, 512), (512, 512), "seam", 2, (4, 7)),
             ((256, 64), (130, 9), "rectify", 1, (3,)), ((640, 480), (600, 350), "random", 1, (2,)),
             ((752, 480), (720, 460), "rotate", 2, (3,))]
    try:
        for (sw, sh), (dw, dh), kind, nmaps, batches in cases:
            maps = []
            for m in range(nmaps):
                if kind == "rectify":
                    maps.append(synth.rectify_maps(dw, dh, sw, sh, seed=10 + m))
                elif kind == "seam":
                    mx, my = seam_maps(dw, dh, sw, sh)
                    maps.append((mx + m, my))
                elif kind == "rotate":
                    maps.append(synth.rectify_maps(dw, dh, sw, sh, seed=3, rot_deg=(2.0, -1.0, 25.0 + m)))
                else:
                    maps.append(_maps(rng, dw, dh, sw, sh, "random"))
            mapsx, mapsy = np.stack([a for a, _ in maps]), np.stack([b for _, b in maps])
            pp = orbx.Preproc(sw, sh, channels=1, maps=(mapsx, mapsy), max_batch=max(batches))
            for n in batches:
                frames = np.stack([synth.mono_frame(sw, sh, 100 + i) if sw >= 512 else rng.integers(0, 256, (sh, sw), dtype=np.uint8)
                                   for i in range(min(n, 3))])
                frames = np.ascontiguousarray(frames[np.arange(n) % len(frames)])
                frames[:, ::7, ::5] = rng.integers(0, 256, frames[:, ::7, ::5].shape, dtype=np.uint8)
                raw = DeviceBuffer.from_numpy(frames)
                want = [oracle.remap(frames[i], mapsx[i % nmaps], mapsy[i % nmaps]) for i in range(n)]
                for hook in (1, 0):
                    orbx.lib().orbx_debug_set_remap_lds(hook)
                    ptr, w, h, rp, ip = pp.run_device(raw.ptr.value, n, sw, sw * sh)
                    got = np.zeros((n, ip), np.uint8)
                    hipmem._ck(hipmem.hip().hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), got.nbytes, 2))
                    got = got[:, :h * rp].reshape(n, h, rp)[:, :, :w]
                    for i in range(n):
                        assert np.array_equal(got[i], want[i]), (kind, (sw, sh), (dw, dh), n, hook, i)
    finally:
        orbx.lib().orbx_debug_set_remap_lds(1)


@pytest.mark.gpu
def test_hip_clahe_matches_oracle(oracle):
    import orb_slam3_fast_amd as orbx
    rng = np.random.default_rng(12)
    cases = [(synth.mono_frame(512, 512, 3), 3.0, (8, 8)), (synth.mono_frame(752, 480, 4), 3.0, (8, 8)),
             (synth.mono_frame(333, 247, 5), 3.0, (8, 8)), (synth.mono_frame(320, 243, 6), 2.0, (4, 6)),
             (rng.integers(0, 256, (96, 130), dtype=np.uint8), 40.0, (8, 8)),
             (rng.integers(100, 110, (64, 64), dtype=np.uint8), 0.0, (2, 2)), (np.full((128, 160), 77, np.uint8), 3.0, (8, 8)),
             ((synth.mono_frame(1280, 720, 7) // 8 + 100).astype(np.uint8), 1.0, (8, 8))]
    for img, clip, tiles in cases:
        got = orbx.CLAHE(clip, tiles).apply(img)
        assert np.array_equal(got, oracle.clahe(img, clip, tiles)), (img.shape, clip, tiles)
    # both apply forms (round 6: one workgroup per interpolation cell with the table in LDS / per-pixel table gathers), incl. tile
    # sizes whose cell boundaries are not exact in float (96, 160 px: the host falls back unless the float rule agrees)
    cases2 = [cases[0], cases[-1], (synth.mono_frame(768, 512, 8), 3.0, (8, 8)), (synth.mono_frame(1024, 768, 9), 2.0, (4, 4)),
              (synth.mono_frame(256, 256, 10), 3.0, (8, 8))]
    try:
        for hook in (0, 1):
            orbx.lib().orbx_debug_set_clahe_cell_kernel(hook)
            for img, clip, tiles in cases2:
                assert np.array_equal(orbx.CLAHE(clip, tiles).apply(img), oracle.clahe(img, clip, tiles)), (hook, img.shape, clip, tiles)
    finally:
        orbx.lib().orbx_debug_set_clahe_cell_kernel(1)
    with pytest.raises(orbx.OrbxError):
        orbx.CLAHE(3.0, (0, 8)).apply(np.zeros((32, 32), np.uint8))
    with pytest.raises(orbx.OrbxError):
        orbx.CLAHE(3.0, (8, 8)).apply(np.zeros((8, 8), np.uint8))  # fewer pixels than tiles along an axis


@pytest.mark.gpu
def test_hip_preproc_chain_and_raw_extraction(oracle):
    """Raw frames -> CLAHE -> rectification on the device -> extractor, against the oracle applied stage by stage."""
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    sw, sh, dw, dh = 752, 480, 720, 460
    L, R = synth.stereo_pair(sw, sh, 21)
    ml, mr = synth.rectify_maps(dw, dh, sw, sh, seed=1), synth.rectify_maps(dw, dh, sw, sh, seed=2, rot_deg=(-0.3, 0.5, -0.2))
    maps = (np.stack([ml[0], mr[0]]), np.stack([ml[1], mr[1]]))
    pp = orbx.Preproc(sw, sh, channels=1, maps=maps, clahe=(3.0, (8, 8)), max_batch=4)
    assert (pp.out_w, pp.out_h) == (dw, dh)
    want = [oracle.remap(oracle.clahe(L), *ml), oracle.remap(oracle.clahe(R), *mr)]
    assert np.array_equal(pp.run(L, 0), want[0]) and np.array_equal(pp.run(R, 1), want[1])
    # device-resident batch of two stereo pairs: frames L R L R use maps 0 1 0 1
    raw = DeviceBuffer.from_numpy(np.stack([L, R, R, L]))
    ptr, w, h, rp, ip = pp.run_device(raw.ptr.value, 4, sw, sw * sh)
    assert (w, h) == (dw, dh)
    ex = orbx.ORBextractor(1000, 1.2, 8, 20, 7, max_width=dw, max_height=dh, max_batch=4)
    ex.extract_batch_raw_device(pp, raw.ptr.value, 4, sw, sw * sh)
    ex.sync()
    want4 = want + [oracle.remap(oracle.clahe(R), *ml), oracle.remap(oracle.clahe(L), *mr)]
    oex = oracle.OracleExtractor(1000)
    for i in range(4):
        assert np.array_equal(ex.image_pyramid(0, image=i), want4[i]), i  # level 0 IS the pre-processor's output
        mono, k, d = ex.download(i)
        om, ok_, od = oex.extract(want4[i])
        assert mono == om and k.tobytes() == ok_.tobytes() and np.array_equal(d, od), i
    # colour frames: resize (no maps) then gray, the System::TrackStereo / GrabImageStereo order
    col = np.stack([L, np.roll(L, 2, 1), (L // 2 + 50).astype(np.uint8)], 2)
    pc = orbx.Preproc(sw, sh, channels=3, rgb=False, out_size=(600, 384), max_batch=1)
    assert np.array_equal(pc.run(col), oracle.cvt_gray(oracle.resize_c(col, 600, 384), rgb=False))
    # colour frames through the rectification: cv::remap works per interleaved channel, GrabImageStereo converts afterwards
    pr = orbx.Preproc(sw, sh, channels=3, rgb=True, maps=maps, max_batch=2)
    for eye, (mx, my) in enumerate((ml, mr)):
        planes = np.stack([oracle.remap(np.ascontiguousarray(col[..., c]), mx, my) for c in range(3)], 2)
        assert np.array_equal(pr.run(col, eye), oracle.cvt_gray(planes, rgb=True)), eye
    # nothing enabled: pass-through
    p0 = orbx.Preproc(sw, sh, max_batch=1)
    assert np.array_equal(p0.run(L), L)
    with pytest.raises(orbx.OrbxError):
        orbx.Preproc(sw, sh, channels=3, clahe=(3.0, (8, 8)))  # cv::CLAHE is single-channel
