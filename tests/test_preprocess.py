"""Image pre-processing in front of the extractor (the part that is pinned by published constants):
cv::cvtColor(..., *2GRAY) (src/Tracking.cc:1394-1412) and cv::resize(..., INTER_LINEAR) of the input image
(src/System.cc:297-298).  Integer arithmetic: device == oracle exactly.  (cv::remap rectification and CLAHE:
tests/test_rectify_clahe.py.)"""
import numpy as np
import pytest


def _img(rng, h, w, cn):
    base = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1, cn), dtype=np.uint8)
    img = np.kron(base, np.ones((8, 8, 1), np.uint8))[:h, :w]
    return np.ascontiguousarray((img.astype(np.int16) + rng.integers(-20, 21, img.shape)).clip(0, 255).astype(np.uint8))


def test_oracle_gray_formula(oracle):
    rng = np.random.default_rng(1)
    img = _img(rng, 37, 53, 3)
    ref = ((img[..., 0].astype(np.int64) * 9798 + img[..., 1].astype(np.int64) * 19235 + img[..., 2].astype(np.int64) * 3735 + 16384)
           >> 15).astype(np.uint8)
    assert np.array_equal(oracle.cvt_gray(img, rgb=True), ref)
    assert np.array_equal(oracle.cvt_gray(img[..., ::-1], rgb=False), ref)
    rgba = np.concatenate([img, rng.integers(0, 256, (37, 53, 1), dtype=np.uint8)], 2)
    assert np.array_equal(oracle.cvt_gray(rgba, rgb=True), ref)
    # known answers: white, black, pure R / G / B (0.299, 0.587, 0.114 in 15-bit fixed point, sum exactly 1)
    for px, v in (((255, 255, 255), 255), ((0, 0, 0), 0), ((255, 0, 0), 76), ((0, 255, 0), 150), ((0, 0, 255), 29), ((128, 128, 128), 128)):
        assert oracle.cvt_gray(np.array([[px]], np.uint8))[0, 0] == v
    # the 14-bit constants of OpenCV < 3.4.2 differ in places: the variant switch is real
    assert not np.array_equal(oracle.cvt_gray(img, True, 14), ref)


def test_oracle_multichannel_resize_is_per_channel(oracle):
    rng = np.random.default_rng(2)
    img = _img(rng, 96, 128, 3)
    for dw, dh in ((100, 70), (64, 48), (200, 150), (128, 96)):
        r3 = oracle.resize_c(img, dw, dh)
        planes = np.stack([oracle.resize(np.ascontiguousarray(img[..., c]), dw, dh) for c in range(3)], 2)
        assert np.array_equal(r3, planes)
    assert np.array_equal(oracle.resize_c(img, 128, 96), img)  # identity size: fx == 0 everywhere


@pytest.mark.gpu
def test_hip_preprocess_matches_oracle(oracle):
    import orb_slam3_fast_amd as orbx
    rng = np.random.default_rng(3)
    for (h, w, cn) in ((480, 640, 3), (350, 601, 4), (33, 47, 3), (720, 1280, 3), (720, 1280, 4), (64, 1296, 3), (5, 16, 4)):   # (round 6: 16-pixel streaming form + row tails)
        img = _img(rng, h, w, cn)
        for rgb in (True, False):
            assert np.array_equal(orbx.cvtColorGray(img, rgb), oracle.cvt_gray(img, rgb))
    for (h, w, cn, dw, dh) in ((480, 752, 1, 600, 350), (480, 640, 3, 512, 384), (720, 1280, 1, 640, 360), (100, 90, 4, 171, 203)):
        img = _img(rng, h, w, cn) if cn > 1 else _img(rng, h, w, 1)[..., 0].copy()
        assert np.array_equal(orbx.resize(img, dw, dh), oracle.resize_c(img, dw, dh))
    # strided input (a view into a wider buffer), as cv::Mat ROIs are
    wide = _img(rng, 120, 400, 3)
    view = wide[:, 40:300]
    assert np.array_equal(orbx.cvtColorGray(np.ascontiguousarray(view)), oracle.cvt_gray(np.ascontiguousarray(view)))
    with pytest.raises(orbx.OrbxError):
        orbx.cvtColorGray(np.zeros((4, 4, 2), np.uint8))


@pytest.mark.gpu
def test_hip_plan_resize_of_mono_frames_matches_oracle(oracle):
    """System::TrackStereo's input resize (src/System.cc:297-298) in a pre-processing plan, single-channel frames: since round 6 the
    pyramid's tiled cv::resize kernel on a two-level geometry (launch_resize_plain) instead of the per-pixel kernel.  Scale factors
    from 0.8 (enlarging) to 3 (footprints of 50 rows), sizes that are no multiples of the 256 x 16 tile, the EuRoC pair
    752x480 -> 600x350, batches of one and several frames, a source whose last dword is partial -- all equal to the oracle's
    cv::resize restatement."""
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    rng = np.random.default_rng(8)
    for (w, h, dw, dh, n) in ((752, 480, 600, 350, 3), (1280, 720, 640, 360, 2), (640, 480, 800, 600, 1), (1281, 721, 427, 241, 2),
                              (600, 350, 257, 17, 1), (333, 222, 111, 74, 4), (512, 512, 500, 100, 1), (97, 61, 83, 50, 2)):
        frames = np.stack([_img(rng, h, w, 1)[..., 0] for _ in range(n)])
        dev = DeviceBuffer.from_numpy(frames)
        pp = orbx.Preproc(w, h, channels=1, out_size=(dw, dh), max_batch=n)
        assert (pp.out_w, pp.out_h) == (dw, dh)
        for i in range(n):
            assert np.array_equal(pp.run(frames[i]), oracle.resize(frames[i], dw, dh)), (w, h, dw, dh, i)
        ptr, ow, oh, rp, ip = pp.run_device(dev.ptr.value, n, w, w * h)
        got = np.zeros((n, ip), np.uint8)
        import ctypes as C
        from orb_slam3_fast_amd import hipmem
        hipmem._ck(hipmem.hip().hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), got.nbytes, 2))
        got = got[:, :oh * rp].reshape(n, oh, rp)[:, :, :ow]
        for i in range(n):
            assert np.array_equal(got[i], oracle.resize(frames[i], dw, dh)), (w, h, dw, dh, i)


@pytest.mark.gpu
def test_hip_color_frame_to_keypoints_flow(oracle):
    """TUM-like flow: colour frame -> gray (mbRGB) -> resize to the settings' size -> ORBextractor."""
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd import synth
    g = synth.mono_frame(640, 480, 33)
    rng = np.random.default_rng(4)
    col = np.stack([g, np.roll(g, 3, 1), (g // 2 + 60).astype(np.uint8)], 2)
    gray = orbx.resize(orbx.cvtColorGray(col, True), 600, 450)
    ogray = oracle.resize_c(oracle.cvt_gray(col, True), 600, 450)
    assert np.array_equal(gray, ogray)
    ex = orbx.ORBextractor(800, 1.2, 8, 20, 7, max_width=600, max_height=450)
    mono, k, d = ex(gray)
    om, ok_, od = oracle.OracleExtractor(800).extract(ogray)
    assert mono == om and k.tobytes() == ok_.tobytes() and np.array_equal(d, od)
