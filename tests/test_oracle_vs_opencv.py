"""The oracle's restatements of OpenCV ("[OCV-mem]") against REAL OpenCV outputs.

The fixture tests/golden/opencv_pins.npz is produced by `python tools/gen_golden_opencv.py` on a box that has cv2
(this image does not: no network) -- the tests skip while it is absent.  With it, every rule the oracle restates from
memory is compared bit for bit (float results to the tolerance written here), and the GPU-marked twin runs the HIP
path against the same OpenCV outputs without the oracle."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")
FIX = os.path.join(G, "opencv_pins.npz")
# ORBX_REQUIRE_OPENCV_PINS=1 (CI of a box that is SUPPOSED to hold the fixture): an absent fixture FAILS every test below instead
# of skipping it -- "parity unpinned" must not pass silently where a pin is expected.
REQUIRE = os.environ.get("ORBX_REQUIRE_OPENCV_PINS", "0") == "1"
pytestmark = pytest.mark.skipif(not os.path.exists(FIX) and not REQUIRE, reason="tests/golden/opencv_pins.npz absent: neither the build "
                                "container nor the MI355X box holds OpenCV or reaches a package index "
                                "(profiles/r3_opencv_probe_{buildbox,gpubox}.txt) -- PARITY UNPINNED for the OpenCV "
                                "fixed-point kernels; tools/gen_golden_opencv.py writes the fixture on any box with "
                                "opencv-python >= 4.5.1")


@pytest.fixture(scope="module")
def cvp():
    if not os.path.exists(FIX):
        pytest.fail("ORBX_REQUIRE_OPENCV_PINS=1 but tests/golden/opencv_pins.npz is absent: run `python tools/gen_golden_opencv.py` "
                    "on a box with opencv-python >= 4.5.1 and commit the fixture (PARITY UNPINNED until then)")
    return np.load(FIX)


@pytest.fixture(scope="module")
def imgs():
    pin = np.load(os.path.join(G, "fast9_skimage.npz"))
    return {n: pin["img_" + n] for n in ("g384", "g400L", "g160")}


def _cv_version(cvp):
    return tuple(int(x) for x in str(cvp["cv_version"]).split(".")[:3])


def _blur_variants(cvp):
    """The oracle blur models that can describe the fixture's OpenCV: >= 4.5.1 has ONE (451); 4.0 .. 4.5.0 has the 257-sum taps
    through the scalar path (440) or with the flooring 16- / 32-lane vector body (44016 / 44032, orb_oracle.cpp gaussian_blur7)."""
    return (451,) if _cv_version(cvp) >= (4, 5, 1) else (44032, 44016, 440)


def test_opencv_version(cvp):
    v = _cv_version(cvp)
    assert v >= (4, 0, 0), "OpenCV 3.x blurs CV_8U through the float filter: not modelled"
    assert not bool(cvp["use_ipp"]), "IPP was active: ippiResizeLinear differs from the generic path (SURVEY B2)"


def test_resize(oracle, cvp, imgs):      # src/ORBextractor.cc:1122
    for n, im in imgs.items():
        cur, l = im, 1
        while "resize_%s_L%d" % (n, l) in cvp.files:
            ref = cvp["resize_%s_L%d" % (n, l)]
            cur = oracle.resize(cur, ref.shape[1], ref.shape[0])
            assert np.array_equal(cur, ref), (n, l, int((cur != ref).sum()))
            l += 1
        assert l > 2
    for key in ("resize_g384_to_251x97", "resize_g384_half", "resize_g384_up"):
        ref = cvp[key]
        assert np.array_equal(oracle.resize(imgs["g384"], ref.shape[1], ref.shape[0]), ref), key


def test_fast(oracle, cvp, imgs):        # src/ORBextractor.cc:810-826
    for n, im in imgs.items():
        for t in (20, 7):
            ref = cvp["fast_%s_t%d" % (n, t)]
            got = oracle.fast(im, t, nms=True).astype(np.float32)
            assert got.shape == ref.shape and np.array_equal(got, ref), (n, t)       # x, y, response, in cv order
            raw = cvp["fastraw_%s_t%d" % (n, t)]
            assert np.array_equal(oracle.fast(im, t, nms=False)[:, :2].astype(np.float32), raw), (n, t)
    big = imgs["g400L"]
    for i, (y, x, hh, ww) in enumerate(cvp["fastroi_rects"]):
        roi = big[y:y + hh, x:x + ww]
        assert np.array_equal(oracle.fast(roi, 20, nms=True).astype(np.float32), cvp["fastroi_%d" % i]), i


def test_blur(oracle, cvp, imgs):        # src/ORBextractor.cc:1075
    """Exactly which oracle model reproduces the fixture: the answer for a 4.0 .. 4.5.0 fixture is the value to hand to
    orbx_set_opencv_compat / oro_set_blur_taps for a reference built against that OpenCV."""
    imp = np.zeros((15, 15), np.uint8)
    imp[7, 7] = 255
    match = [v for v in _blur_variants(cvp)
             if all(np.array_equal(oracle.blur(im, v), cvp["blur_" + n]) for n, im in imgs.items())
             and np.array_equal(oracle.blur(imp, v), cvp["blur_impulse"])]
    print("OpenCV %s GaussianBlur == oracle blur variant(s) %s" % (str(cvp["cv_version"]), match))
    assert match, "no oracle blur model reproduces cv2.GaussianBlur of OpenCV %s" % str(cvp["cv_version"])


def test_fast_atan2(oracle, cvp):        # src/ORBextractor.cc:98
    got = np.array([oracle.fast_atan2(y, x) for y, x in cvp["atan2_yx"]], np.float32)
    assert np.array_equal(got.view(np.uint32), cvp["atan2_deg"].view(np.uint32))


def test_knn(oracle, cvp):               # src/Frame.cc:46,1293
    st = np.load(os.path.join(G, "stereo_400x300.npz"))
    idx, dist, _ = oracle.bf_knn2(st["dL"], st["dR"])
    assert np.array_equal(idx, cvp["knn_idx"]) and np.array_equal(dist.astype(np.float32), cvp["knn_dist"])
    idx, _, _ = oracle.bf_knn2(st["dL"][:60], cvp["knn_tie_train"])
    assert np.array_equal(idx, cvp["knn_tie_idx"])


def test_remap_clahe_gray(oracle, cvp, imgs):   # src/System.cc:294, stereo_tum_vi.cc:142, src/Tracking.cc:1394
    rc = np.load(os.path.join(G, "rectify_clahe.npz"))
    assert np.array_equal(oracle.remap(rc["img"], rc["map_x"], rc["map_y"]), cvp["remap"])
    assert np.array_equal(oracle.clahe(rc["img"], 3.0, (8, 8)), cvp["clahe_3_8x8"])
    assert np.array_equal(oracle.clahe(rc["img"], 2.0, (4, 3)), cvp["clahe_2_4x3"])
    assert np.array_equal(oracle.clahe(imgs["g384"], 3.0, (8, 8)), cvp["clahe_g384"])
    assert np.array_equal(oracle.cvt_gray(cvp["gray_in"], rgb=True), cvp["gray_rgb"])
    assert np.array_equal(oracle.cvt_gray(cvp["gray_in"], rgb=False), cvp["gray_bgr"])
    assert np.array_equal(oracle.cvt_gray(cvp["gray4_in"], rgb=True), cvp["gray_rgba"])
    assert np.array_equal(oracle.cvt_gray(cvp["gray4_in"], rgb=False), cvp["gray_bgra"])


def test_undistort(oracle, cvp):         # src/Frame.cc:869
    ud = np.load(os.path.join(G, "undistort.npz"))
    for rig in ("euroc", "tum1"):
        kp = ud[rig + "_kps"].copy().view(oracle.KP_DTYPE).reshape(-1)
        un = oracle.undistort_keypoints(kp, ud[rig + "_K"], ud[rig + "_D"])
        got = np.stack([un["x"], un["y"]], 1)
        # float arithmetic inside cv::undistortPoints (double iterations, float I/O): agreement to 1e-3 px
        assert np.abs(got - cvp["undistort_" + rig]).max() < 1e-3


@pytest.mark.gpu
def test_hip_path_against_opencv(cvp, imgs):
    """The HIP kernels against OpenCV's outputs directly (no oracle): pyramid chain, blurred levels, pre-processing."""
    import orb_slam3_fast_amd as orbx
    if orbx.device_count() < 1:
        pytest.fail("no HIP device visible")
    for n, im in imgs.items():
        nl = 3 if n == "g160" else 8
        ex = orbx.ORBextractor(500, 1.2, nl, 20, 7, max_width=im.shape[1], max_height=im.shape[0])
        ex(im, (0, 0))
        for l in range(1, nl):
            assert np.array_equal(ex.image_pyramid(l), cvp["resize_%s_L%d" % (n, l)]), (n, l)
        hit = []
        for v in _blur_variants(cvp):          # (the variant test_blur reports for this OpenCV)
            ex.set_opencv_compat(v)
            ex(im, (0, 0))
            hit += [v] if np.array_equal(ex.image_pyramid(0, blurred=True), cvp["blur_" + n]) else []
        assert hit, n
    rc = np.load(os.path.join(G, "rectify_clahe.npz"))
    assert np.array_equal(orbx.remap(rc["img"], rc["map_x"], rc["map_y"]), cvp["remap"])
    assert np.array_equal(orbx.CLAHE(3.0, (8, 8)).apply(rc["img"]), cvp["clahe_3_8x8"])
    assert np.array_equal(orbx.cvtColorGray(cvp["gray_in"], rgb=True), cvp["gray_rgb"])
    st = np.load(os.path.join(G, "stereo_400x300.npz"))
    idx, dist, _ = orbx.bf_knn2(st["dL"], st["dR"])
    assert np.array_equal(idx, cvp["knn_idx"])
