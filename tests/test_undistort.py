"""Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:853-919) = cv::undistortPoints with P = K.

The oracle restates OpenCV's iteration (5 fixed-point steps in double) from the published algorithm; OpenCV is not in
this image, so that restatement is unpinned like the other OpenCV kernels.  Device and oracle run the
same IEEE double sequence without contraction: the comparison is exact."""
import numpy as np
import pytest

EUROC_K = (458.654, 457.296, 367.215, 248.375)                       # Examples/Monocular/EuRoC.yaml
EUROC_D = (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05)
TUM_K = (517.306408, 516.469215, 318.643040, 255.313989)             # Examples/RGB-D/TUM1.yaml (5 coefficients)
TUM_D = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)


def _kps(mod, n, w, h, seed):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, mod.KP_DTYPE)
    k["x"], k["y"] = rng.uniform(0, w, n), rng.uniform(0, h, n)
    k["size"], k["angle"], k["response"], k["octave"], k["class_id"] = 31, rng.uniform(0, 360, n), 40, rng.integers(0, 8, n), -1
    return k


def _distort(K, D, x, y):
    """OpenCV forward model (radial k1 k2 k3 + tangential p1 p2) in float64."""
    k1, k2, p1, p2 = D[:4]
    k3 = D[4] if len(D) > 4 else 0.0
    xn, yn = (x - K[2]) / K[0], (y - K[3]) / K[1]
    r2 = xn * xn + yn * yn
    cd = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = xn * cd + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * cd + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return xd * K[0] + K[2], yd * K[1] + K[3]


@pytest.mark.parametrize("K,D,w,h", [(EUROC_K, EUROC_D, 752, 480), (TUM_K, TUM_D, 640, 480)])
def test_oracle_undistort_inverts_the_distortion_model(oracle, K, D, w, h):
    k = _kps(oracle, 500, w, h, 1)
    u = oracle.undistort_keypoints(k, K, D)
    # only pt changes; the principal point is a fixed point; re-distorting lands on the input (5 iterations: sub-pixel)
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(u[f], k[f])
    xd, yd = _distort(K, D, u["x"].astype(np.float64), u["y"].astype(np.float64))
    inner = (np.abs(k["x"] - K[2]) < 0.35 * w) & (np.abs(k["y"] - K[3]) < 0.35 * h)
    assert np.abs(xd - k["x"])[inner].max() < 0.1 and np.abs(yd - k["y"])[inner].max() < 0.1
    assert np.abs(xd - k["x"]).max() < 1.0 and np.abs(yd - k["y"]).max() < 1.0
    c = np.zeros(1, oracle.KP_DTYPE)
    c["x"], c["y"] = K[2], K[3]
    uc = oracle.undistort_keypoints(c, K, D)
    assert abs(uc["x"][0] - np.float32(K[2])) < 1e-4 and abs(uc["y"][0] - np.float32(K[3])) < 1e-4


def test_oracle_undistort_identity_and_bounds(oracle):
    k = _kps(oracle, 50, 752, 480, 2)
    assert oracle.undistort_keypoints(k, EUROC_K, (0.0, 0.1, 0.0, 0.0)).tobytes() == k.tobytes()  # mDistCoef(0) == 0: copy
    assert oracle.image_bounds(752, 480, EUROC_K, (0.0, 0.0, 0.0, 0.0)).tolist() == [0.0, 0.0, 752.0, 480.0]
    b = oracle.image_bounds(752, 480, EUROC_K, EUROC_D)   # barrel distortion: the undistorted corners lie outside
    assert b[0] < -100 and b[1] < -60 and b[2] > 850 and b[3] > 540
    corners = np.zeros(4, oracle.KP_DTYPE)
    corners["x"], corners["y"] = [0, 752, 0, 752], [0, 0, 480, 480]
    u = oracle.undistort_keypoints(corners, EUROC_K, EUROC_D)
    assert b.tolist() == [min(u["x"][0], u["x"][2]), min(u["y"][0], u["y"][1]), max(u["x"][1], u["x"][3]), max(u["y"][2], u["y"][3])]


@pytest.mark.gpu
@pytest.mark.parametrize("K,D,w,h", [(EUROC_K, EUROC_D, 752, 480), (TUM_K, TUM_D, 640, 480),
                                     (TUM_K, TUM_D + (0.01, -0.02, 0.003), 640, 480)])
def test_hip_undistort_matches_oracle_exactly(oracle, K, D, w, h):
    import orb_slam3_fast_amd as orbx
    k = _kps(orbx, 3000, w, h, 3)
    assert orbx.UndistortKeyPoints(k, K, D).tobytes() == oracle.undistort_keypoints(k, K, D).tobytes()
    assert orbx.ComputeImageBounds(w, h, K, D).tobytes() == oracle.image_bounds(w, h, K, D).tobytes()


@pytest.mark.gpu
def test_hip_undistort_edge_cases():
    import orb_slam3_fast_amd as orbx
    k = _kps(orbx, 20, 640, 480, 4)
    assert orbx.UndistortKeyPoints(k, TUM_K, (0.0, 0.2, 0.0, 0.0)).tobytes() == k.tobytes()
    assert orbx.UndistortKeyPoints(k, TUM_K, ()).tobytes() == k.tobytes()
    assert len(orbx.UndistortKeyPoints(k[:0], TUM_K, TUM_D)) == 0
    assert orbx.ComputeImageBounds(640, 480, TUM_K, (0.0,) * 4).tolist() == [0.0, 0.0, 640.0, 480.0]
    with pytest.raises(orbx.OrbxError):
        orbx.UndistortKeyPoints(k, (0.0, 1.0, 0.0, 0.0), TUM_D)
    with pytest.raises(orbx.OrbxError):   # tilted sensor model
        orbx.UndistortKeyPoints(k, TUM_K, (0.1,) * 12 + (0.01, 0.0))


GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "undistort.npz")


def _golden_cases(mod):
    g = np.load(GOLDEN)
    for tag in ("euroc", "tum1"):
        k = np.ascontiguousarray(g[tag + "_kps"]).view(mod.KP_DTYPE).reshape(-1)
        yield g[tag + "_K"], g[tag + "_D"], [int(v) for v in g[tag + "_size"]], k, g[tag + "_un"], g[tag + "_bounds"]


def test_oracle_reproduces_undistort_golden(oracle):
    for K, D, (w, h), k, un, bounds in _golden_cases(oracle):
        assert np.array_equal(oracle.undistort_keypoints(k, K, D).view(np.uint8).reshape(-1, 28), un)
        assert oracle.image_bounds(w, h, K, D).tobytes() == bounds.tobytes()


@pytest.mark.gpu
def test_hip_reproduces_undistort_golden():
    import orb_slam3_fast_amd as orbx
    for K, D, (w, h), k, un, bounds in _golden_cases(orbx):
        assert np.array_equal(orbx.UndistortKeyPoints(k, K, D).view(np.uint8).reshape(-1, 28), un)
        assert orbx.ComputeImageBounds(w, h, K, D).tobytes() == bounds.tobytes()
