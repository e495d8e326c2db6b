"""Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1273-1331) + KannalaBrandt8::TriangulateMatches
(src/CameraModels/KannalaBrandt8.cpp:341-432): the floating-point routine of the path.

Integer results (2-NN indices, ratio test, descMatches) are compared exactly.  The float results are compared with a
stated tolerance, because (a) the device libm (atan2f / tanf / cosf / sinf) differs from the host's by a few ulp and
(b) neither side can run Eigen::JacobiSVD<Matrix4f> (Eigen is not in this image; both use a one-sided Jacobi SVD in
double):
    accepted depth / 3-D point : |hip - oracle| <= REL_TOL * |oracle|   (REL_TOL = 2e-4; the triangulation amplifies
                                 a 1-ulp ray difference by up to 1 / parallax angle ~ 50x at the 0.9998 gate)
    accept / reject decision   : must agree unless the oracle's gated quantity lies within GATE_TOL (1e-3 relative)
                                 of its threshold, in which case either outcome is float noise.
"""
import os

import numpy as np
import pytest

from orb_slam3_fast_amd import synth

REL_TOL = 2e-4
GATE_TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "fisheye_stereo.npz")


def _rig(mod, sc):
    return mod.kb8_rig(sc["cam1"], sc["cam2"], sc["R12"], sc["t12"])


def _borderline(g):
    """g = oracle gate record {cosParallax, z1, z2, err1/thr1, err2/thr2, return value} of one ratio-test survivor."""
    cosp, z1, z2, e1, e2, ret = (float(v) for v in g)
    near = [abs(cosp - 0.9998) < GATE_TOL * 1e-2]  # cos lives near 1: 1e-5 absolute
    if not np.isnan(z1):
        near.append(abs(z1) < 1e-3)  # cheirality and the depth > 1e-4 acceptance
    if not np.isnan(z2):
        near.append(abs(z2) < 1e-3)
    if not np.isnan(e1):
        near.append(abs(e1 - 1.0) < 10 * GATE_TOL)
    if not np.isnan(e2):
        near.append(abs(e2 - 1.0) < 10 * GATE_TOL)
    return any(near)


def compare_float_results(hip, ora, gates, mono_left):
    """hip / ora = (n, nd, l2r, r2l, depth, p3d).  Returns the number of borderline decisions skipped."""
    hn, hnd, hl2r, hr2l, hdep, hpts = hip
    on, ond, ol2r, or2l, odep, opts = ora
    assert hnd == ond, "ratio-test survivors are integer work: exact"
    skipped = 0
    exp_r2l = np.full(len(or2l), -1, np.int32)
    for i in range(len(ol2r)):
        if np.isnan(gates[i, 5]):
            assert hl2r[i] == -1 and hdep[i] == -1.0
            continue
        if (hl2r[i] >= 0) != (ol2r[i] >= 0):
            assert _borderline(gates[i]), "decision differs away from every gate: keypoint %d gates %s" % (i, gates[i])
            skipped += 1
        elif ol2r[i] >= 0:
            assert hl2r[i] == ol2r[i]
            assert abs(hdep[i] - odep[i]) <= REL_TOL * abs(odep[i])
            assert np.all(np.abs(hpts[i] - opts[i]) <= REL_TOL * np.linalg.norm(opts[i]))
        else:
            assert hdep[i] == -1.0 and not hpts[i].any()
        if hl2r[i] >= 0:
            exp_r2l[hl2r[i]] = i  # ascending i: the last claimant stays, like the serial loop
    assert np.array_equal(hr2l, exp_r2l)
    assert hn == int((hl2r >= 0).sum())
    return skipped


# ---------------------------------------------------------------------------------------------- oracle (CPU)
def test_kb8_project_unproject_round_trip(oracle):
    rng = np.random.default_rng(5)
    for cam in (synth.TUMVI_CAM1, synth.TUMVI_CAM2):
        for _ in range(200):
            rad, az = rng.uniform(0, 220), rng.uniform(0, 2 * np.pi)  # inside the image circle (theta < 1.2 rad)
            u, v = cam[2] + rad * np.cos(az), cam[3] + rad * np.sin(az)
            ray = oracle.kb8_unproject(cam, u, v)
            assert ray[2] == 1.0
            uv = oracle.kb8_project(cam, ray * np.float32(rng.uniform(0.2, 30)))
            assert abs(uv[0] - u) < 2e-3 and abs(uv[1] - v) < 2e-3
    # the principal point unprojects to the optical axis (theta_d = 0 skips the Newton loop, scale = 1)
    assert np.array_equal(oracle.kb8_unproject(synth.TUMVI_CAM1, synth.TUMVI_CAM1[2], synth.TUMVI_CAM1[3]), [0, 0, 1])


def test_kb8_project_matches_float64_model(oracle):
    rng = np.random.default_rng(6)
    X = rng.normal(0, 1, (300, 3)) * [2, 2, 0.5] + [0, 0, 2.5]
    ref = synth.kb8_project_np(synth.TUMVI_CAM1, X)
    got = np.array([oracle.kb8_project(synth.TUMVI_CAM1, x.astype(np.float32)) for x in X])
    assert np.abs(got - ref).max() < 1e-3


def test_null_vector_matches_numpy_svd(oracle):
    rng = np.random.default_rng(7)
    for k in range(100):
        A = rng.normal(0, 1, (4, 4)).astype(np.float32)
        if k % 3 == 0:  # nearly rank-3, like a triangulation system
            A[3] = (A[0] + 2 * A[1] - A[2]) + rng.normal(0, 1e-3, 4).astype(np.float32)
        v = oracle.null_vector4(A).astype(np.float64)
        vt = np.linalg.svd(A.astype(np.float64))[2][3]
        assert abs(abs(v @ vt) - 1.0) < 1e-6 and abs(np.linalg.norm(v) - 1.0) < 1e-6


def test_triangulate_recovers_known_points_and_gates(oracle):
    R12, t12 = np.eye(3, dtype=np.float32), np.array([0.1, 0.0, 0.0], np.float32)
    rig = oracle.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM2, R12, t12)
    for X in ([0.3, -0.2, 1.5], [-1.0, 0.4, 2.0], [0.05, 0.02, 0.6]):
        X = np.array(X, np.float64)
        uv1, uv2 = synth.kb8_project_np(synth.TUMVI_CAM1, X), synth.kb8_project_np(synth.TUMVI_CAM2, X - t12)
        d, p, gate = oracle.kb8_triangulate(rig, uv1, uv2)
        assert abs(d - X[2]) < 2e-3 * X[2] and np.abs(p - X).max() < 2e-3 * np.linalg.norm(X)
        assert gate[0] < 0.9998 and gate[3] < 1e-2 and gate[4] < 1e-2
    far = np.array([1.0, 1.0, 60.0])  # parallax gate (:356)
    d, _, gate = oracle.kb8_triangulate(rig, synth.kb8_project_np(synth.TUMVI_CAM1, far),
                                        synth.kb8_project_np(synth.TUMVI_CAM2, far - t12))
    assert d == -1 and gate[0] > 0.9998
    X = np.array([0.2, 0.1, 1.0])  # swapped eyes: the rays diverge, the intersection lies behind the cameras (:380-388)
    d, _, _ = oracle.kb8_triangulate(rig, synth.kb8_project_np(synth.TUMVI_CAM2, X - t12), synth.kb8_project_np(synth.TUMVI_CAM1, X))
    assert d in (-2, -3)
    uv1, uv2 = synth.kb8_project_np(synth.TUMVI_CAM1, X), synth.kb8_project_np(synth.TUMVI_CAM2, X - t12)
    d, _, gate = oracle.kb8_triangulate(rig, uv1, uv2 + [0, 9.0])  # vertical disparity: chi-square gates (:393-411)
    assert d in (-4, -5) and max(gate[3], np.nan_to_num(gate[4])) > 1.0


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_fisheye_scene_semantics(oracle, seed):
    sc = synth.fisheye_stereo_scene(seed)
    n, nd, l2r, r2l, dep, pts, gates = oracle.fisheye_stereo_match(sc["kL"], sc["dL"], sc["mono_left"], sc["kR"], sc["dR"],
                                                                   sc["mono_right"], _rig(oracle, sc), sc["level_sigma2"])
    assert 100 < n <= nd and n == int((l2r >= 0).sum()) == int((dep > 0).sum())
    assert not (l2r[: sc["mono_left"]] >= 0).any() and (l2r[l2r >= 0] >= sc["mono_right"]).all()
    codes = gates[:, 5][~np.isnan(gates[:, 5])]
    assert {-1.0, -2.0, -4.0, -5.0} <= set(codes[codes < 0].tolist())
    # serial semantics: a right keypoint claimed by several left ones keeps the last
    claimed = {}
    for i in np.nonzero(l2r >= 0)[0]:
        claimed[int(l2r[i])] = int(i)
    assert sum(1 for r in set(l2r[l2r >= 0].tolist()) if (l2r == r).sum() > 1) >= 5
    assert all(r2l[r] == i for r, i in claimed.items()) and int((r2l >= 0).sum()) == len(claimed)
    # accepted true pairs triangulate near the generating point (pixel noise bounds the accuracy, not the arithmetic)
    ql, qr = sc["true_left"], sc["true_right"]
    good = l2r[ql] == qr
    assert good.sum() > 100
    assert np.all(pts[ql[good], 2] == dep[ql[good]])


def test_oracle_reproduces_fisheye_golden(oracle):
    g = np.load(GOLDEN)
    kL = np.ascontiguousarray(g["kL"]).view(oracle.KP_DTYPE).reshape(-1)
    kR = np.ascontiguousarray(g["kR"]).view(oracle.KP_DTYPE).reshape(-1)
    n, nd, l2r, r2l, dep, pts, _ = oracle.fisheye_stereo_match(kL, g["dL"], int(g["mono_left"]), kR, g["dR"], int(g["mono_right"]),
                                                               g["rig"], g["level_sigma2"])
    assert (n, nd) == (int(g["n"]), int(g["nd"])) and np.array_equal(l2r, g["l2r"]) and np.array_equal(r2l, g["r2l"])
    assert np.allclose(dep, g["depth"], rtol=1e-6, atol=0) and np.allclose(pts, g["p3d"], rtol=1e-6, atol=1e-9)


# ---------------------------------------------------------------------------------------------- HIP path (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_hip_fisheye_stereo_match_parity(oracle, seed):
    import orb_slam3_fast_amd as orbx
    sc = synth.fisheye_stereo_scene(seed)
    args = (sc["kL"], sc["dL"], sc["mono_left"], sc["kR"], sc["dR"], sc["mono_right"])
    ora = oracle.fisheye_stereo_match(*args, _rig(oracle, sc), sc["level_sigma2"])
    hip = orbx.ComputeStereoFishEyeMatches(*args, _rig(orbx, sc), sc["level_sigma2"])
    skipped = compare_float_results(hip, ora[:6], ora[6], sc["mono_left"])
    assert skipped <= 2 and hip[0] > 100


@pytest.mark.gpu
def test_hip_reproduces_fisheye_golden():
    import orb_slam3_fast_amd as orbx
    g = np.load(GOLDEN)
    kL = np.ascontiguousarray(g["kL"]).view(orbx.KP_DTYPE).reshape(-1)
    kR = np.ascontiguousarray(g["kR"]).view(orbx.KP_DTYPE).reshape(-1)
    hip = orbx.ComputeStereoFishEyeMatches(kL, g["dL"], int(g["mono_left"]), kR, g["dR"], int(g["mono_right"]), g["rig"],
                                           g["level_sigma2"])
    gold = (int(g["n"]), int(g["nd"]), g["l2r"], g["r2l"], g["depth"], g["p3d"])
    assert compare_float_results(hip, gold, g["gates"], int(g["mono_left"])) <= 2


@pytest.mark.gpu
def test_hip_fisheye_edge_cases():
    import orb_slam3_fast_amd as orbx
    sc = synth.fisheye_stereo_scene(9, n_left=40, n_right=30, mono_left=10, mono_right=8)
    rig = _rig(orbx, sc)
    # no lapping rows on the left / fewer than two on the right: knnMatch yields no pair (src/Frame.cc:1302)
    for ml, mr in ((40, 8), (10, 30), (10, 29)):
        n, nd, l2r, r2l, dep, pts = orbx.ComputeStereoFishEyeMatches(sc["kL"], sc["dL"], ml, sc["kR"], sc["dR"], mr, rig,
                                                                     sc["level_sigma2"])
        assert (n, nd) == (0, 0) and (l2r == -1).all() and (r2l == -1).all() and (dep == -1).all() and not pts.any()
    n, *_ = orbx.ComputeStereoFishEyeMatches(sc["kL"][:0], sc["dL"][:0], 0, sc["kR"], sc["dR"], 0, rig, sc["level_sigma2"])
    assert n == 0
    with pytest.raises(orbx.OrbxError):
        orbx.ComputeStereoFishEyeMatches(sc["kL"], sc["dL"], 41, sc["kR"], sc["dR"], 8, rig, sc["level_sigma2"])


@pytest.mark.gpu
def test_hip_fisheye_flow_from_extraction(oracle):
    """Config C4 end to end: both eyes extracted with lapping areas on the device, then ComputeStereoFishEyeMatches
    on the extractor outputs; a small-baseline rig so that the synthetic pair (plane disparity) yields accepted pairs."""
    import orb_slam3_fast_amd as orbx
    w = h = 512
    L, R = synth.stereo_pair(w, h, 95)
    exL = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    exR = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    mL, kL, dL = exL(L, (100, 511))
    mR, kR, dR = exR(R, (0, 400))
    sigma2 = exL.GetScaleSigmaSquares()  # Frame::mvLevelSigma2 (src/Frame.cc:1222)
    rig = orbx.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM1, np.eye(3), [0.1, 0.0, 0.0])
    hip = orbx.ComputeStereoFishEyeMatches(kL, dL, mL, kR, dR, mR, rig, sigma2)
    ora = oracle.fisheye_stereo_match(kL, dL, mL, kR, dR, mR, oracle.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM1, np.eye(3), [0.1, 0, 0]),
                                      sigma2)
    assert compare_float_results(hip, ora[:6], ora[6], mL) <= 2
    assert hip[1] > 20


@pytest.mark.gpu
def test_hip_fisheye_batch_on_device_results(oracle):
    """Batched, device-resident variant (orbx_fisheye_stereo_match_batch): three fisheye stereo pairs extracted in one
    batch on one handle (left eyes = images 0..2, right eyes = 3..5, per-image lapping areas), associated without
    leaving the device, against the oracle run on the downloaded keypoints."""
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    w = h = 512
    pairs = [synth.stereo_pair(w, h, 96 + i) for i in range(3)]
    imgs = np.stack([p[0] for p in pairs] + [p[1] for p in pairs])
    lap = np.array([[100, 511]] * 3 + [[0, 400]] * 3, np.int32)
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=6)
    d = DeviceBuffer.from_numpy(imgs)
    ex.extract_batch_device(d.ptr.value, 6, w, h, w, w * h, lap=lap)
    rig = orbx.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM1, np.eye(3), [0.1, 0.0, 0.0])
    orbx.fisheye_match_async(ex, ex, rig, first_left=0, first_right=3, n_pairs=3)
    ex.sync()
    sigma2 = ex.GetScaleSigmaSquares()
    for p in range(3):
        mL, kL, dL = ex.download(p)
        mR, kR, dR = ex.download(3 + p)
        assert 0 < mL < len(kL) and 0 < mR < len(kR)
        n, nd, l2r, r2l, dep, pts = orbx.fisheye_download(ex, ex, p)
        assert (l2r[len(kL):] == -1).all() and (r2l[len(kR):] == -1).all() and (dep[len(kL):] == -1).all()
        hip = (n, nd, l2r[: len(kL)], r2l[: len(kR)], dep[: len(kL)], pts[: len(kL)])
        ora = oracle.fisheye_stereo_match(kL, dL, mL, kR, dR, mR, oracle.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM1, np.eye(3), [0.1, 0, 0]),
                                          sigma2)
        assert compare_float_results(hip, ora[:6], ora[6], mL) <= 2 and nd > 20
        # and it is the same routine as the host-array entry point
        host = orbx.ComputeStereoFishEyeMatches(kL, dL, mL, kR, dR, mR, rig, sigma2)
        assert host[0] == n and np.array_equal(host[2], hip[2]) and host[4].tobytes() == hip[4].tobytes()


@pytest.mark.gpu
def test_hip_fisheye_batch_scan_edge_cases(oracle):
    """The batched association's 2-NN scan (round 6: k_fisheye_scan, Hamming distances as an i8 matrix product) on crafted
    lapping sets that extractor output never produces, injected with orbx_debug_upload_results: train counts around the 16-row
    MFMA group and the 128-row sub-tile (0, 1, 2, 15 .. 17, 127 .. 129, 257), query counts around the 16- / 32- / 128-query
    wave and workgroup edges, exact ties (duplicated train descriptors at several distances: BFMatcher keeps the FIRST minimum,
    a duplicate of the best makes the Lowe ratio fail), all-zero and all-one descriptors (distance 0 and 256), and a pair whose
    left lapping area is empty.  Integer results (pair indices, descriptor-match counts) must equal the oracle's exactly, the
    float results within the tolerance of compare_float_results."""
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    w = h = 512
    npairs = 4
    imgs = np.stack([synth.mono_frame(w, h, 300 + i) for i in range(2 * npairs)])
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2 * npairs)
    d = DeviceBuffer.from_numpy(imgs)
    ex.extract_batch_device(d.ptr.value, 2 * npairs, w, h, w, w * h)   # (allocates the result arrays; contents are replaced below)
    ex.sync()
    sigma2 = ex.GetScaleSigmaSquares()
    rng = np.random.default_rng(4242)
    sizes = [(0, 5), (3, 0), (3, 1), (1, 2), (5, 15), (17, 16), (33, 17), (127, 127), (128, 128), (129, 129), (130, 257), (300, 31)]
    cases = []
    for nq, nt in sizes:
        sc = synth.fisheye_stereo_scene(500 + nq + 7 * nt, n_left=nq + 6, n_right=nt + 4, mono_left=6, mono_right=4)
        kL, dL, kR, dR = sc["kL"].copy(), sc["dL"].copy(), sc["kR"].copy(), sc["dR"].copy()
        if nt >= 8 and nq >= 3:
            dR[4 + nt - 1] = dR[4 + 1]                       # the LAST train duplicates train 1: ties go to the lower index
            dR[4 + 3] = dR[4 + 2]                            # adjacent duplicate
            dL[6 + 0] = dR[4 + 1]                            # query 0 at distance 0 from both copies: ratio test fails (0 < 0.7 * 0)
            dL[6 + 1] = dR[4 + 2]
            dL[6 + 1, 0] ^= 1                                # query 1 at distance 1 from trains 2 and 3
            dL[6 + 2] = 0
            dR[4 + 5] = 255                                  # distance 256 from an all-zero query
            dR[4 + 6] = 0                                    # and distance 0
        cases.append((kL, dL, 6, kR, dR, 4, sc))
    lib = orbx.lib()
    for c0 in range(0, len(cases), npairs):
        chunk = cases[c0:c0 + npairs]
        for p, (kL, dL, mL, kR, dR, mR, sc) in enumerate(chunk):
            for image, (k, dd, mono) in ((p, (kL, dL, mL)), (npairs + p, (kR, dR, mR))):
                k = np.ascontiguousarray(k.astype(orbx.KP_DTYPE))
                dd = np.ascontiguousarray(dd, np.uint8)
                orbx._check(lib.orbx_debug_upload_results(ex._h, image, orbx._p(k), orbx._p(dd), len(k), mono))
        rig = _rig(orbx, chunk[0][6])
        orbx.fisheye_match_async(ex, ex, rig, first_left=0, first_right=npairs, n_pairs=len(chunk))
        ex.sync()
        for p, (kL, dL, mL, kR, dR, mR, sc) in enumerate(chunk):
            n, nd, l2r, r2l, dep, pts = orbx.fisheye_download(ex, ex, p)
            hip = (n, nd, l2r[: len(kL)], r2l[: len(kR)], dep[: len(kL)], pts[: len(kL)])
            ora = oracle.fisheye_stereo_match(kL, dL, mL, kR, dR, mR, _rig(oracle, chunk[0][6]), sigma2)
            assert nd == ora[1], (len(kL) - mL, len(kR) - mR, nd, ora[1])      # the Lowe-accepted pairs: pure integer work
            assert compare_float_results(hip, ora[:6], ora[6], mL) <= 2, (len(kL) - mL, len(kR) - mR)
            assert (l2r[len(kL):] == -1).all() and (r2l[len(kR):] == -1).all()
            if len(kR) - mR >= 8 and len(kL) - mL >= 3:
                # the 2-NN itself, without the geometry: the oracle's brute-force scan on the same rows
                idx, dist, ok = oracle.bf_knn2(dL[mL:], dR[mR:])
                assert idx[0, 0] == 1 and dist[0, 0] == 0 and dist[0, 1] == 0      # duplicate pair: first minimum = the lower index
                assert idx[1, 0] == 2 and idx[1, 1] == 3 and dist[1, 0] == 1
                assert dist[2, 0] == 0 and idx[2, 0] == 6


# ---------------------------------------------------------------------------------------------- SearchByProjection, Nleft != -1
def _fisheye_frame(mod, seed, w=512, h=512):
    """A stereo-fisheye frame as the tracker sees it: N = nL + nR keypoints (mvKeys then mvKeysRight) from the synthetic
    stereo pair, partner arrays from a brute-force association, and seeded map-point / projected-point views that
    project near left keypoints (left camera) and near their partners or random right keypoints (right camera)."""
    rng = np.random.default_rng(1000 + seed)
    L, R = synth.stereo_pair(w, h, 140 + seed)
    from oracle import oracle_py as O
    eL, eR = O.OracleExtractor(800), O.OracleExtractor(800)
    _, kL, dL = eL.extract(L)
    _, kR, dR = eR.extract(R)
    nL, nR = len(kL), len(kR)
    idx, dist, ok = O.bf_knn2(dL, dR)
    l2r = np.where(ok.astype(bool) & (rng.random(nL) < 0.8), idx[:, 0], -1).astype(np.int32)
    r2l = np.full(nR, -1, np.int32)
    for i in np.nonzero(l2r >= 0)[0]:
        r2l[l2r[i]] = i
    kps = np.concatenate([kL, kR])
    desc = np.concatenate([dL, dR])
    sf = eL.tables()["scale"]
    n = 900
    src = rng.integers(0, nL, n)                 # the left keypoint each view is derived from
    flips = rng.random((n, 32, 8)) < 0.05
    vdesc = dL[src] ^ np.packbits(flips, axis=2).reshape(n, 32)
    part = np.where(l2r[src] >= 0, l2r[src], rng.integers(0, nR, n))
    mps = np.zeros(n, mod.MP_DTYPE)
    mps["proj_x"], mps["proj_y"] = kL["x"][src] + rng.normal(0, 2.5, n), kL["y"][src] + rng.normal(0, 2.5, n)
    mps["proj_xr"] = kR["x"][part] + rng.normal(0, 2.5, n)
    mps["view_cos"], mps["track_depth"] = rng.choice([0.9, 0.9985], n), rng.uniform(1, 80, n)
    mps["predicted_level"] = np.clip(kL["octave"][src] + rng.integers(-1, 2, n), 0, 7)
    mps["in_view"], mps["bad"], mps["has_observations"] = rng.random(n) < 0.8, rng.random(n) < 0.05, rng.random(n) < 0.7
    mps["desc"] = vdesc
    mpr = np.zeros(n, mod.MPR_DTYPE)
    mpr["proj_yr"], mpr["view_cos_r"] = kR["y"][part] + rng.normal(0, 2.5, n), rng.choice([0.9, 0.9985], n)
    mpr["predicted_level_r"] = np.where(rng.random(n) < 0.1, -1, np.clip(kR["octave"][part] + rng.integers(-1, 2, n), 0, 7))
    mpr["in_view_r"] = rng.random(n) < 0.7
    pts = np.zeros(n, mod.PP_DTYPE)
    pts["u"], pts["v"] = mps["proj_x"], mps["proj_y"]
    pts["radius"], pts["angle"] = (np.float32(7.0) * sf[kL["octave"][src]]), kL["angle"][src] + rng.normal(0, 4, n)
    pts["min_level"], pts["max_level"] = kL["octave"][src] - 1, kL["octave"][src] + 1
    pts["valid"], pts["has_observations"], pts["desc"] = rng.random(n) < 0.85, mps["has_observations"], vdesc
    uvr = np.stack([mps["proj_xr"], mpr["proj_yr"]], 1).astype(np.float32)
    occ = (rng.random(nL + nR) < 0.05).astype(np.uint8)
    return dict(kps=kps, desc=desc, nL=nL, sf=sf, mps=mps, mpr=mpr, pts=pts, uvr=uvr, l2r=l2r, r2l=r2l, occ=occ,
                bounds=(0.0, 0.0, float(w), float(h)))


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_fisheye_projection_semantics(oracle, seed):
    f = _fisheye_frame(oracle, seed)
    nL = f["nL"]
    n, m, occ = oracle.search_by_projection_fisheye(f["kps"], f["desc"], nL, f["bounds"], f["sf"], f["mps"], f["mpr"], 3.0, True, 60.0,
                                                    0.8, f["l2r"], f["r2l"], f["occ"])
    assert n > 150 and (m[:nL] >= 0).sum() > 50 and (m[nL:] >= 0).sum() > 50
    # a left keypoint taken by a point takes its stereo partner along (unless a later point re-took the partner slot)
    both = [i for i in np.nonzero(m[:nL] >= 0)[0] if f["l2r"][i] >= 0 and m[nL + f["l2r"][i]] == m[i]]
    assert len(both) > 20
    # without partners and with the right camera switched off the left half equals the pinhole routine without mvuRight
    off = f["mpr"].copy()
    off["in_view_r"] = 0
    none_l, none_r = np.full(nL, -1, np.int32), np.full(len(f["kps"]) - nL, -1, np.int32)
    n1, m1, o1 = oracle.search_by_projection_fisheye(f["kps"], f["desc"], nL, f["bounds"], f["sf"], f["mps"], off, 3.0, True, 60.0, 0.8,
                                                     none_l, none_r, f["occ"])
    n2, m2, o2 = oracle.search_by_projection(f["kps"][:nL], f["desc"][:nL], None, f["bounds"], f["sf"], f["mps"], 3.0, True, 60.0, 0.8,
                                             f["occ"][:nL])
    assert n1 == n2 and np.array_equal(m1[:nL], m2) and (m1[nL:] == -1).all() and np.array_equal(o1[:nL], o2)
    nf, mf, of = oracle.search_by_projection_frame_fisheye(f["kps"], f["desc"], nL, f["bounds"], f["pts"], f["uvr"], True, f["occ"])
    assert nf > 100 and (mf[nL:] >= 0).sum() > 30


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_hip_fisheye_projection_matchers_parity(oracle, seed):
    import orb_slam3_fast_amd as orbx
    f = _fisheye_frame(orbx, seed)
    m = orbx.ORBmatcher(0.8, True)
    for th, far in ((3.0, True), (1.0, False)):
        got = m.SearchByProjectionFisheye(f["kps"], f["desc"], f["nL"], f["bounds"], f["sf"], f["mps"], f["mpr"], f["l2r"], f["r2l"],
                                          f["occ"], th, far, 60.0)
        exp = oracle.search_by_projection_fisheye(f["kps"], f["desc"], f["nL"], f["bounds"], f["sf"], f["mps"].view(oracle.MP_DTYPE),
                                                  f["mpr"].view(oracle.MPR_DTYPE), th, far, 60.0, 0.8, f["l2r"], f["r2l"], f["occ"])
        assert got[0] == exp[0] and np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]) and got[0] > 100
    for ori in (True, False):
        mm = orbx.ORBmatcher(0.8, ori)
        got = mm.SearchByProjectionFrameFisheye(f["kps"], f["desc"], f["nL"], f["bounds"], f["pts"], f["uvr"], f["occ"])
        exp = oracle.search_by_projection_frame_fisheye(f["kps"], f["desc"], f["nL"], f["bounds"], f["pts"].view(oracle.PP_DTYPE),
                                                        f["uvr"], ori, f["occ"])
        assert got[0] == exp[0] and np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]) and got[0] > 80
    # degenerate frames: no right keypoints / no points
    nL = f["nL"]
    got = m.SearchByProjectionFisheye(f["kps"][:nL], f["desc"][:nL], nL, f["bounds"], f["sf"], f["mps"], f["mpr"],
                                      np.full(nL, -1, np.int32), np.zeros(0, np.int32), f["occ"][:nL], 3.0, True, 60.0)
    exp = oracle.search_by_projection_fisheye(f["kps"][:nL], f["desc"][:nL], nL, f["bounds"], f["sf"], f["mps"].view(oracle.MP_DTYPE),
                                              f["mpr"].view(oracle.MPR_DTYPE), 3.0, True, 60.0, 0.8, np.full(nL, -1, np.int32),
                                              np.zeros(0, np.int32), f["occ"][:nL])
    assert got[0] == exp[0] and np.array_equal(got[1], exp[1])
    got = m.SearchByProjectionFrameFisheye(f["kps"], f["desc"], nL, f["bounds"], f["pts"][:0], f["uvr"][:0], f["occ"])
    assert got[0] == 0 and (got[1] == -1).all()


GOLDEN_PROJ = os.path.join(os.path.dirname(__file__), "golden", "fisheye_projection.npz")


def _golden_proj(mod):
    g = np.load(GOLDEN_PROJ)
    kps = np.ascontiguousarray(g["kps"]).view(mod.KP_DTYPE).reshape(-1)
    mps = np.ascontiguousarray(g["mps"]).view(mod.MP_DTYPE).reshape(-1)
    mpr = np.ascontiguousarray(g["mpr"]).view(mod.MPR_DTYPE).reshape(-1)
    pts = np.ascontiguousarray(g["pts"]).view(mod.PP_DTYPE).reshape(-1)
    return g, kps, mps, mpr, pts, tuple(float(v) for v in g["bounds"])


def test_oracle_reproduces_fisheye_projection_golden(oracle):
    g, kps, mps, mpr, pts, bounds = _golden_proj(oracle)
    n1, m1, o1 = oracle.search_by_projection_fisheye(kps, g["desc"], int(g["n_left"]), bounds, g["scale"], mps, mpr, 3.0, True, 60.0, 0.8,
                                                     g["l2r"], g["r2l"], g["occ"])
    assert n1 == int(g["map_n"]) and np.array_equal(m1, g["map_match"]) and np.array_equal(o1, g["map_occ"])
    n2, m2, o2 = oracle.search_by_projection_frame_fisheye(kps, g["desc"], int(g["n_left"]), bounds, pts, g["uvr"], True, g["occ"])
    assert n2 == int(g["frame_n"]) and np.array_equal(m2, g["frame_match"]) and np.array_equal(o2, g["frame_occ"])


@pytest.mark.gpu
def test_hip_reproduces_fisheye_projection_golden():
    import orb_slam3_fast_amd as orbx
    g, kps, mps, mpr, pts, bounds = _golden_proj(orbx)
    m = orbx.ORBmatcher(0.8, True)
    n1, m1, o1 = m.SearchByProjectionFisheye(kps, g["desc"], int(g["n_left"]), bounds, g["scale"], mps, mpr, g["l2r"], g["r2l"], g["occ"],
                                             3.0, True, 60.0)
    assert n1 == int(g["map_n"]) and np.array_equal(m1, g["map_match"]) and np.array_equal(o1, g["map_occ"])
    n2, m2, o2 = m.SearchByProjectionFrameFisheye(kps, g["desc"], int(g["n_left"]), bounds, pts, g["uvr"], g["occ"])
    assert n2 == int(g["frame_n"]) and np.array_equal(m2, g["frame_match"]) and np.array_equal(o2, g["frame_occ"])


@pytest.mark.gpu
def test_hip_single_call_fisheye_frame(oracle):
    """One fisheye stereo frame through the latency path: orbx_extract_stereo (lapping areas, both eyes in one batch) and
    the device-resident association on the same handle (left = image 0, right = image 1)."""
    import orb_slam3_fast_amd as orbx
    w = h = 512
    L, R = synth.stereo_pair(w, h, 97)
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    (mL, kL, dL), (mR, kR, dR) = ex.extract_stereo(L, R, (100, 511), (0, 400))
    rig = orbx.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM1, np.eye(3), [0.1, 0.0, 0.0])
    orbx.fisheye_match_async(ex, ex, rig, first_left=0, first_right=1, n_pairs=1)
    n, nd, l2r, r2l, dep, pts = orbx.fisheye_download(ex, ex, 0)
    ora = oracle.fisheye_stereo_match(kL, dL, mL, kR, dR, mR, oracle.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM1, np.eye(3), [0.1, 0, 0]),
                                      ex.GetScaleSigmaSquares())
    hip = (n, nd, l2r[: len(kL)], r2l[: len(kR)], dep[: len(kL)], pts[: len(kL)])
    assert compare_float_results(hip, ora[:6], ora[6], mL) <= 2 and nd > 20


@pytest.mark.gpu
def test_hip_fisheye_projection_matchers_batched_over_an_extraction_batch(oracle):
    """VERDICT (round 4), item 5: both stereo-fisheye SearchByProjection flavours (src/ORBmatcher.cc:41-221, 1594-1806 with
    Nleft != -1) on the two-camera frames of an extraction batch in one call each -- the keypoints / descriptors of both cameras stay
    in HBM (laid side by side on the device), every kernel of the chain runs once for all frames and cameras, the fixed-point rounds
    are enqueued without a convergence-flag read.  Per frame the result is the oracle's (= the one-shot call's); frames with
    different point counts incl. none; then the redo path (tiny candidate capacity, a single blind round) in a fresh process."""
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    F, w, h = 3, 640, 480
    fr = [_fisheye_frame(orbx, 10 + i, w, h) for i in range(F)]
    imgs = np.stack([synth.stereo_pair(w, h, 140 + 10 + i)[0] for i in range(F)] + [synth.stereo_pair(w, h, 140 + 10 + i)[1] for i in range(F)])
    dev = DeviceBuffer.from_numpy(imgs)
    ex = orbx.ORBextractor(800, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2 * F)
    ex.extract_batch_device(dev.ptr.value, 2 * F, w, h, w, w * h)
    ex.sync()
    cap = ex.capacity
    stride = 900
    mps, mpr = np.zeros((F, stride), orbx.MP_DTYPE), np.zeros((F, stride), orbx.MPR_DTYPE)
    pts, uvr = np.zeros((F, stride), orbx.PP_DTYPE), np.zeros((F, stride, 2), np.float32)
    l2r, r2l = np.full((F, cap), -1, np.int32), np.full((F, cap), -1, np.int32)
    occ = np.zeros((F, 2 * cap), np.uint8)
    npts = np.zeros(F, np.int32)
    for f in range(F):
        g = fr[f]
        nL, n = g["nL"], len(g["kps"])
        _, kL, dL = ex.download(f)
        _, kR, dR = ex.download(F + f)
        assert np.array_equal(np.concatenate([kL, kR]).view(np.uint8), g["kps"].view(np.uint8)) and np.array_equal(np.concatenate([dL, dR]), g["desc"])
        npts[f] = 0 if f == 1 else 900 - 100 * f
        mps[f], mpr[f], pts[f], uvr[f] = g["mps"], g["mpr"], g["pts"], g["uvr"]
        l2r[f, :nL], r2l[f, :n - nL] = g["l2r"], g["r2l"]
        occ[f, :n] = g["occ"]
    for th, far in ((3.0, True), (1.0, False)):
        m = orbx.ORBmatcher(0.8, True)
        nm, match, oc = m.SearchByProjectionFisheyeBatch(ex, 0, F, F, fr[0]["bounds"], mps, mpr, npts, l2r, r2l, occ, th, far, 60.0)
        for f in range(F):
            g, k = fr[f], int(npts[f])
            n = len(g["kps"])
            exp = oracle.search_by_projection_fisheye(g["kps"], g["desc"], g["nL"], g["bounds"], g["sf"], g["mps"][:k].view(oracle.MP_DTYPE),
                                                      g["mpr"][:k].view(oracle.MPR_DTYPE), th, far, 60.0, 0.8, g["l2r"], g["r2l"], g["occ"])
            assert nm[f] == exp[0] and np.array_equal(match[f, :n], exp[1]) and np.array_equal(oc[f, :n], exp[2]), (f, th, nm[f], exp[0])
            assert (match[f, n:] == -1).all()
        assert nm.sum() > 150
    for ori in (True, False):
        mm = orbx.ORBmatcher(0.8, ori)
        nm, match, oc = mm.SearchByProjectionFrameFisheyeBatch(ex, 0, F, F, fr[0]["bounds"], pts, uvr, npts, occ)
        for f in range(F):
            g, k = fr[f], int(npts[f])
            n = len(g["kps"])
            exp = oracle.search_by_projection_frame_fisheye(g["kps"], g["desc"], g["nL"], g["bounds"], g["pts"][:k].view(oracle.PP_DTYPE),
                                                            g["uvr"][:k], ori, g["occ"])
            assert nm[f] == exp[0] and np.array_equal(match[f, :n], exp[1]) and np.array_equal(oc[f, :n], exp[2]), (f, ori, nm[f], exp[0])
        assert nm.sum() > 100
    if os.environ.get("ORBX_PROJ_CAND_CAP") is None:
        import subprocess
        import sys
        e = dict(os.environ, ORBX_PROJ_CAND_CAP="64", ORBX_PROJ_BLIND="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__),
                            "-k", "test_hip_fisheye_projection_matchers_batched_over_an_extraction_batch"], env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
