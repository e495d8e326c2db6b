"""Device-side projection for the batched local-map matcher (round 6): Frame::isInFrustum +
MapPoint::PredictScale (src/Frame.cc:632-690, src/MapPoint.cc:559-573) evaluated on the GPU for every (map point, frame), the views
consumed in place by the batched SearchByProjection (src/ORBmatcher.cc:41-221).

Parity: the projection is float arithmetic -- tolerance parity like the KB8 tail.  Gate decisions must equal the oracle's unless the
oracle reports the decisive quantity within 1e-5 (relative) of its threshold; coordinates / depth / viewing cosine within 1e-5
relative (the expression order is the reference's, so they are in fact bit-equal except for the division / square-root
implementation); the predicted level equal unless log(ratio) / logScaleFactor lies within 1e-4 of an integer.  The MATCHER is
exact: fed with the device-made views, the oracle's SearchByProjection must return the device's matches bit for bit."""
import numpy as np
import pytest

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _scene(rng, w, h, F, kps_per_frame, descs_per_frame, fx=420.0):
    """Map points back-projected from the frames' own keypoints (so that the matcher finds them) plus points behind / beside the
    cameras; poses = small motions around the identity."""
    cx, cy, bf = w / 2.0, h / 2.0, 0.12 * fx
    poses, Rs, ts = [], [], []
    for f in range(F):
        R = _rot(*(rng.normal(0, 0.01, 3)))
        t = rng.normal(0, 0.05, 3)
        Ow = -R.T @ t
        poses.append(np.concatenate([R.reshape(-1), t, Ow, [fx, fx, cx, cy, bf]]).astype(np.float32))
        Rs.append(R), ts.append(t)
    pos, nrm, mind, maxd, desc, flags = [], [], [], [], [], []
    for f in range(F):
        k, d = kps_per_frame[f], descs_per_frame[f]
        take = rng.choice(len(k), size=min(260, len(k)), replace=False)
        for i in take:
            z = rng.uniform(2.0, 30.0)
            pc = np.array([(k["x"][i] + rng.normal(0, 1.5) - cx) / fx * z, (k["y"][i] + rng.normal(0, 1.5) - cy) / fx * z, z])
            P = Rs[f].T @ (pc - ts[f])
            pos.append(P)
            v = P - (-Rs[f].T @ ts[f])
            dist = np.linalg.norm(v)
            n_ = v / dist + rng.normal(0, 0.25, 3)
            nrm.append(n_ / np.linalg.norm(n_))
            sc = 1.2 ** int(k["octave"][i])
            maxd.append(dist * sc * rng.uniform(0.95, 1.05))
            mind.append(maxd[-1] / 1.2 ** 7)
            desc.append(d[i] ^ np.packbits(rng.random((32, 8)) < 0.03, axis=1).reshape(32))
            flags.append((0 if rng.random() > 0.04 else 1) | (2 if rng.random() < 0.9 else 0))
    for _ in range(300):   # clutter: behind the cameras, outside the image, far outside the distance range
        P = rng.normal(0, 8.0, 3)
        pos.append(P)
        n_ = rng.normal(0, 1, 3)
        nrm.append(n_ / np.linalg.norm(n_))
        m = rng.uniform(0.5, 40.0)
        maxd.append(m)
        mind.append(m / 3.5)
        desc.append(rng.integers(0, 256, 32, dtype=np.uint8))
        flags.append(2)
    return (np.stack(poses), np.array(pos, np.float32), np.array(nrm, np.float32), np.array(mind, np.float32), np.array(maxd, np.float32),
            np.array(desc, np.uint8), np.array(flags, np.uint8))


@pytest.mark.gpu
def test_device_projection_and_matcher_against_the_oracle(oracle):
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    w, h, nf, F = 640, 480, 1000, 5
    rng = np.random.default_rng(31)
    cur = [synth.stereo_pair(w, h, 140 + f, 1) for f in range(F)]
    dev = DeviceBuffer.from_numpy(np.stack([c[0] for c in cur] + [c[1] for c in cur]))
    ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2 * F)
    ex.extract_batch_device(dev.ptr.value, 2 * F, w, h, w, w * h)
    orbx.stereo_match_async(ex, ex, 0.12 * 532.03, 0.12, first_left=0, first_right=F, n_pairs=F)
    ex.sync()
    frames = [ex.download(f)[1:] for f in range(F)]
    poses, pos, nrm, mind, maxd, desc, flags = _scene(rng, w, h, F, [k for k, _ in frames], [d for _, d in frames])
    n = len(pos)
    bounds = (0.0, 0.0, float(w), float(h))
    skip = (rng.random((F, n)) < 0.05).astype(np.uint8)
    ex.map_upload(pos, nrm, mind, maxd, desc, flags)
    views = ex.project_map_points(poses, bounds, 0.5, skip, want_views=True)
    logsf = np.float32(np.log(np.float32(1.2)))
    n_view = n_gate = n_lvl = 0
    for f in range(F):
        ov, mg = oracle.is_in_frustum(poses[f], pos, nrm, mind, maxd, bounds, 0.5, logsf, 8, flags, desc)
        ov["in_view"] = np.where(skip[f] != 0, 0, ov["in_view"])
        v = views[f]
        assert np.array_equal(v["bad"], ov["bad"]) and np.array_equal(v["has_observations"], ov["has_observations"])
        assert np.array_equal(v["desc"], ov["desc"])
        diff = v["in_view"] != ov["in_view"]
        assert (mg[diff, 0] < 1e-5).all(), (f, mg[diff])          # a gate decision may only differ AT its threshold
        n_gate += int(diff.sum())
        both = (v["in_view"] != 0) & (ov["in_view"] != 0)
        n_view += int(both.sum())
        for name in ("proj_x", "proj_y", "proj_xr", "track_depth", "view_cos"):
            a, b = v[name][both].astype(np.float64), ov[name][both].astype(np.float64)
            assert (np.abs(a - b) <= 1e-5 * np.maximum(1.0, np.abs(b))).all(), name
        ld = both & (v["predicted_level"] != ov["predicted_level"])
        assert (mg[ld, 1] < 1e-4).all() and (np.abs(v["predicted_level"][ld] - ov["predicted_level"][ld]) <= 1).all()
        n_lvl += int(ld.sum())
    assert n_view > 600 and n_gate <= 3 and n_lvl <= 3, (n_view, n_gate, n_lvl)
    # the matcher on the device-resident views == the oracle's SearchByProjection on the SAME (downloaded) views, bit for bit
    occ_in = (rng.random((F, ex.capacity)) < 0.04).astype(np.uint8)
    u_all, _ = orbx.ComputeStereoMatches(ex, ex, 0.12 * 532.03, 0.12, first_left=0, first_right=F, n_pairs=F)
    sf = ex.GetScaleFactors()
    total = 0
    for use_ur in (True, False):
        nm, match, occ = orbx.ORBmatcher(0.8, True).SearchByProjectionBatchDevice(ex, 0, F, bounds, occ_in, th=3.0,
                                                                                  stereo_pair0=0 if use_ur else -1)
        for f in range(F):
            k, d = frames[f]
            on, om, oo = oracle.search_by_projection(k, d, u_all[f, :len(k)] if use_ur else None, bounds, sf, views[f], 3.0, False, 50.0,
                                                     0.8, occ_in[f, :len(k)])
            assert nm[f] == on and np.array_equal(match[f, :len(k)], om) and np.array_equal(occ[f, :len(k)], oo), (use_ur, f)
            total += on
        # and equals the host-view batched entry
        nm2, match2, occ2 = orbx.ORBmatcher(0.8, True).SearchByProjectionBatch(ex, 0, F, bounds, views, np.full(F, n, np.int32), occ_in,
                                                                               th=3.0, stereo_pair0=0 if use_ur else -1)
        assert np.array_equal(nm, nm2) and np.array_equal(match, match2) and np.array_equal(occ, occ2)
    assert total > 400, total
    # argument errors
    with pytest.raises(orbx.OrbxError):
        orbx.ORBmatcher(0.8, True).SearchByProjectionBatchDevice(ex, 0, F + 1, bounds)       # more frames than were projected


def _quat(R):
    """Unit quaternion (x, y, z, w) of a rotation matrix (w > 0: small rotations), float64."""
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


@pytest.mark.gpu
def test_device_projection_of_last_frames_and_matcher_against_the_oracle(oracle):
    """Round 6, the frame-to-frame flavour: the projection block of SearchByProjection(CurrentFrame, LastFrame)
    (src/ORBmatcher.cc:1606-1669; Tcw * x3Dw = Sophus' quaternion sandwich, Thirdparty/Sophus/sophus/so3.hpp:358-366) on the device
    for the LastFrames of five cameras, consumed in place by the batched matcher.  Parity as above: a gate decision may differ from
    the oracle's only where the oracle reports the decisive quantity within 1e-5 of its threshold, coordinates within 1e-5
    relative (in fact bit-equal up to the division), level windows / radius / angle / descriptors equal; the MATCHER is exact: on
    the device-made views the oracle's SearchByProjectionFrame returns the device's matches bit for bit, with and without the
    stereo-consistency gate, and the device-view call equals the host-view batched entry."""
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    w, h, nf, F = 640, 480, 1000, 5
    rng = np.random.default_rng(77)
    cur = [synth.stereo_pair(w, h, 160 + f, 1) for f in range(F)]
    dev = DeviceBuffer.from_numpy(np.stack([c[0] for c in cur] + [c[1] for c in cur]))
    ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2 * F)
    ex.extract_batch_device(dev.ptr.value, 2 * F, w, h, w, w * h)
    orbx.stereo_match_async(ex, ex, 0.12 * 532.03, 0.12, first_left=0, first_right=F, n_pairs=F)
    ex.sync()
    frames = [ex.download(f)[1:] for f in range(F)]
    fx, cx, cy = 420.0, w / 2.0, h / 2.0
    bf = 0.12 * fx
    stride = 700
    npts = np.array([700, 650, 0, 512, 699], np.int32)
    pos = np.zeros((F, stride, 3), np.float32)
    octv = np.zeros((F, stride), np.int32)
    ang = np.zeros((F, stride), np.float32)
    desc = rng.integers(0, 256, (F, stride, 32), dtype=np.uint8)
    flags = np.zeros((F, stride), np.uint8)
    poses = np.zeros((F, 12), np.float32)
    directions = np.array([0, 1, 2, 0, 1], np.int32)
    for f in range(F):
        R = _rot(*(rng.normal(0, 0.01, 3)))
        t = rng.normal(0, 0.05, 3)
        poses[f] = np.concatenate([_quat(R), t, [fx, fx, cx, cy, bf]])
        k, d = frames[f]
        n = int(npts[f])
        src = rng.integers(0, len(k), n)                     # the current-frame keypoint each LastFrame point lands near
        z = rng.uniform(2.0, 30.0, n)
        pc = np.stack([(k["x"][src] + rng.normal(0, 1.5, n) - cx) / fx * z, (k["y"][src] + rng.normal(0, 1.5, n) - cy) / fx * z, z], 1)
        clutter = rng.random(n) < 0.15                        # behind the camera / outside the image
        pc[clutter] = rng.normal(0, 6.0, (int(clutter.sum()), 3))
        pos[f, :n] = (pc - t) @ R                             # x3Dw = R^T (x3Dc - t)
        octv[f, :n] = np.clip(k["octave"][src] + rng.integers(-1, 2, n), 0, 7)
        ang[f, :n] = k["angle"][src] + rng.normal(0, 4, n)
        desc[f, :n] = d[src] ^ np.packbits(rng.random((n, 32, 8)) < 0.03, axis=2).reshape(n, 32)
        flags[f, :n] = (rng.random(n) < 0.9).astype(np.uint8) | ((rng.random(n) < 0.85).astype(np.uint8) << 1)
    bounds = (0.0, 0.0, float(w), float(h))
    th = 7.0
    ex.last_frames_upload(npts, pos, octv, ang, desc, flags)
    views = ex.project_last_frames(poses, directions, bounds, th, want_views=True)
    sf = ex.GetScaleFactors()
    n_valid = n_gate = 0
    for f in range(F):
        n = int(npts[f])
        ov, mg = oracle.project_last_frame(poses[f], directions[f], pos[f, :n], octv[f, :n], ang[f, :n], flags[f, :n], desc[f, :n], th, sf,
                                           bounds)
        v = views[f, :n]
        assert (views[f, n:]["valid"] == 0).all()
        assert np.array_equal(v["desc"], ov["desc"]) and np.array_equal(v["has_observations"], ov["has_observations"])
        assert np.array_equal(v["angle"].view(np.uint32), ov["angle"].view(np.uint32))
        diff = v["valid"] != ov["valid"]
        assert (mg[diff] < 1e-5).all(), (f, mg[diff])
        n_gate += int(diff.sum())
        both = (v["valid"] != 0) & (ov["valid"] != 0)
        n_valid += int(both.sum())
        for name in ("u", "v", "ur"):
            a, b = v[name][both].astype(np.float64), ov[name][both].astype(np.float64)
            assert (np.abs(a - b) <= 1e-5 * np.maximum(1.0, np.abs(b))).all(), name
        for name in ("radius", "min_level", "max_level"):
            assert np.array_equal(v[name][both], ov[name][both]), name
    assert n_valid > 1500 and n_gate <= 3, (n_valid, n_gate)
    occ_in = (rng.random((F, ex.capacity)) < 0.04).astype(np.uint8)
    u_all, _ = orbx.ComputeStereoMatches(ex, ex, 0.12 * 532.03, 0.12, first_left=0, first_right=F, n_pairs=F)
    total = 0
    for use_ur in (True, False):
        m = orbx.ORBmatcher(0.9, True)
        nm, match, occ = m.SearchByProjectionFrameBatchDevice(ex, 0, F, bounds, occ_in, stereo_pair0=0 if use_ur else -1)
        for f in range(F):
            k, d = frames[f]
            on, om, oo = oracle.search_by_projection_frame(k, d, u_all[f, :len(k)] if use_ur else None, bounds, views[f, :int(npts[f])], True,
                                                           occ_in[f, :len(k)])
            assert nm[f] == on and np.array_equal(match[f, :len(k)], om) and np.array_equal(occ[f, :len(k)], oo), (use_ur, f)
            total += on
        nm2, match2, occ2 = m.SearchByProjectionFrameBatch(ex, 0, F, bounds, views, npts, occ_in, stereo_pair0=0 if use_ur else -1)
        assert np.array_equal(nm, nm2) and np.array_equal(match, match2) and np.array_equal(occ, occ2)
    assert total > 600, total
    with pytest.raises(orbx.OrbxError):
        orbx.ORBmatcher(0.9, True).SearchByProjectionFrameBatchDevice(ex, 0, F + 1, bounds)


def test_oracle_last_frame_projection_against_a_float64_model(oracle):
    """The oracle's restatement of the projection block (Sophus' quaternion sandwich in float) against an independent model: the
    rotation MATRIX of the same quaternion, everything in float64.  Coordinates agree to float rounding, the gates wherever the
    float64 value is not within 1e-4 of a threshold, the level windows follow the direction flag, invalid records are zeroed."""
    rng = np.random.default_rng(5)
    w, h, fx = 640.0, 480.0, 420.0
    n = 4000
    R = _rot(0.02, -0.015, 0.03)
    t = np.array([0.04, -0.02, 0.1])
    q = _quat(R)
    pose = np.concatenate([q, t, [fx, fx, w / 2, h / 2, 0.12 * fx]]).astype(np.float32)
    pc = np.stack([rng.uniform(-12, 12, n), rng.uniform(-9, 9, n), rng.uniform(-2, 20, n)], 1)
    pw = ((pc - t) @ R).astype(np.float32)
    octv = rng.integers(0, 8, n).astype(np.int32)
    ang = rng.uniform(0, 360, n).astype(np.float32)
    flags = ((rng.random(n) < 0.9).astype(np.uint8)) | ((rng.random(n) < 0.5).astype(np.uint8) << 1)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    q64, t64 = pose[:4].astype(np.float64), pose[4:7].astype(np.float64)
    x, y, z, ww = q64
    Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                   [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                   [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])
    X = pw.astype(np.float64) @ Rq.T + t64
    with np.errstate(divide="ignore", invalid="ignore"):
        u = fx * X[:, 0] / X[:, 2] + w / 2
        v = fx * X[:, 1] / X[:, 2] + h / 2
    for direction in (0, 1, 2):
        views, _ = oracle.project_last_frame(pose, direction, pw, octv, ang, flags, desc, 7.0, sf, (0.0, 0.0, w, h))
        want = ((flags & 1) != 0) & (X[:, 2] > 0) & (u >= 0) & (u <= w) & (v >= 0) & (v <= h)
        near = (np.abs(X[:, 2]) < 1e-4) | (np.abs(u) < 1e-3) | (np.abs(u - w) < 1e-3) | (np.abs(v) < 1e-3) | (np.abs(v - h) < 1e-3)
        ok = views["valid"] != 0
        assert np.array_equal(ok[~near], want[~near])
        assert ok.sum() > 500
        assert (np.abs(views["u"][ok] - u[ok]) < 2e-3).all() and (np.abs(views["v"][ok] - v[ok]) < 2e-3).all()
        assert (np.abs(views["ur"][ok] - (u[ok] - 0.12 * fx / X[ok, 2])) < 2e-3).all()
        assert np.array_equal(views["radius"][ok], (np.float32(7.0) * sf[octv[ok]]).astype(np.float32))
        lo = {0: octv - 1, 1: octv, 2: np.zeros_like(octv)}[direction]
        hi = {0: octv + 1, 1: np.full_like(octv, -1), 2: octv}[direction]
        assert np.array_equal(views["min_level"][ok], lo[ok]) and np.array_equal(views["max_level"][ok], hi[ok])
        bad = ~ok
        assert not views["u"][bad].any() and not views["radius"][bad].any() and not views["min_level"][bad].any()
        assert np.array_equal(views["angle"], ang) and np.array_equal(views["has_observations"], (flags >> 1) & 1)
        assert np.array_equal(views["desc"], desc)


@pytest.mark.gpu
def test_device_projection_for_fisheye_frames_and_matcher_against_the_oracle(oracle):
    """Round 6: Frame::isInFrustum of stereo-fisheye frames (isInFrustumChecks per camera with KannalaBrandt8::project,
    src/Frame.cc:689-697, 1333-1410) on the device, both cameras' views consumed in place by the batched fisheye matcher
    (src/ORBmatcher.cc:41-221 with Nleft != -1).  Parity: per camera, gate decisions equal the oracle's restatement unless the oracle
    reports the decisive quantity within 1e-4 of its threshold (device atan2f / cosf / sinf), coordinates within 2e-4 relative,
    levels equal unless log(ratio) / logScaleFactor is within 1e-4 of an integer; the MATCHER is exact: the device-view call equals
    the host-view batched entry fed with the downloaded views, which equals the oracle's matcher on them."""
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    w, h, nf, F = 640, 480, 800, 3
    rng = np.random.default_rng(99)
    pairs = [synth.stereo_pair(w, h, 180 + f) for f in range(F)]
    dev = DeviceBuffer.from_numpy(np.stack([p[0] for p in pairs] + [p[1] for p in pairs]))
    ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2 * F)
    ex.extract_batch_device(dev.ptr.value, 2 * F, w, h, w, w * h)
    ex.sync()
    cap = ex.capacity
    kb1 = np.array([190.97, 190.97, w / 2 + 1.3, h / 2 - 0.8, 0.0034, 0.00071, -0.0020, 0.00020], np.float32)
    kb2 = np.array([190.44, 190.44, w / 2 - 2.1, h / 2 + 1.1, 0.0034, 0.00071, -0.0020, 0.00020], np.float32)
    Rrl, trl = _rot(0.004, -0.006, 0.002), np.array([-0.101, 0.002, 0.001])
    framesL = [ex.download(f)[1:] for f in range(F)]
    framesR = [ex.download(F + f)[1:] for f in range(F)]

    def unproject(kb, u, v):   # rough inverse of the KB8 projection (theta from r by a few Newton steps), float64: scene construction only
        x, y = (u - kb[2]) / kb[0], (v - kb[3]) / kb[1]
        r = np.hypot(x, y)
        th = r.copy()
        for _ in range(8):
            t2 = th * th
            fth = th * (1 + kb[4] * t2 + kb[5] * t2 ** 2 + kb[6] * t2 ** 3 + kb[7] * t2 ** 4) - r
            dth = 1 + 3 * kb[4] * t2 + 5 * kb[5] * t2 ** 2 + 7 * kb[6] * t2 ** 3 + 9 * kb[7] * t2 ** 4
            th = th - fth / dth
        s = np.where(r > 1e-9, np.tan(th) / np.maximum(r, 1e-9), 1.0)
        return np.stack([x * s, y * s, np.ones_like(x)], 1)

    posesL, posesR, Rs, ts = [], [], [], []
    for f in range(F):
        R = _rot(*(rng.normal(0, 0.01, 3)))
        t = rng.normal(0, 0.05, 3)
        Ow = -R.T @ t
        Rr, tr = Rrl @ R, Rrl @ t + trl
        twc_r = R.T @ (-Rrl.T @ trl) + Ow       # mRwc * mTlr.translation() + mOw
        posesL.append(np.concatenate([R.reshape(-1), t, Ow, kb1]).astype(np.float32))
        posesR.append(np.concatenate([Rr.reshape(-1), tr, twc_r, kb2]).astype(np.float32))
        Rs.append(R), ts.append(t)
    pos, nrm, mind, maxd, desc, flags = [], [], [], [], [], []
    for f in range(F):
        for (k, d), kb, Rc, tc in ((framesL[f], kb1, Rs[f], ts[f]), (framesR[f], kb2, Rrl @ Rs[f], Rrl @ ts[f] + trl)):
            take = rng.choice(len(k), size=min(180, len(k)), replace=False)
            rays = unproject(kb.astype(np.float64), k["x"][take] + rng.normal(0, 1.5, len(take)), k["y"][take] + rng.normal(0, 1.5, len(take)))
            for j, i in enumerate(take):
                z = rng.uniform(2.0, 30.0)
                P = Rc.T @ (rays[j] * z - tc)
                pos.append(P)
                v = P - (-Rs[f].T @ ts[f])
                dist = np.linalg.norm(v)
                n_ = v / dist + rng.normal(0, 0.25, 3)
                nrm.append(n_ / np.linalg.norm(n_))
                sc = 1.2 ** int(k["octave"][i])
                maxd.append(dist * sc * rng.uniform(0.95, 1.05))
                mind.append(maxd[-1] / 1.2 ** 7)
                desc.append(d[i] ^ np.packbits(rng.random((32, 8)) < 0.03, axis=1).reshape(32))
                flags.append((0 if rng.random() > 0.04 else 1) | (2 if rng.random() < 0.9 else 0))
    for _ in range(250):
        P = rng.normal(0, 8.0, 3)
        pos.append(P)
        n_ = rng.normal(0, 1, 3)
        nrm.append(n_ / np.linalg.norm(n_))
        m = rng.uniform(0.5, 40.0)
        maxd.append(m)
        mind.append(m / 3.5)
        desc.append(rng.integers(0, 256, 32, dtype=np.uint8))
        flags.append(2)
    pos, nrm = np.array(pos, np.float32), np.array(nrm, np.float32)
    mind, maxd = np.array(mind, np.float32), np.array(maxd, np.float32)
    desc, flags = np.array(desc, np.uint8), np.array(flags, np.uint8)
    n = len(pos)
    bounds = (0.0, 0.0, float(w), float(h))
    ex.map_upload(pos, nrm, mind, maxd, desc, flags)
    vl, vr = ex.project_map_points_fisheye(np.stack(posesL), np.stack(posesR), bounds, 0.5, None, want_views=True)
    logsf = np.float32(np.log(np.float32(1.2)))
    seen = [0, 0]
    for f in range(F):
        ol, ml = oracle.is_in_frustum_kb8(posesL[f], pos, nrm, mind, maxd, bounds, 0.5, logsf, 8, flags, desc)
        orr, mr = oracle.is_in_frustum_kb8(posesR[f], pos, nrm, mind, maxd, bounds, 0.5, logsf, 8, flags, desc)
        assert np.array_equal(vl[f]["desc"], ol["desc"]) and np.array_equal(vl[f]["bad"], ol["bad"])
        for cam, (inv, px, py, vc, lv, ov, mg) in enumerate((
                (vl[f]["in_view"], vl[f]["proj_x"], vl[f]["proj_y"], vl[f]["view_cos"], vl[f]["predicted_level"], ol, ml),
                (vr[f]["in_view_r"], vl[f]["proj_xr"], vr[f]["proj_yr"], vr[f]["view_cos_r"], vr[f]["predicted_level_r"], orr, mr))):
            diff = (inv != 0) != (ov["in_view"] != 0)
            assert (mg[diff, 0] < 1e-4).all(), (f, cam, mg[diff])
            both = (inv != 0) & (ov["in_view"] != 0)
            seen[cam] += int(both.sum())
            for got, name in ((px, "proj_x"), (py, "proj_y"), (vc, "view_cos")):
                a_, b_ = got[both].astype(np.float64), ov[name][both].astype(np.float64)
                assert (np.abs(a_ - b_) <= 2e-4 * np.maximum(1.0, np.abs(b_))).all(), (cam, name)
            ld = both & (lv != ov["predicted_level"])
            assert (mg[ld, 1] < 1e-4).all()
        assert (vr[f]["predicted_level_r"][vr[f]["in_view_r"] == 0] == -1).all()
    assert seen[0] > 300 and seen[1] > 300, seen
    # the matcher: device views == host-view batched entry on the downloaded views == the oracle on them
    l2r, r2l = np.full((F, cap), -1, np.int32), np.full((F, cap), -1, np.int32)
    for f in range(F):
        idx, dist, ok = oracle.bf_knn2(framesL[f][1], framesR[f][1])
        lr = np.where(ok.astype(bool) & (rng.random(len(idx)) < 0.8), idx[:, 0], -1).astype(np.int32)
        l2r[f, :len(lr)] = lr
        for i in np.nonzero(lr >= 0)[0]:
            r2l[f, lr[i]] = i
    occ = (rng.random((F, 2 * cap)) < 0.04).astype(np.uint8)
    m = orbx.ORBmatcher(0.8, True)
    total = 0
    for th, far in ((3.0, True), (1.0, False)):
        nm, match, oc = m.SearchByProjectionFisheyeBatchDevice(ex, 0, F, F, bounds, l2r, r2l, occ, th, far, 60.0)
        nm2, match2, oc2 = m.SearchByProjectionFisheyeBatch(ex, 0, F, F, bounds, vl, vr, np.full(F, n, np.int32), l2r, r2l, occ, th, far, 60.0)
        assert np.array_equal(nm, nm2) and np.array_equal(match, match2) and np.array_equal(oc, oc2)
        sf = ex.GetScaleFactors()
        for f in range(F):
            kL, dL = framesL[f]
            kR, dR = framesR[f]
            nL, nn = len(kL), len(kL) + len(kR)
            occf = occ[f, :nn]                      # rows are [left keypoints | right keypoints] at Nleft, like the one-shot arrays
            exp = oracle.search_by_projection_fisheye(np.concatenate([kL, kR]), np.concatenate([dL, dR]), nL, bounds, sf,
                                                      vl[f].view(oracle.MP_DTYPE), vr[f].view(oracle.MPR_DTYPE), th, far, 60.0, 0.8,
                                                      l2r[f, :nL], r2l[f, :len(kR)], occf)
            got_match, got_occ = match[f, :nn], oc[f, :nn]
            assert nm[f] == exp[0] and np.array_equal(got_match, exp[1]) and np.array_equal(got_occ, exp[2]), (f, th)
            total += int(nm[f])
    assert total > 150, total


def test_oracle_kb8_frustum_against_a_float64_model(oracle):
    """The oracle's isInFrustumChecks restatement with KannalaBrandt8::project against the same formulas in float64 (numpy):
    gates equal away from their thresholds, projections / viewing cosine / depth to float rounding, PredictScale's level equal
    unless the float64 quotient is within 1e-4 of an integer."""
    rng = np.random.default_rng(12)
    w, h, n = 640.0, 480.0, 5000
    kb = np.array([190.97, 190.97, w / 2 + 1.3, h / 2 - 0.8, 0.0034, 0.00071, -0.0020, 0.00020])
    R, t = _rot(0.02, -0.01, 0.03), np.array([0.03, -0.02, 0.05])
    Ow = -R.T @ t
    pose = np.concatenate([R.reshape(-1), t, Ow, kb]).astype(np.float32)
    P = rng.normal(0, 6.0, (n, 3)).astype(np.float32)
    Pn = rng.normal(0, 1, (n, 3))
    Pn = (Pn / np.linalg.norm(Pn, axis=1)[:, None]).astype(np.float32)
    maxd = rng.uniform(2, 40, n).astype(np.float32)
    mind = (maxd / 3.5).astype(np.float32)
    flags = np.full(n, 2, np.uint8)
    desc = np.zeros((n, 32), np.uint8)
    logsf = np.float32(np.log(np.float32(1.2)))
    views, mg = oracle.is_in_frustum_kb8(pose, P, Pn, mind, maxd, (0.0, 0.0, w, h), 0.5, logsf, 8, flags, desc)
    R64, t64, O64 = pose[:9].astype(np.float64).reshape(3, 3), pose[9:12].astype(np.float64), pose[12:15].astype(np.float64)
    Pc = P.astype(np.float64) @ R64.T + t64
    th = np.arctan2(np.hypot(Pc[:, 0], Pc[:, 1]), Pc[:, 2])
    psi = np.arctan2(Pc[:, 1], Pc[:, 0])
    r = th + kb[4] * th ** 3 + kb[5] * th ** 5 + kb[6] * th ** 7 + kb[7] * th ** 9
    u, v = kb[0] * r * np.cos(psi) + kb[2], kb[1] * r * np.sin(psi) + kb[3]
    PO = P.astype(np.float64) - O64
    dist = np.linalg.norm(PO, axis=1)
    vc = (PO * Pn.astype(np.float64)).sum(1) / dist
    ok = (Pc[:, 2] >= 0) & (u >= 0) & (u <= w) & (v >= 0) & (v <= h) & (dist >= 0.8 * mind) & (dist <= 1.2 * maxd) & (vc >= 0.5)
    clear = mg[:, 0] > 1e-4
    got = views["in_view"] != 0
    assert np.array_equal(got[clear], ok[clear]) and got.sum() > 100
    assert (np.abs(views["proj_x"][got] - u[got]) < 2e-3).all() and (np.abs(views["proj_y"][got] - v[got]) < 2e-3).all()
    assert (np.abs(views["view_cos"][got] - vc[got]) < 1e-5).all()
    assert (np.abs(views["track_depth"][got] - np.linalg.norm(Pc, axis=1)[got]) < 1e-4).all()
    q = np.log(maxd.astype(np.float64) / dist) / np.float64(logsf)
    lvl = np.clip(np.ceil(q), 0, 7).astype(np.int32)
    sure = got & (np.abs(q - np.rint(q)) > 1e-4)
    assert np.array_equal(views["predicted_level"][sure], lvl[sure])
