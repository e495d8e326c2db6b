"""Widening past SURVEY 8f: the next consumers of the device-resident descriptors.
  ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)   src/ORBmatcher.cc:1808-1918 (relocalisation)
  ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse)   src/ORBmatcher.cc:886-1106
CPU part: literal Python transcriptions (numpy float32 where the reference computes in float) against the C++ oracle.
GPU part (-m gpu): the HIP kernels through the C ABI against the oracle, bit for bit."""
import bisect

import numpy as np
import pytest

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
from test_oracle_matchers import Grid, HISTO, TH_LOW, f32, hamming, rot_bin, three_maxima


# ---- literal transcriptions -------------------------------------------------------------------------------------------------
def search_by_projection_keyframe_py(kps, desc, bounds, pts, orb_dist, check_ori, occupied):
    grid = Grid(kps, bounds)
    mvp = [-2 if o else -1 for o in occupied]  # CurrentFrame.mvpMapPoints: -1 NULL, -2 there before, >= 0 pKF's point
    hist = [[] for _ in range(HISTO)]
    nmatches = 0
    for i in range(len(pts)):
        p = pts[i]
        if not p["valid"]:
            continue
        cand = grid.area(p["u"], p["v"], p["radius"], int(p["min_level"]), int(p["max_level"]))
        if not cand:
            continue
        best, best_idx = 256, -1
        for i2 in cand:
            if mvp[i2] != -1:
                continue
            d = hamming(p["desc"], desc[i2])
            if d < best:
                best, best_idx = d, i2
        if best <= orb_dist:
            mvp[best_idx] = i
            nmatches += 1
            if check_ori:
                hist[rot_bin(p["angle"], kps["angle"][best_idx])].append(best_idx)
    if check_ori:
        keep = three_maxima(hist)
        for b in range(HISTO):
            if b in keep:
                continue
            for idx in hist[b]:
                mvp[idx] = -1
                nmatches -= 1
    match = np.array([m if m >= 0 else -1 for m in mvp], np.int32)
    occ = np.array([m != -1 for m in mvp], np.uint8)
    return nmatches, match, occ


def epipolar_constrain_py(kp1, kp2, F, unc):
    """Pinhole::epipolarConstrain, src/CameraModels/Pinhole.cpp:136-148 (F = F12 row-major, float arithmetic)."""
    x1, y1, x2, y2 = f32(kp1["x"]), f32(kp1["y"]), f32(kp2["x"]), f32(kp2["y"])
    a = x1 * F[0] + y1 * F[3] + F[6]
    b = x1 * F[1] + y1 * F[4] + F[7]
    c = x1 * F[2] + y1 * F[5] + F[8]
    num = a * x2 + b * y2 + c
    den = a * a + b * b
    if den == 0:
        return False
    dsqr = num * num / den
    return float(dsqr) < 3.84 * float(unc)


def search_for_triangulation_py(fv1, k1, d1, mp1, ur1, fv2, k2, d2, mp2, ur2, sf2, sg2, ep, F, only_stereo, coarse, check_ori):
    F = np.asarray(F, np.float32).reshape(9)
    m1 = {int(n): [int(x) for x in fv1[2][fv1[1][j]:fv1[1][j + 1]]] for j, n in enumerate(fv1[0])}   # std::map: ascending keys
    m2 = {int(n): [int(x) for x in fv2[2][fv2[1][j]:fv2[1][j + 1]]] for j, n in enumerate(fv2[0])}
    keys1, keys2 = sorted(m1), sorted(m2)
    matches12 = np.full(len(k1), -1, np.int32)
    matched2 = [False] * len(k2)   # vbMatched2: tested, never set
    hist = [[] for _ in range(HISTO)]
    nmatches = 0
    i, j = 0, 0
    while i < len(keys1) and j < len(keys2):
        if keys1[i] == keys2[j]:
            for idx1 in m1[keys1[i]]:
                if mp1[idx1]:
                    continue
                stereo1 = ur1 is not None and ur1[idx1] >= 0
                if only_stereo and not stereo1:
                    continue
                best, best_idx2 = TH_LOW, -1
                for idx2 in m2[keys2[j]]:
                    if matched2[idx2] or mp2[idx2]:
                        continue
                    stereo2 = ur2 is not None and ur2[idx2] >= 0
                    if only_stereo and not stereo2:
                        continue
                    dist = hamming(d1[idx1], d2[idx2])
                    if dist > TH_LOW or dist > best:
                        continue
                    kp2 = k2[idx2]
                    if not stereo1 and not stereo2:
                        ex, ey = f32(ep[0]) - f32(kp2["x"]), f32(ep[1]) - f32(kp2["y"])
                        if ex * ex + ey * ey < f32(100) * f32(sf2[kp2["octave"]]):
                            continue
                    if coarse or epipolar_constrain_py(k1[idx1], kp2, F, sg2[kp2["octave"]]):
                        best_idx2, best = idx2, dist
                if best_idx2 >= 0:
                    matches12[idx1] = best_idx2
                    nmatches += 1
                    if check_ori:
                        hist[rot_bin(k1["angle"][idx1], k2["angle"][best_idx2])].append(idx1)
            i += 1
            j += 1
        elif keys1[i] < keys2[j]:
            i = bisect.bisect_left(keys1, keys2[j])
        else:
            j = bisect.bisect_left(keys2, keys1[i])
    if check_ori:
        keep = three_maxima(hist)
        for b in range(HISTO):
            if b in keep:
                continue
            for idx in hist[b]:
                matches12[idx] = -1
                nmatches -= 1
    return nmatches, matches12


def fuse_search_py(kps, desc, ur, bounds, inv_sigma2, pts, max_dist=TH_LOW):
    """ORBmatcher::Fuse, src/ORBmatcher.cc:1195-1256 (+ KeyFrame::GetFeaturesInArea, src/KeyFrame.cc:705-749)."""
    grid = Grid(kps, bounds)
    best_idx = np.full(len(pts), -1, np.int32)
    best_dist = np.full(len(pts), 256, np.int32)
    nfused = 0
    for i in range(len(pts)):
        p = pts[i]
        if not p["valid"]:
            continue
        lvl = int(p["predicted_level"])
        cand = grid.area(p["u"], p["v"], p["radius"], -1, -1)
        if not cand:
            continue
        b, bi = 256, -1
        for idx in cand:
            kp = kps[idx]
            if kp["octave"] < lvl - 1 or kp["octave"] > lvl:
                continue
            ex, ey = f32(p["u"]) - f32(kp["x"]), f32(p["v"]) - f32(kp["y"])
            if ur is not None and ur[idx] >= 0:
                er = f32(p["ur"]) - f32(ur[idx])
                e2 = ex * ex + ey * ey + er * er
                if float(e2 * f32(inv_sigma2[kp["octave"]])) > 7.8:
                    continue
            else:
                e2 = ex * ex + ey * ey
                if float(e2 * f32(inv_sigma2[kp["octave"]])) > 5.99:
                    continue
            d = hamming(p["desc"], desc[idx])
            if d < b:
                b, bi = d, idx
        best_dist[i] = b
        if b <= max_dist:
            best_idx[i] = bi
            nfused += 1
    return nfused, best_idx, best_dist


def _fuse_points(mod, rng, f, th, noise):
    """Map points near the key frame's own keypoints (k2: position noise, a level off, flipped descriptor bits) plus the other
    frame's keypoints (k1: displaced by the stream's motion, mostly rejected by the chi-square gate)."""
    k1 = np.concatenate([f["k2"], f["k1"][::3]])
    d1 = np.concatenate([f["d2"], f["d1"][::3]])
    sf = f["sf"]
    n = len(k1)
    pts = np.zeros(n, mod.FP_DTYPE)
    pts["u"], pts["v"] = k1["x"] + rng.normal(0, noise, n), k1["y"] + rng.normal(0, noise, n)
    pts["ur"] = pts["u"] - rng.uniform(2, 40, n).astype(np.float32)
    lvl = np.clip(k1["octave"] + rng.integers(0, 2, n), 0, 7)
    pts["predicted_level"] = lvl
    pts["radius"] = (np.float32(th) * sf[lvl]).astype(np.float32)
    pts["valid"] = rng.random(n) < 0.85
    pts["desc"] = d1 ^ np.packbits(rng.random((n, 32, 8)) < 0.05, axis=2).reshape(n, 32)
    return pts


def search_by_projection_sim3_py(kps, desc, bounds, pts, pred_level, ratio_hamming, matched):
    """The loop-closing SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming), src/ORBmatcher.cc:406-503, from the
    window search on (:464-499): KeyFrame::GetFeaturesInArea without levels, `if (vpMatched[idx]) continue`, the level window
    [nPredictedLevel - 1, nPredictedLevel] inside the loop, `bestDist <= TH_LOW * ratioHamming` (float), no orientation check."""
    grid = Grid(kps, bounds)
    vp = [-2 if o else -1 for o in matched]
    nmatches = 0
    for i in range(len(pts)):
        p = pts[i]
        if not p["valid"]:
            continue
        cand = grid.area(p["u"], p["v"], p["radius"], -1, -1)
        if not cand:
            continue
        best, best_idx = 256, -1
        for idx in cand:
            if vp[idx] != -1:
                continue
            lvl = kps["octave"][idx]
            if lvl < pred_level[i] - 1 or lvl > pred_level[i]:
                continue
            d = hamming(p["desc"], desc[idx])
            if d < best:
                best, best_idx = d, idx
        if f32(best) <= f32(TH_LOW) * f32(ratio_hamming):
            vp[best_idx] = i
            nmatches += 1
    return nmatches, np.array([m if m >= 0 else -1 for m in vp], np.int32), np.array([m != -1 for m in vp], np.uint8)


def search_by_bow_keyframes_py(fv1, d1, ang1, valid1, fv2, d2, ang2, valid2, nnratio, check_ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12), src/ORBmatcher.cc:766-884."""
    m1 = {int(n): [int(x) for x in fv1[2][fv1[1][j]:fv1[1][j + 1]]] for j, n in enumerate(fv1[0])}
    m2 = {int(n): [int(x) for x in fv2[2][fv2[1][j]:fv2[1][j + 1]]] for j, n in enumerate(fv2[0])}
    keys1, keys2 = sorted(m1), sorted(m2)
    matches12 = np.full(len(d1), -1, np.int32)
    matched2 = [False] * len(d2)
    hist = [[] for _ in range(HISTO)]
    nmatches = 0
    i, j = 0, 0
    while i < len(keys1) and j < len(keys2):
        if keys1[i] == keys2[j]:
            for idx1 in m1[keys1[i]]:
                if not valid1[idx1]:
                    continue
                b1, bi, b2 = 256, -1, 256
                for idx2 in m2[keys2[j]]:
                    if matched2[idx2] or not valid2[idx2]:
                        continue
                    dist = hamming(d1[idx1], d2[idx2])
                    if dist < b1:
                        b2, b1, bi = b1, dist, idx2
                    elif dist < b2:
                        b2 = dist
                if b1 < TH_LOW:
                    if f32(b1) < f32(nnratio) * f32(b2):
                        matches12[idx1] = bi
                        matched2[bi] = True
                        if check_ori:
                            hist[rot_bin(ang1[idx1], ang2[bi])].append(idx1)
                        nmatches += 1
            i += 1
            j += 1
        elif keys1[i] < keys2[j]:
            i = bisect.bisect_left(keys1, keys2[j])
        else:
            j = bisect.bisect_left(keys2, keys1[i])
    if check_ori:
        keep = three_maxima(hist)
        for b in range(HISTO):
            if b in keep:
                continue
            for idx in hist[b]:
                matches12[idx] = -1
                nmatches -= 1
    return nmatches, matches12


# ---- inputs -----------------------------------------------------------------------------------------------------------------
def _frames(oracle, w, h, nf, stream):
    f0, f1 = synth.mono_frame(w, h, stream, 0), synth.mono_frame(w, h, stream, 2)
    ex = oracle.OracleExtractor(nf)
    _, k1, d1 = ex.extract(f0)
    _, k2, d2 = ex.extract(f1)
    t = ex.tables()
    return dict(w=w, h=h, k1=k1, d1=d1, k2=k2, d2=d2, sf=t["scale"], sigma2=(t["scale"] * t["scale"]).astype(np.float32),
                bounds=(0.0, 0.0, float(w), float(h)))


@pytest.fixture(scope="module")
def small(oracle):
    return _frames(oracle, 480, 360, 500, 310)


def _kf_points(mod, rng, f, th):
    """pKF's map points (key frame = frame 1 of the pair) projected into the current frame (= frame 2)."""
    k1, d1, sf = f["k1"], f["d1"], f["sf"]
    n = len(k1)
    pts = np.zeros(n, mod.PP_DTYPE)
    pts["u"], pts["v"] = k1["x"] + rng.normal(0, 3.0, n), k1["y"] + rng.normal(0, 3.0, n)
    lvl = np.clip(k1["octave"] + rng.integers(-1, 2, n), 0, 7)    # nPredictedLevel
    pts["radius"] = (np.float32(th) * sf[lvl]).astype(np.float32)
    pts["min_level"], pts["max_level"] = lvl - 1, lvl + 1
    pts["angle"] = k1["angle"]
    pts["valid"] = rng.random(n) < 0.8
    pts["has_observations"] = rng.random(n) < 0.5                 # must not matter
    pts["ur"] = rng.uniform(-50, 500, n)                          # must not matter
    pts["desc"] = d1 ^ np.packbits(rng.random((n, 32, 8)) < 0.06, axis=2).reshape(n, 32)
    return pts


def _feature_vector(desc, rng, nodes=48, drop=0.1, stop=0.03):
    """A FeatureVector as CSR: node = a hash of the descriptor's first bits (similar descriptors share nodes, like a vocabulary
    node at level L - levelsup), some node ids absent from one of the two frames, a few features in no node (stopped words)."""
    node = (desc[:, 0].astype(np.uint32) * 7 + (desc[:, 1] >> 6)) % nodes * 5 + 3
    keep = rng.random(len(desc)) >= stop
    gone = rng.choice(np.unique(node), max(1, int(drop * nodes)), replace=False)
    keep &= ~np.isin(node, gone)
    ids = np.unique(node[keep])
    start, feats = [0], []
    for nid in ids:
        idx = np.nonzero(keep & (node == nid))[0]
        feats.extend(idx.tolist())
        start.append(len(feats))
    return ids.astype(np.uint32), np.array(start, np.int32), np.array(feats, np.uint32)


def _tri_inputs(f, rng, mono=False):
    k1, d1, k2, d2 = f["k1"], f["d1"], f["k2"], f["d2"]
    # frame 2 of a synthetic stream is frame 0 shifted by a few pixels: the image motion of a camera translating parallel to the
    # image plane, F12 = [t]x with t along the shift (+ a perturbation so that all nine entries are exercised)
    dist = np.unpackbits(d1[:, None, :] ^ d2[None, :, :], axis=2).sum(2)
    j = dist.argmin(1)
    good = dist[np.arange(len(k1)), j] < 30
    dx = float(np.median((k2["x"][j] - k1["x"])[good])); dy = float(np.median((k2["y"][j] - k1["y"])[good]))
    nrm = max(1e-6, (dx * dx + dy * dy) ** 0.5)
    tx, ty = dx / nrm, dy / nrm
    F = np.array([[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]], np.float64) + rng.normal(0, 2e-5, (3, 3)) * [[1, 1, 300], [1, 1, 300], [300, 300, 1]]
    F = F.astype(np.float32)
    ep = np.array([f["w"] * 0.5, f["h"] * 0.45], np.float32)
    mp1 = (rng.random(len(k1)) < 0.25).astype(np.uint8)
    mp2 = (rng.random(len(k2)) < 0.25).astype(np.uint8)
    ur1 = None if mono else np.where(rng.random(len(k1)) < 0.5, k1["x"] - rng.uniform(0, 30, len(k1)), -1).astype(np.float32)
    ur2 = None if mono else np.where(rng.random(len(k2)) < 0.5, k2["x"] - rng.uniform(0, 30, len(k2)), -1).astype(np.float32)
    fv1, fv2 = _feature_vector(d1, rng), _feature_vector(d2, rng)
    return fv1, mp1, ur1, fv2, mp2, ur2, ep, F


TRI_MODES = [(False, False, True), (False, True, True), (True, False, True), (False, False, False)]   # onlyStereo, coarse, checkOri


# ---- CPU: transcription == oracle -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,th,orb_dist,ori", [(1, 10.0, 100, True), (2, 3.0, 64, True), (3, 10.0, 100, False)])
def test_python_search_by_projection_keyframe_matches_oracle(oracle, small, seed, th, orb_dist, ori):
    f = small
    rng = np.random.default_rng(seed)
    pts = _kf_points(oracle, rng, f, th)
    occ = (rng.random(len(f["k2"])) < 0.3).astype(np.uint8)    # relocalisation: the PnP inliers already hold their points
    e = search_by_projection_keyframe_py(f["k2"], f["d2"], f["bounds"], pts, orb_dist, ori, occ)
    o = oracle.search_by_projection_keyframe(f["k2"], f["d2"], f["bounds"], pts, orb_dist, ori, occ)
    assert e[0] == o[0] and np.array_equal(e[1], o[1]) and np.array_equal(e[2], o[2]) and o[0] > 20
    # what distinguishes this flavour: a keypoint is taken at most once, pre-occupied ones never, culled ones are free again
    taken = o[1] >= 0
    assert not (taken & (occ != 0)).any() and np.array_equal(o[2] != 0, taken | (occ != 0))
    assert len(np.unique(o[1][taken])) == taken.sum() == o[0]


@pytest.mark.parametrize("seed,mono", [(4, False), (5, True)])
def test_python_search_for_triangulation_matches_oracle(oracle, small, seed, mono):
    f = small
    rng = np.random.default_rng(seed)
    fv1, mp1, ur1, fv2, mp2, ur2, ep, F = _tri_inputs(f, rng, mono)
    total = 0
    for only_stereo, coarse, ori in TRI_MODES:
        e = search_for_triangulation_py(fv1, f["k1"], f["d1"], mp1, ur1, fv2, f["k2"], f["d2"], mp2, ur2, f["sf"], f["sigma2"], ep, F,
                                        only_stereo, coarse, ori)
        o = oracle.search_for_triangulation(fv1, f["k1"], f["d1"], mp1, ur1, fv2, f["k2"], f["d2"], mp2, ur2, f["sf"], f["sigma2"],
                                            ep, F, only_stereo, coarse, ori)
        assert e[0] == o[0] and np.array_equal(e[1], o[1]) and o[0] == (o[1] >= 0).sum()
        assert not (mp1[o[1] >= 0]).any() and not mp2[o[1][o[1] >= 0]].any()
        if only_stereo:
            assert mono and o[0] == 0 or (not mono and (ur1[o[1] >= 0] >= 0).all())
        total += o[0]
    assert total > 60


def test_triangulation_epipolar_gate_is_active(oracle, small):
    """The epipolar test must actually reject: coarse (no test) finds strictly more than the fine search, and a wrong F almost none."""
    f = small
    rng = np.random.default_rng(6)
    fv1, mp1, ur1, fv2, mp2, ur2, ep, F = _tri_inputs(f, rng, True)
    args = (fv1, f["k1"], f["d1"], mp1, ur1, fv2, f["k2"], f["d2"], mp2, ur2, f["sf"], f["sigma2"], ep)
    fine = oracle.search_for_triangulation(*args, F, False, False, False)[0]
    coarse = oracle.search_for_triangulation(*args, F, False, True, False)[0]
    wrong = oracle.search_for_triangulation(*args, np.ascontiguousarray(F.T[::-1]) * 50, False, False, False)[0]
    assert coarse > fine > 30 and wrong < fine // 2


@pytest.mark.parametrize("seed,th,stereo", [(7, 3.0, True), (8, 5.0, False)])
def test_python_fuse_search_matches_oracle(oracle, small, seed, th, stereo):
    f = small
    rng = np.random.default_rng(seed)
    pts = _fuse_points(oracle, rng, f, th, 1.0)
    k2 = f["k2"]
    ur = np.where(rng.random(len(k2)) < 0.5, k2["x"] - rng.uniform(2, 40, len(k2)), -1).astype(np.float32) if stereo else None
    inv = (1.0 / f["sigma2"]).astype(np.float32)
    e = fuse_search_py(k2, f["d2"], ur, f["bounds"], inv, pts)
    o = oracle.fuse_search(k2, f["d2"], ur, f["bounds"], inv, pts)
    assert e[0] == o[0] and np.array_equal(e[1], o[1]) and np.array_equal(e[2], o[2]) and o[0] > 15
    assert o[0] == (o[1] >= 0).sum() and ((o[2] <= TH_LOW) == (o[1] >= 0)).all()
    # several map points may fuse into one keypoint (the search keeps no occupancy): the bookkeeping that resolves it is the caller's
    assert stereo or len(np.unique(o[1][o[1] >= 0])) <= o[0]


@pytest.mark.parametrize("seed,ratio", [(9, 1.0), (10, 0.75)])
def test_loop_closing_flavours_are_served_by_the_same_entries(oracle, small, seed, ratio):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) (src/ORBmatcher.cc:406-503; its vpMatchedKF twin :505-612
    has the same loop) is the relocalisation entry with levels (nPredictedLevel - 1, nPredictedLevel), no orientation check and
    ORBdist = floor(TH_LOW * ratioHamming); Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1279-1390) is the Fuse search without
    the chi-square gate (inv_level_sigma2 = 0).  Checked here on the oracle against literal transcriptions of those loops."""
    f = small
    rng = np.random.default_rng(seed)
    pts = _kf_points(oracle, rng, f, 4.0)
    pred = np.asarray(pts["max_level"] - 1)                   # _kf_points stores (lvl - 1, lvl + 1)
    pts["max_level"] = pred                                   # -> the window [pred - 1, pred]
    occ = (rng.random(len(f["k2"])) < 0.2).astype(np.uint8)   # vpMatched[idx] != NULL
    e = search_by_projection_sim3_py(f["k2"], f["d2"], f["bounds"], pts, pred, ratio, occ)
    o = oracle.search_by_projection_keyframe(f["k2"], f["d2"], f["bounds"], pts, int(np.floor(np.float32(TH_LOW) * np.float32(ratio))),
                                             False, occ)
    assert e[0] == o[0] and np.array_equal(e[1], o[1]) and np.array_equal(e[2], o[2]) and o[0] > 10
    fp = _fuse_points(oracle, rng, f, 4.0, 2.5)
    zero = np.zeros(8, np.float32)
    inv = (1.0 / f["sigma2"]).astype(np.float32)
    a = oracle.fuse_search(f["k2"], f["d2"], None, f["bounds"], zero, fp)
    b = fuse_search_py(f["k2"], f["d2"], None, f["bounds"], zero, fp)
    g = oracle.fuse_search(f["k2"], f["d2"], None, f["bounds"], inv, fp)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[0] > g[0] > 10                                   # the gate was really switched off


def _sim3_inputs(mod, rng, f):
    """Map points of two key frames projected into each other: feature i1 of KF1 (k1) lands near the KF2 feature that observes the
    same scene point (found here by descriptor, displaced by the stream's motion) plus noise, and vice versa."""
    k1, d1, k2, d2, sf = f["k1"], f["d1"], f["k2"], f["d2"], f["sf"]
    dist = np.unpackbits(d1[:, None, :] ^ d2[None, :, :], axis=2).sum(2)
    j12, j21 = dist.argmin(1), dist.argmin(0)

    def mk(ka, da, kb, j, n):
        pts = np.zeros(n, mod.FP_DTYPE)
        pts["u"] = kb["x"][j] + rng.normal(0, 1.5, n)
        pts["v"] = kb["y"][j] + rng.normal(0, 1.5, n)
        lvl = np.clip(ka["octave"] + rng.integers(0, 2, n), 0, 7)
        pts["predicted_level"] = lvl
        pts["radius"] = (np.float32(7.5) * sf[lvl]).astype(np.float32)
        pts["valid"] = rng.random(n) < 0.7           # has a good, not yet matched map point inside the other image
        pts["desc"] = da ^ np.packbits(rng.random((n, 32, 8)) < 0.03, axis=2).reshape(n, 32)
        return pts
    return mk(k1, d1, k2, j12, len(k1)), mk(k2, d2, k1, j21, len(k2))


def search_by_sim3_py(k1, d1, b1, k2, d2, b2, p12, p21):
    """src/ORBmatcher.cc:1437-1583: two searches (no chi-square gate, TH_HIGH) and the agreement check."""
    z = np.zeros(8, np.float32)
    _, m1, _ = fuse_search_py(k2, d2, None, b2, z, p12, 100)
    _, m2, _ = fuse_search_py(k1, d1, None, b1, z, p21, 100)
    out = np.full(len(m1), -1, np.int32)
    n = 0
    for i1 in range(len(m1)):
        idx2 = m1[i1]
        if idx2 >= 0 and m2[idx2] == i1:
            out[i1] = idx2
            n += 1
    return n, out


def test_python_search_by_sim3_matches_oracle(oracle, small):
    f = small
    rng = np.random.default_rng(11)
    p12, p21 = _sim3_inputs(oracle, rng, f)
    z = np.zeros(8, np.float32)
    en, em = search_by_sim3_py(f["k1"], f["d1"], f["bounds"], f["k2"], f["d2"], f["bounds"], p12, p21)
    _, m1, _ = oracle.fuse_search(f["k2"], f["d2"], None, f["bounds"], z, p12, 100)
    _, m2, _ = oracle.fuse_search(f["k1"], f["d1"], None, f["bounds"], z, p21, 100)
    ok = (m1 >= 0)
    ok[ok] = m2[m1[ok]] == np.arange(len(m1))[ok]
    assert en == ok.sum() > 40 and np.array_equal(em, np.where(ok, m1, -1))
    assert (m1 >= 0).sum() > en                      # the agreement check really drops one-sided matches


@pytest.mark.parametrize("seed,ratio,ori", [(12, 0.75, True), (13, 0.9, False)])
def test_python_search_by_bow_keyframes_matches_oracle(oracle, small, seed, ratio, ori):
    f = small
    rng = np.random.default_rng(seed)
    fv1, fv2 = _feature_vector(f["d1"], rng, nodes=12), _feature_vector(f["d2"], rng, nodes=12)   # few nodes: long lists, contention
    v1, v2 = (rng.random(len(f["k1"])) < 0.8).astype(np.uint8), (rng.random(len(f["k2"])) < 0.8).astype(np.uint8)
    e = search_by_bow_keyframes_py(fv1, f["d1"], f["k1"]["angle"], v1, fv2, f["d2"], f["k2"]["angle"], v2, ratio, ori)
    o = oracle.search_by_bow_keyframes(fv1, f["d1"], f["k1"]["angle"], v1, fv2, f["d2"], f["k2"]["angle"], v2, ratio, ori)
    assert e[0] == o[0] and np.array_equal(e[1], o[1]) and o[0] > 30
    got = o[1][o[1] >= 0]
    assert len(np.unique(got)) == len(got) and v1[o[1] >= 0].all() and v2[got].all()   # vbMatched2: a feature of pKF2 is taken once


# ---- GPU: HIP == oracle through the C ABI -----------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu():
    if orbx.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")
    return True


@pytest.fixture(scope="module")
def big(oracle):
    return _frames(oracle, 752, 480, 1500, 72)


@pytest.mark.gpu
def test_gpu_search_by_projection_keyframe(gpu, oracle, big):
    f = big
    rng = np.random.default_rng(21)
    for th, orb_dist, ori, pocc, least in [(10.0, 100, True, 0.3, 100), (3.0, 64, True, 0.6, 1), (10.0, 100, False, 0.0, 100),
                                           (25.0, 255, True, 0.1, 100), (10.0, 0, True, 0.0, 0)]:
        pts = _kf_points(orbx, rng, f, th)
        occ = (rng.random(len(f["k2"])) < pocc).astype(np.uint8)
        m = orbx.ORBmatcher(0.9, ori)
        nm, match, o2 = m.SearchByProjectionKeyFrame(f["k2"], f["d2"], f["bounds"], pts, occ, orb_dist)
        onm, omatch, oocc = oracle.search_by_projection_keyframe(f["k2"], f["d2"], f["bounds"], pts, orb_dist, ori, occ)
        assert onm >= least
        assert nm == onm and np.array_equal(match, omatch) and np.array_equal(o2, oocc)
    # the loop-closing usage: window [pred - 1, pred], no orientation check, ORBdist = floor(TH_LOW * ratioHamming)
    pts = _kf_points(orbx, rng, f, 4.0)
    pts["max_level"] = pts["max_level"] - 1
    nm, match, o2 = orbx.ORBmatcher(0.9, False).SearchByProjectionKeyFrame(f["k2"], f["d2"], f["bounds"], pts, occ, 37)
    onm, omatch, oocc = oracle.search_by_projection_keyframe(f["k2"], f["d2"], f["bounds"], pts, 37, False, occ)
    assert nm == onm > 5 and np.array_equal(match, omatch) and np.array_equal(o2, oocc)
    m = orbx.ORBmatcher(0.9, True)
    nm, match, o2 = m.SearchByProjectionKeyFrame(f["k2"], f["d2"], f["bounds"], pts[:0], occ, 100)
    assert nm == 0 and (match == -1).all() and np.array_equal(o2, occ)
    with pytest.raises(orbx.OrbxError):
        m.SearchByProjectionKeyFrame(f["k2"], f["d2"], f["bounds"], pts, occ, 256)


@pytest.mark.gpu
def test_gpu_keyframe_flavour_under_serial_walk_and_small_capacity(gpu, oracle, big):
    """The relocalisation flavour through the fallback paths of the projection search (one-wave serial walk, capacity retry)."""
    import json, os, subprocess, sys
    code = r"""
import json, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import orb_slam3_fast_amd as orbx
from oracle import oracle_py as oracle
import test_reloc_triangulation as T
f = T._frames(oracle, 752, 480, 1500, 72)
rng = np.random.default_rng(33)
pts = T._kf_points(orbx, rng, f, 10.0)
occ = (rng.random(len(f["k2"])) < 0.3).astype(np.uint8)
a = orbx.ORBmatcher(0.9, True).SearchByProjectionKeyFrame(f["k2"], f["d2"], f["bounds"], pts, occ, 100)
b = oracle.search_by_projection_keyframe(f["k2"], f["d2"], f["bounds"], pts, 100, True, occ)
print(json.dumps(dict(ok=bool(a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])), n=int(b[0]))))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    for env in ({"ORBX_PROJ_SERIAL": "1"}, {"ORBX_PROJ_CAND_CAP": "64"}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        r = json.loads(out.stdout.strip().splitlines()[-1])
        assert r["ok"] and r["n"] > 100, (env, r)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,mono", [(41, False), (42, True)])
def test_gpu_search_for_triangulation(gpu, oracle, big, seed, mono):
    f = big
    rng = np.random.default_rng(seed)
    fv1, mp1, ur1, fv2, mp2, ur2, ep, F = _tri_inputs(f, rng, mono)
    total = 0
    for only_stereo, coarse, ori in TRI_MODES:
        m = orbx.ORBmatcher(0.6, ori)
        n, pairs, m12 = m.SearchForTriangulation(fv1, f["k1"], f["d1"], mp1, ur1, fv2, f["k2"], f["d2"], mp2, ur2, f["sf"], f["sigma2"], ep, F,
                                                 only_stereo, coarse)
        on, om12 = oracle.search_for_triangulation(fv1, f["k1"], f["d1"], mp1, ur1, fv2, f["k2"], f["d2"], mp2, ur2, f["sf"], f["sigma2"],
                                                   ep, F, only_stereo, coarse, ori)
        assert n == on and np.array_equal(m12, om12)
        assert len(pairs) == n and np.array_equal(pairs[:, 1], om12[pairs[:, 0]]) and (np.diff(pairs[:, 0]) > 0).all()
        total += on
    assert total > 200


@pytest.mark.gpu
def test_gpu_search_for_triangulation_edges(gpu, oracle, big):
    f = big
    rng = np.random.default_rng(43)
    fv1, mp1, ur1, fv2, mp2, ur2, ep, F = _tri_inputs(f, rng, True)
    m = orbx.ORBmatcher(0.6, True)
    empty = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.uint32))
    common = (f["sf"], f["sigma2"], ep, F)
    n, pairs, m12 = m.SearchForTriangulation(empty, f["k1"], f["d1"], mp1, None, fv2, f["k2"], f["d2"], mp2, None, *common)
    assert n == 0 and len(pairs) == 0 and (m12 == -1).all()
    n, pairs, m12 = m.SearchForTriangulation(fv1, f["k1"], f["d1"], mp1, None, empty, f["k2"], f["d2"], mp2, None, *common)
    assert n == 0 and (m12 == -1).all()
    # every feature of pKF2 already holds a map point: nothing to pair; no feature of pKF1 is free: nothing either
    n, _, _ = m.SearchForTriangulation(fv1, f["k1"], f["d1"], mp1, None, fv2, f["k2"], f["d2"], np.ones_like(mp2), None, *common)
    assert n == 0
    n, _, _ = m.SearchForTriangulation(fv1, f["k1"], f["d1"], np.ones_like(mp1), None, fv2, f["k2"], f["d2"], mp2, None, *common)
    assert n == 0
    # disjoint node sets (the lower_bound walk never meets)
    a = (fv1[0] * 2, fv1[1], fv1[2])
    b = (fv2[0] * 2 + 1, fv2[1], fv2[2])
    n, _, m12 = m.SearchForTriangulation(a, f["k1"], f["d1"], mp1, None, b, f["k2"], f["d2"], mp2, None, *common)
    on, om12 = oracle.search_for_triangulation(a, f["k1"], f["d1"], mp1, None, b, f["k2"], f["d2"], mp2, None, f["sf"], f["sigma2"], ep, F)
    assert n == on == 0 and np.array_equal(m12, om12)
    # one node holding every feature of both frames (the largest candidate lists the call can see)
    one1 = (np.array([9], np.uint32), np.array([0, len(f["k1"])], np.int32), np.arange(len(f["k1"]), dtype=np.uint32))
    one2 = (np.array([9], np.uint32), np.array([0, len(f["k2"])], np.int32), np.arange(len(f["k2"]), dtype=np.uint32))
    n, _, m12 = m.SearchForTriangulation(one1, f["k1"], f["d1"], mp1, None, one2, f["k2"], f["d2"], mp2, None, *common)
    on, om12 = oracle.search_for_triangulation(one1, f["k1"], f["d1"], mp1, None, one2, f["k2"], f["d2"], mp2, None, f["sf"], f["sigma2"],
                                               ep, F)
    assert n == on and on > 100 and np.array_equal(m12, om12)
    # argument checks: descending node ids, a feature index outside the frame
    bad = (fv1[0][::-1].copy(), fv1[1], fv1[2])
    with pytest.raises(orbx.OrbxError):
        m.SearchForTriangulation(bad, f["k1"], f["d1"], mp1, None, fv2, f["k2"], f["d2"], mp2, None, *common)
    bad = (fv1[0], fv1[1], np.where(np.arange(len(fv1[2])) == 3, len(f["k1"]), fv1[2]).astype(np.uint32))
    with pytest.raises(orbx.OrbxError):
        m.SearchForTriangulation(bad, f["k1"], f["d1"], mp1, None, fv2, f["k2"], f["d2"], mp2, None, *common)


@pytest.mark.gpu
def test_gpu_fuse_search(gpu, oracle, big):
    f = big
    k2 = f["k2"]
    inv = (1.0 / f["sigma2"]).astype(np.float32)
    rng = np.random.default_rng(61)
    m = orbx.ORBmatcher(0.6, True)
    for th, noise, stereo, least in [(3.0, 1.0, True, 50), (4.0, 0.7, False, 100), (25.0, 3.0, True, 20), (0.5, 0.2, False, 1)]:
        pts = _fuse_points(orbx, rng, f, th, noise)
        ur = np.where(rng.random(len(k2)) < 0.5, k2["x"] - rng.uniform(2, 40, len(k2)), -1).astype(np.float32) if stereo else None
        n, bi, bd = m.FuseSearch(k2, f["d2"], ur, f["bounds"], inv, pts)
        on, obi, obd = oracle.fuse_search(k2, f["d2"], ur, f["bounds"], inv, pts)
        assert on >= least, (th, on)
        assert n == on and np.array_equal(bi, obi) and np.array_equal(bd, obd)
    # Fuse(pKF, Scw, ...) of loop closing: the same search without the chi-square gate
    pts = _fuse_points(orbx, rng, f, 4.0, 2.5)
    zero = np.zeros(8, np.float32)
    n, bi, bd = m.FuseSearch(k2, f["d2"], None, f["bounds"], zero, pts)
    on, obi, obd = oracle.fuse_search(k2, f["d2"], None, f["bounds"], zero, pts)
    assert n == on > 100 and np.array_equal(bi, obi) and np.array_equal(bd, obd)
    # points outside the image bounds / windows that leave the grid, no points, no keypoints
    pts = _fuse_points(orbx, rng, f, 3.0, 1.0)
    pts["u"][::3] = -500
    pts["v"][1::3] = 5000
    pts["u"][2::7] = f["w"] - 0.01
    n, bi, bd = m.FuseSearch(k2, f["d2"], None, f["bounds"], inv, pts)
    on, obi, obd = oracle.fuse_search(k2, f["d2"], None, f["bounds"], inv, pts)
    assert n == on and np.array_equal(bi, obi) and np.array_equal(bd, obd)
    n, bi, bd = m.FuseSearch(k2, f["d2"], None, f["bounds"], inv, pts[:0])
    assert n == 0 and len(bi) == 0
    n, bi, bd = m.FuseSearch(k2[:0], f["d2"][:0], None, f["bounds"], inv, pts)
    assert n == 0 and (bi == -1).all() and (bd == 256).all()
    bad = k2.copy()
    bad["octave"][5] = 8
    with pytest.raises(orbx.OrbxError):
        m.FuseSearch(bad, f["d2"], None, f["bounds"], inv, pts)


@pytest.mark.gpu
def test_gpu_search_by_sim3(gpu, oracle, big):
    f = big
    rng = np.random.default_rng(71)
    p12, p21 = _sim3_inputs(orbx, rng, f)
    n, m12 = orbx.ORBmatcher(0.75, True).SearchBySim3(f["k1"], f["d1"], f["bounds"], f["k2"], f["d2"], f["bounds"], p12, p21)
    z = np.zeros(8, np.float32)
    _, m1, _ = oracle.fuse_search(f["k2"], f["d2"], None, f["bounds"], z, p12, 100)
    _, m2, _ = oracle.fuse_search(f["k1"], f["d1"], None, f["bounds"], z, p21, 100)
    ok = (m1 >= 0)
    ok[ok] = m2[m1[ok]] == np.arange(len(m1))[ok]
    assert n == ok.sum() > 150 and np.array_equal(m12, np.where(ok, m1, -1))


@pytest.mark.gpu
def test_gpu_search_by_bow_keyframes(gpu, oracle, big):
    f = big
    rng = np.random.default_rng(81)
    total = 0
    for nodes, ratio, ori, pv in [(48, 0.75, True, 0.8), (12, 0.9, True, 0.9), (3, 0.75, False, 1.0), (48, 0.6, True, 0.5)]:
        fv1, fv2 = _feature_vector(f["d1"], rng, nodes=nodes), _feature_vector(f["d2"], rng, nodes=nodes)
        v1, v2 = (rng.random(len(f["k1"])) < pv).astype(np.uint8), (rng.random(len(f["k2"])) < pv).astype(np.uint8)
        n, m = orbx.SearchByBoWKeyFrames(fv1, f["k1"], f["d1"], v1, fv2, f["k2"], f["d2"], v2, ratio, ori)
        on, om = oracle.search_by_bow_keyframes(fv1, f["d1"], f["k1"]["angle"], v1, fv2, f["d2"], f["k2"]["angle"], v2, ratio, ori)
        assert n == on and np.array_equal(m, om)
        total += on
    assert total > 300
    # one node with every feature: lists longer than one wave (the chunked best / second merge), heavy contention for pKF2's features
    one1 = (np.array([4], np.uint32), np.array([0, len(f["k1"])], np.int32), np.arange(len(f["k1"]), dtype=np.uint32))
    one2 = (np.array([4], np.uint32), np.array([0, len(f["k2"])], np.int32), np.arange(len(f["k2"]), dtype=np.uint32))
    v1, v2 = np.ones(len(f["k1"]), np.uint8), np.ones(len(f["k2"]), np.uint8)
    n, m = orbx.SearchByBoWKeyFrames(one1, f["k1"], f["d1"], v1, one2, f["k2"], f["d2"], v2, 0.9, True)
    on, om = oracle.search_by_bow_keyframes(one1, f["d1"], f["k1"]["angle"], v1, one2, f["d2"], f["k2"]["angle"], v2, 0.9, True)
    assert n == on > 100 and np.array_equal(m, om)
    empty = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.uint32))
    n, m = orbx.SearchByBoWKeyFrames(empty, f["k1"], f["d1"], v1, one2, f["k2"], f["d2"], v2)
    assert n == 0 and (m == -1).all()
    n, m = orbx.SearchByBoWKeyFrames(one1, f["k1"], f["d1"], v1, one2, f["k2"], f["d2"], np.zeros_like(v2))
    assert n == 0 and (m == -1).all()


# ---- two-camera rigs: SearchForTriangulation through KannalaBrandt8::epipolarConstrain (float: tolerance parity) ------------
def _rig_inputs(mod, seed):
    """One stereo-fisheye frame as BOTH key frames (features = left | right): a left feature and the right feature that observes the
    same point triangulate under Tlr = the rig's extrinsics (and right -> left under its inverse), while left-left / right-right
    pairs have T = identity, no parallax, and are rejected at the first gate -- all four (R12, t12, camera) selections are used."""
    sc = synth.fisheye_stereo_scene(seed, 700, 650, 100, 80)
    kk, dd = np.concatenate([sc["kL"], sc["kR"]]), np.concatenate([sc["dL"], sc["dR"]])
    nL = len(sc["kL"])
    R, t = sc["R12"].astype(np.float64), sc["t12"].astype(np.float64)
    rig = np.zeros((), mod.TRI_RIG_DTYPE)
    rig["cam"] = np.stack([sc["cam1"], sc["cam2"], sc["cam1"], sc["cam2"]])
    rig["precision"] = 1e-6
    eye = np.eye(3)
    rig["R"] = np.stack([eye.ravel(), R.ravel(), R.T.ravel(), eye.ravel()]).astype(np.float32)        # ll, lr, rl, rr
    rig["t"] = np.stack([np.zeros(3), t, -R.T @ t, np.zeros(3)]).astype(np.float32)
    rng = np.random.default_rng(seed)
    # feature vector over all N features: true pairs share descriptors up to 3 % flipped bits, so hash the node from a majority of
    # stable bits (the first byte's high nibble) -- most pairs land in the same node, as with a real vocabulary
    node = (dd[:, 0] >> 4).astype(np.uint32) * 3 + 1
    ids = np.unique(node)
    start, feats = [0], []
    for nid in ids:
        feats.extend(np.nonzero(node == nid)[0].tolist())
        start.append(len(feats))
    fv = (ids, np.array(start, np.int32), np.array(feats, np.uint32))
    mp = (rng.random(len(kk)) < 0.15).astype(np.uint8)
    return sc, kk, dd, nL, rig, fv, mp


def test_oracle_rig_triangulation_selects_the_cross_camera_pairs(oracle):
    sc, kk, dd, nL, rig, fv, mp = _rig_inputs(oracle, 3)
    s2 = sc["level_sigma2"]
    n, m, bl = oracle.search_for_triangulation_rig(fv, kk, dd, mp, nL, fv, kk, dd, mp, nL, s2, s2, rig, False, False, False)
    got = np.nonzero(m >= 0)[0]
    assert n == len(got) > 100
    # every accepted pair crosses the cameras (same-camera pairs have no parallax), and most are the generating correspondences
    assert ((got < nL) != (m[got] < nL)).all()
    truth = {int(a): int(b) + nL for a, b in zip(sc["true_left"], sc["true_right"])}
    hit = sum(1 for i in got if i < nL and truth.get(int(i)) == int(m[i]))
    assert hit > 0.6 * sum(1 for i in got if i < nL)
    nc, mc, _ = oracle.search_for_triangulation_rig(fv, kk, dd, mp, nL, fv, kk, dd, mp, nL, s2, s2, rig, False, True, False)
    assert nc > n                                   # bCoarse skips the epipolar test: a feature also pairs with itself (distance 0)
    assert oracle.search_for_triangulation_rig(fv, kk, dd, mp, nL, fv, kk, dd, mp, nL, s2, s2, rig, True, False, False)[0] == 0
    assert bl.sum() < 0.1 * len(bl)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 4])
def test_gpu_search_for_triangulation_rig(gpu, oracle, seed):
    sc, kk, dd, nL, rig, fv, mp = _rig_inputs(orbx, seed)
    s2 = sc["level_sigma2"]
    for coarse, ori in ((False, True), (False, False), (True, True)):
        n, m = orbx.ORBmatcher(0.6, ori).SearchForTriangulationRig(fv, kk, dd, mp, nL, fv, kk, dd, mp, nL, s2, s2, rig, False, coarse)
        on, om, bl = oracle.search_for_triangulation_rig(fv, kk, dd, mp, nL, fv, kk, dd, mp, nL, s2, s2, rig.view(oracle.TRI_RIG_DTYPE),
                                                         False, coarse, ori)
        assert on > (15 if ori else 100)              # (the scene's keypoint angles are random: the rotation cull keeps ~3 of 30 bins)
        if coarse:                                    # no float gate: exact
            assert n == on and np.array_equal(m, om)
            continue
        diff = np.nonzero(m != om)[0]
        # tolerance parity: a decision may only differ where the oracle's gated quantity sits within rounding noise of its threshold
        assert all(bl[i] for i in diff), ("decision differs away from every gate", diff[:10], m[diff[:10]], om[diff[:10]])
        assert len(diff) <= 3
        if not ori:
            assert abs(n - on) <= len(diff)
    n, m = orbx.ORBmatcher(0.6, True).SearchForTriangulationRig(fv, kk, dd, mp, nL, fv, kk, dd, mp, nL, s2, s2, rig, True, False)
    assert n == 0 and (m == -1).all()
