"""Second, literal restatements (Python, numpy float32 where the reference computes in float) of the guided matchers and
of the frame grid, transcribed statement by statement from the reference and checked exactly against the C++ oracle:
  Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea   src/Frame.cc:520-547, 765-844
  ORBmatcher::SearchForInitialization + ComputeThreeMaxima       src/ORBmatcher.cc:618-764, 1920-1955
  ORBmatcher::SearchByProjection (local map / last frame), Nleft == -1 and Nleft != -1   :41-221, :1594-1806"""
import math

import numpy as np
import pytest

from orb_slam3_fast_amd import synth

f32 = np.float32
COLS, ROWS, HISTO, TH_HIGH, TH_LOW = 64, 48, 30, 100, 50


def c_round(v):
    v = float(v)
    return int(math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5))


def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


class Grid:
    def __init__(self, kps, bounds):
        self.kps = kps
        self.minX, self.minY, maxX, maxY = (f32(v) for v in bounds)
        self.invW = f32(COLS) / f32(maxX - self.minX)
        self.invH = f32(ROWS) / f32(maxY - self.minY)
        self.cells = [[[] for _ in range(ROWS)] for _ in range(COLS)]
        for i in range(len(kps)):
            px = c_round((f32(kps["x"][i]) - self.minX) * self.invW)
            py = c_round((f32(kps["y"][i]) - self.minY) * self.invH)
            if px < 0 or px >= COLS or py < 0 or py >= ROWS:
                continue
            self.cells[px][py].append(i)

    def area(self, x, y, r, minLevel, maxLevel):
        x, y, r = f32(x), f32(y), f32(r)
        out = []
        c0 = max(0, int(math.floor(float((x - self.minX - r) * self.invW))))
        if c0 >= COLS:
            return out
        c1 = min(COLS - 1, int(math.ceil(float((x - self.minX + r) * self.invW))))
        if c1 < 0:
            return out
        r0 = max(0, int(math.floor(float((y - self.minY - r) * self.invH))))
        if r0 >= ROWS:
            return out
        r1 = min(ROWS - 1, int(math.ceil(float((y - self.minY + r) * self.invH))))
        if r1 < 0:
            return out
        check = minLevel > 0 or maxLevel >= 0
        for ix in range(c0, c1 + 1):
            for iy in range(r0, r1 + 1):
                for j in self.cells[ix][iy]:
                    kp = self.kps[j]
                    if check:
                        if kp["octave"] < minLevel:
                            continue
                        if maxLevel >= 0 and kp["octave"] > maxLevel:
                            continue
                    if abs(f32(kp["x"]) - x) < r and abs(f32(kp["y"]) - y) < r:
                        out.append(j)
        return out


def three_maxima(hist):
    max1 = max2 = max3 = 0
    ind1 = ind2 = ind3 = -1
    for i in range(HISTO):
        s = len(hist[i])
        if s > max1:
            max3, max2, max1, ind3, ind2, ind1 = max2, max1, s, ind2, ind1, i
        elif s > max2:
            max3, max2, ind3, ind2 = max2, s, ind2, i
        elif s > max3:
            max3, ind3 = s, i
    if f32(max2) < f32(0.1) * f32(max1):
        ind2 = ind3 = -1
    elif f32(max3) < f32(0.1) * f32(max1):
        ind3 = -1
    return ind1, ind2, ind3


def rot_bin(a1, a2):
    rot = f32(a1) - f32(a2)
    if rot < 0.0:
        rot = rot + f32(360.0)
    b = c_round(rot * (f32(1.0) / f32(HISTO)))  # const float factor = 1.0f / HISTO_LENGTH
    return 0 if b == HISTO else b


def search_for_initialization_py(k1, d1, k2, d2, bounds, prev, window, nnratio, check_ori):
    n1, n2 = len(k1), len(k2)
    m12, m21, mdist = [-1] * n1, [-1] * n2, [2 ** 31 - 1] * n2
    hist = [[] for _ in range(HISTO)]
    g2 = Grid(k2, bounds)
    prev = np.array(prev, f32).copy()
    nmatches = 0
    for i1 in range(n1):
        if k1["octave"][i1] > 0:
            continue
        cand = g2.area(prev[i1, 0], prev[i1, 1], window, 0, 0)
        if not cand:
            continue
        best, best2, bidx = 2 ** 31 - 1, 2 ** 31 - 1, -1
        for i2 in cand:
            dist = hamming(d1[i1], d2[i2])
            if mdist[i2] <= dist:
                continue
            if dist < best:
                best2, best, bidx = best, dist, i2
            elif dist < best2:
                best2 = dist
        if best <= TH_LOW and f32(best) < f32(best2) * f32(nnratio):
            if m21[bidx] >= 0:
                m12[m21[bidx]] = -1
                nmatches -= 1
            m12[i1], m21[bidx], mdist[bidx] = bidx, i1, best
            nmatches += 1
            if check_ori:
                hist[rot_bin(k1["angle"][i1], k2["angle"][bidx])].append(i1)
    if check_ori:
        keep = three_maxima(hist)
        for i in range(HISTO):
            if i in keep:
                continue
            for idx1 in hist[i]:
                if m12[idx1] >= 0:
                    m12[idx1] = -1
                    nmatches -= 1
    for i1 in range(n1):
        if m12[i1] >= 0:
            prev[i1] = (k2["x"][m12[i1]], k2["y"][m12[i1]])
    return nmatches, np.array(m12, np.int32), prev


def search_by_projection_py(kps, desc, uR, n_left, bounds, sf, mps, mpr, th, far, th_far, nnratio, l2r, r2l, occupied):
    """n_left < 0: pinhole (one grid over kps, mvuRight gate); n_left >= 0: stereo-fisheye (two grids, partner slots)."""
    fe = n_left >= 0
    nL = n_left if fe else len(kps)
    gL = Grid(kps[:nL], bounds)
    gR = Grid(kps[nL:], bounds) if fe else None
    occ = np.array(occupied, np.uint8).copy()
    match = np.full(len(kps), -1, np.int32)
    nmatches = 0
    bfactor = th != 1.0

    def assign(slot, i):
        match[slot] = i
        occ[slot] = mps["has_observations"][i]

    def best2(cand, base, dmp, gate):
        b, lv, b2, lv2, bi = 256, -1, 256, -1, -1
        for idx in cand:
            if occ[base + idx]:
                continue
            if gate is not None and gate(idx):
                continue
            dist = hamming(dmp, desc[base + idx])
            if dist < b:
                b2, b, lv2, lv, bi = b, dist, lv, int(kps["octave"][base + idx]), idx
            elif dist < b2:
                lv2, b2 = int(kps["octave"][base + idx]), dist
        return b, lv, b2, lv2, bi

    for i in range(len(mps)):
        mp = mps[i]
        in_r = bool(mpr["in_view_r"][i]) if fe else False
        if not mp["in_view"] and not in_r:
            continue
        if far and f32(mp["track_depth"]) > f32(th_far):
            continue
        if mp["bad"]:
            continue
        skip_right = False
        if mp["in_view"]:
            level = int(mp["predicted_level"])
            r = f32(2.5) if float(mp["view_cos"]) > 0.998 else f32(4.0)
            if bfactor:
                r = r * f32(th)
            rad = r * sf[level]
            cand = gL.area(mp["proj_x"], mp["proj_y"], rad, level - 1, level)
            if cand:
                gate = None
                if not fe and uR is not None:
                    gate = lambda idx: uR[idx] > 0 and abs(f32(mp["proj_xr"]) - f32(uR[idx])) > rad  # noqa: E731
                b, lv, b2, lv2, bi = best2(cand, 0, mp["desc"], gate)
                if b <= TH_HIGH:
                    if lv == lv2 and f32(b) > f32(nnratio) * f32(b2):
                        skip_right = True
                    elif lv != lv2 or f32(b) <= f32(nnratio) * f32(b2):
                        assign(bi, i)
                        if fe and l2r[bi] != -1:
                            assign(l2r[bi] + nL, i)
                            nmatches += 1
                        nmatches += 1
        if fe and in_r and not skip_right:
            level = int(mpr["predicted_level_r"][i])
            if level != -1:
                r = f32(2.5) if float(mpr["view_cos_r"][i]) > 0.998 else f32(4.0)
                cand = gR.area(mp["proj_xr"], mpr["proj_yr"][i], r * sf[level], level - 1, level)
                if cand:
                    b, lv, b2, lv2, bi = best2(cand, nL, mp["desc"], None)
                    if b <= TH_HIGH and not (lv == lv2 and f32(b) > f32(nnratio) * f32(b2)):
                        if r2l[bi] != -1:
                            assign(r2l[bi], i)
                            nmatches += 1
                        assign(bi + nL, i)
                        nmatches += 1
    return nmatches, match, occ


def search_by_projection_frame_py(kps, desc, uR, n_left, bounds, pts, uvr, check_ori, occupied):
    fe = n_left >= 0
    nL = n_left if fe else len(kps)
    gL = Grid(kps[:nL], bounds)
    gR = Grid(kps[nL:], bounds) if fe else None
    occ = np.array(occupied, np.uint8).copy()
    match = np.full(len(kps), -1, np.int32)
    hist = [[] for _ in range(HISTO)]
    nmatches = 0
    for i in range(len(pts)):
        p = pts[i]
        if not p["valid"]:
            continue
        for side in ((0, 1) if fe else (0,)):
            base = nL if side else 0
            grid = gR if side else gL
            x, y = (uvr[i][0], uvr[i][1]) if side else (p["u"], p["v"])
            cand = grid.area(x, y, p["radius"], int(p["min_level"]), int(p["max_level"]))
            if not cand:
                if side == 0:
                    break  # `if (vIndices2.empty()) continue;` skips the right camera as well
                continue
            b, bi = 256, -1
            for i2 in cand:
                if occ[base + i2]:
                    continue
                if not fe and uR is not None and uR[i2] > 0 and abs(f32(p["ur"]) - f32(uR[i2])) > f32(p["radius"]):
                    continue
                dist = hamming(p["desc"], desc[base + i2])
                if dist < b:
                    b, bi = dist, i2
            if b <= TH_HIGH:
                match[base + bi] = i
                occ[base + bi] = p["has_observations"]
                nmatches += 1
                if check_ori:
                    hist[rot_bin(p["angle"], kps["angle"][base + bi])].append(base + bi)
    if check_ori:
        keep = three_maxima(hist)
        for b in range(HISTO):
            if b in keep:
                continue
            for idx in hist[b]:
                match[idx] = -1
                nmatches -= 1
    return nmatches, match, occ


@pytest.fixture(scope="module")
def frames(oracle):
    w, h = 480, 360
    f0, f1 = synth.mono_frame(w, h, 310, 0), synth.mono_frame(w, h, 310, 2)
    ex = oracle.OracleExtractor(500)
    _, k1, d1 = ex.extract(f0)
    _, k2, d2 = ex.extract(f1)
    return dict(w=w, h=h, k1=k1, d1=d1, k2=k2, d2=d2, sf=ex.tables()["scale"], bounds=(0.0, 0.0, float(w), float(h)))


def _views(mod, rng, k1, d1, k2, sf):
    n = len(k1)
    mps = np.zeros(n, mod.MP_DTYPE)
    mps["proj_x"], mps["proj_y"] = k1["x"] + rng.normal(0, 3.0, n), k1["y"] + rng.normal(0, 3.0, n)
    mps["proj_xr"] = mps["proj_x"] - rng.uniform(0, 40, n).astype(np.float32)
    mps["view_cos"], mps["track_depth"] = rng.choice([0.9, 0.9985], n), rng.uniform(1, 80, n)
    mps["predicted_level"] = np.clip(k1["octave"] + rng.integers(-1, 2, n), 0, 7)
    mps["in_view"], mps["bad"], mps["has_observations"] = rng.random(n) < 0.85, rng.random(n) < 0.05, rng.random(n) < 0.75
    mps["desc"] = d1 ^ np.packbits(rng.random((n, 32, 8)) < 0.05, axis=2).reshape(n, 32)
    pts = np.zeros(n, mod.PP_DTYPE)
    pts["u"], pts["v"], pts["ur"] = mps["proj_x"], mps["proj_y"], mps["proj_xr"]
    pts["radius"], pts["angle"] = np.float32(7.0) * sf[k1["octave"]], k1["angle"]
    pts["min_level"], pts["max_level"] = k1["octave"] - 1, k1["octave"] + 1
    pts["valid"], pts["has_observations"], pts["desc"] = mps["in_view"], mps["has_observations"], mps["desc"]
    return mps, pts


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (30, 0.8, False), (10, 0.9, True)])
def test_python_search_for_initialization_matches_oracle(oracle, frames, window, ratio, ori):
    f = frames
    prev = np.stack([f["k1"]["x"], f["k1"]["y"]], 1).astype(np.float32)
    en, em, ep = search_for_initialization_py(f["k1"], f["d1"], f["k2"], f["d2"], f["bounds"], prev, window, ratio, ori)
    on, om, op = oracle.search_init(f["k1"], f["d1"], f["k2"], f["d2"], f["bounds"], prev, window, ratio, ori)
    assert en == on and np.array_equal(em, om) and ep.tobytes() == op.tobytes() and (window < 100 or on > 30)


@pytest.mark.parametrize("seed,th,far", [(1, 3.0, True), (2, 1.0, False)])
def test_python_search_by_projection_matches_oracle(oracle, frames, seed, th, far):
    f = frames
    rng = np.random.default_rng(seed)
    mps, pts = _views(oracle, rng, f["k1"], f["d1"], f["k2"], f["sf"])
    uR = np.where(rng.random(len(f["k2"])) < 0.6, f["k2"]["x"] - rng.uniform(0, 30, len(f["k2"])), -1).astype(np.float32)
    occ = (rng.random(len(f["k2"])) < 0.08).astype(np.uint8)
    e = search_by_projection_py(f["k2"], f["d2"], uR, -1, f["bounds"], f["sf"], mps, None, th, far, 40.0, 0.8, None, None, occ)
    o = oracle.search_by_projection(f["k2"], f["d2"], uR, f["bounds"], f["sf"], mps, th, far, 40.0, 0.8, occ)
    assert e[0] == o[0] and np.array_equal(e[1], o[1]) and np.array_equal(e[2], o[2]) and o[0] > 20
    for ori in (True, False):
        e = search_by_projection_frame_py(f["k2"], f["d2"], uR, -1, f["bounds"], pts, None, ori, occ)
        o = oracle.search_by_projection_frame(f["k2"], f["d2"], uR, f["bounds"], pts, ori, occ)
        assert e[0] == o[0] and np.array_equal(e[1], o[1]) and np.array_equal(e[2], o[2]) and o[0] > 20


@pytest.mark.parametrize("seed", [3, 4])
def test_python_fisheye_search_by_projection_matches_oracle(oracle, frames, seed):
    f = frames
    rng = np.random.default_rng(seed)
    kk, dd = np.concatenate([f["k2"], f["k1"]]), np.concatenate([f["d2"], f["d1"]])
    nL, nR, n = len(f["k2"]), len(f["k1"]), len(f["k1"])
    mps, pts = _views(oracle, rng, f["k1"], f["d1"], f["k2"], f["sf"])
    mps["proj_xr"] = f["k1"]["x"] + rng.normal(0, 3.0, n)
    mpr = np.zeros(n, oracle.MPR_DTYPE)
    mpr["proj_yr"], mpr["view_cos_r"] = f["k1"]["y"] + rng.normal(0, 3.0, n), rng.choice([0.9, 0.9985], n)
    mpr["predicted_level_r"] = np.where(rng.random(n) < 0.1, -1, np.clip(f["k1"]["octave"] + rng.integers(-1, 2, n), 0, 7))
    mpr["in_view_r"] = rng.random(n) < 0.8
    l2r = np.where(rng.random(nL) < 0.4, rng.integers(0, nR, nL), -1).astype(np.int32)
    r2l = np.where(rng.random(nR) < 0.4, rng.integers(0, nL, nR), -1).astype(np.int32)
    occ = (rng.random(nL + nR) < 0.08).astype(np.uint8)
    e = search_by_projection_py(kk, dd, None, nL, f["bounds"], f["sf"], mps, mpr, 3.0, True, 40.0, 0.8, l2r, r2l, occ)
    o = oracle.search_by_projection_fisheye(kk, dd, nL, f["bounds"], f["sf"], mps, mpr, 3.0, True, 40.0, 0.8, l2r, r2l, occ)
    assert e[0] == o[0] and np.array_equal(e[1], o[1]) and np.array_equal(e[2], o[2]) and o[0] > 80
    uvr = np.stack([mps["proj_xr"], mpr["proj_yr"]], 1).astype(np.float32)
    e = search_by_projection_frame_py(kk, dd, None, nL, f["bounds"], pts, uvr, True, occ)
    o = oracle.search_by_projection_frame_fisheye(kk, dd, nL, f["bounds"], pts, uvr, True, occ)
    assert e[0] == o[0] and np.array_equal(e[1], o[1]) and np.array_equal(e[2], o[2]) and o[0] > 80


def test_python_grid_matches_oracle(oracle, frames):
    f = frames
    g = Grid(f["k2"], f["bounds"])
    rng = np.random.default_rng(5)
    for _ in range(60):
        x, y, r = rng.uniform(-20, f["w"] + 20), rng.uniform(-20, f["h"] + 20), rng.choice([5.0, 15.0, 60.0])
        lo = int(rng.integers(-1, 4))
        hi = int(rng.integers(-1, 7))
        assert g.area(x, y, r, lo, hi) == oracle.features_in_area(f["k2"], f["bounds"], x, y, r, lo, hi).tolist()
