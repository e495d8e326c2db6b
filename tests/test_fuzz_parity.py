"""Randomised GPU-vs-oracle parity sweep (extraction + stereo association) over image sizes, extractor parameters
and image statistics.  Marked `slow`: run explicitly with  pytest tests/test_fuzz_parity.py -m "gpu and slow"
(ORBX_FUZZ_CASES sets the number of cases, default 60).  Every case is seeded, so a failure is reproducible."""
import os

import numpy as np
import pytest

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def _kb(k):
    return np.ascontiguousarray(k).view(np.uint8).reshape(len(k), 28)


def make_image(rng, w, h, stream):
    L, R = synth.stereo_pair(w, h, stream)
    mode = rng.integers(0, 6)
    out = []
    for im in (L, R):
        im = im.copy()
        if mode == 1:      # low contrast everywhere: many min-threshold cells
            im = (im // 6 + 100).astype(np.uint8)
        elif mode == 2:    # half white noise
            nz = rng.integers(0, 256, im.shape, dtype=np.uint8)
            im[:, : w // 2] = nz[:, : w // 2]
        elif mode == 3:    # large flat regions
            im[h // 4: 3 * h // 4, w // 5: w // 2] = 140
        elif mode == 4:    # strong gradient + texture
            g = np.linspace(0, 120, w, dtype=np.float32)[None, :]
            im = np.clip(im.astype(np.float32) * 0.5 + g, 0, 255).astype(np.uint8)
        out.append(im)
    return out


def test_fuzz_extract_and_stereo(oracle):
    assert orbx.device_count() > 0
    ncases = int(os.environ.get("ORBX_FUZZ_CASES", "60"))
    seed0 = int(os.environ.get("ORBX_FUZZ_SEED", "12345"))
    fails = []
    for case in range(ncases):
        rng = np.random.default_rng(seed0 + case)
        nl = int(rng.integers(2, 9))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5]))
        smallest = 32 + 35 + 6
        while int(np.ceil(smallest * sf ** (nl - 1))) + 8 > 600:
            nl -= 1
        minw = int(np.ceil(smallest * sf ** (nl - 1))) + 8
        w = int(rng.integers(max(minw, 200), 1000))
        h = int(rng.integers(max(minw, 200), 760))
        if max(w, h) / min(w, h) > 2.4:
            h = max(h, int(w / 2.4) + 1)
            if h - 32 < 35 * sf ** (nl - 1):
                continue
        nf = int(rng.integers(150, 3000))
        ini = int(rng.integers(10, 40))
        mn = int(rng.integers(3, ini + 1))
        lap = (0, 0) if rng.random() < 0.6 else (int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w + 50)))
        L, R = make_image(rng, w, h, 500 + case)
        try:
            exL = orbx.ORBextractor(nf, sf, nl, ini, mn, max_width=w, max_height=h)
            exR = orbx.ORBextractor(nf, sf, nl, ini, mn, max_width=w, max_height=h)
        except orbx.OrbxError as e:
            if e.code == orbx.E_UNSUPPORTED:
                continue
            raise
        oL, oR = oracle.OracleExtractor(nf, sf, nl, ini, mn), oracle.OracleExtractor(nf, sf, nl, ini, mn)
        mL, kL, dL = exL(L, lap)
        mR, kR, dR = exR(R, (0, 0))
        omL, okL, odL = oL.extract(L, lap)
        omR, okR, odR = oR.extract(R, (0, 0))
        ok = (mL == omL and mR == omR and np.array_equal(_kb(kL), _kb(okL)) and np.array_equal(dL, odL)
              and np.array_equal(_kb(kR), _kb(okR)) and np.array_equal(dR, odR))
        if ok and lap == (0, 0) and len(kL) and len(kR):
            bf, b = 0.12 * 500.0, 0.12
            u, dep = orbx.ComputeStereoMatches(exL, exR, bf, b)
            ou, od = oracle.stereo_match(oL, oR, okL, odL, okR, odR, bf, b)
            n = len(kL)
            ok = u[0, :n].tobytes() == ou.tobytes() and dep[0, :n].tobytes() == od.tobytes()
        if not ok:
            fails.append((seed0 + case, w, h, nf, sf, nl, ini, mn, lap))
        exL.close()
        exR.close()
    assert not fails, "mismatching cases (seed, w, h, nfeatures, scale, levels, ini, min, lap): %r" % fails


def test_fuzz_matchers(oracle):
    """SearchForInitialization, every SearchByProjection flavour, SearchForTriangulation, the Fuse / SearchBySim3 search, SearchByBoW on
    two key frames, GetFeaturesInArea and kNN on random frame pairs."""
    ncases = int(os.environ.get("ORBX_FUZZ_CASES", "60")) // 3 + 1
    seed0 = int(os.environ.get("ORBX_FUZZ_SEED", "12345")) + 5000
    fails = []
    for case in range(ncases):
        rng = np.random.default_rng(seed0 + case)
        w, h = int(rng.integers(320, 900)), int(rng.integers(300, 700))
        nf = int(rng.integers(300, 4000))
        try:
            ex = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
        except orbx.OrbxError as e:
            if e.code == orbx.E_UNSUPPORTED:   # portrait aspect < 0.5: the reference divides by zero there
                continue
            raise
        f0, f1 = synth.mono_frame(w, h, 900 + case, 0), synth.mono_frame(w, h, 900 + case, int(rng.integers(1, 4)))
        lap = (0, 1000) if rng.random() < 0.5 else (0, 0)
        try:
            _, k1, d1 = ex(f0, lap)
            _, k2, d2 = ex(f1, lap)
        except orbx.OrbxError as e:
            if e.code == orbx.E_UNSUPPORTED:   # portrait aspect < 0.5: the reference divides by zero there
                continue
            raise
        if len(k1) < 20 or len(k2) < 20:
            continue
        bounds = (0.0, 0.0, float(w), float(h))
        prev = np.stack([k1["x"], k1["y"]], 1) + rng.normal(0, 2.0, (len(k1), 2)).astype(np.float32)
        win = int(rng.choice([10, 30, 100]))
        ratio = float(rng.choice([0.6, 0.8, 0.9]))
        ori = bool(rng.integers(0, 2))
        n, m12, np_ = orbx.ORBmatcher(ratio, ori).SearchForInitialization(k1, d1, k2, d2, bounds, prev, win)
        on, om12, onp = oracle.search_init(k1, d1, k2, d2, bounds, prev, win, ratio, ori)
        ok = n == on and np.array_equal(m12, om12) and np_.tobytes() == onp.tobytes()
        # projection matchers with k1 as the "map points"
        nmp = len(k1)
        sf = ex.GetScaleFactors()
        mps = np.zeros(nmp, orbx.MP_DTYPE)
        mps["proj_x"] = k1["x"] + rng.normal(0, 4.0, nmp)
        mps["proj_y"] = k1["y"] + rng.normal(0, 4.0, nmp)
        mps["proj_xr"] = mps["proj_x"] - rng.uniform(0, 50, nmp).astype(np.float32)
        mps["view_cos"] = rng.uniform(0.99, 1.0, nmp).astype(np.float32)
        mps["track_depth"] = rng.uniform(1, 80, nmp).astype(np.float32)
        mps["predicted_level"] = np.clip(k1["octave"] + rng.integers(-1, 2, nmp), 0, 7)
        mps["in_view"] = rng.random(nmp) < 0.9
        mps["bad"] = rng.random(nmp) < 0.05
        mps["has_observations"] = rng.random(nmp) < 0.8
        mps["desc"] = d1 ^ np.packbits(rng.random((nmp, 32, 8)) < 0.05, axis=2).reshape(nmp, 32)
        uR = np.where(rng.random(len(k2)) < 0.6, k2["x"] - rng.uniform(0, 40, len(k2)), -1).astype(np.float32)
        occ = (rng.random(len(k2)) < 0.1).astype(np.uint8)
        th = float(rng.choice([1.0, 2.0, 5.0]))
        pts = np.zeros(nmp, orbx.PP_DTYPE)
        pts["u"], pts["v"], pts["ur"] = mps["proj_x"], mps["proj_y"], mps["proj_xr"]
        pts["radius"] = (np.float32(rng.choice([7.0, 15.0])) * sf[k1["octave"]]).astype(np.float32)
        pts["angle"] = k1["angle"]
        pts["min_level"], pts["max_level"] = k1["octave"] - 1, k1["octave"] + 1
        pts["valid"] = mps["in_view"]
        pts["has_observations"] = mps["has_observations"]
        pts["desc"] = mps["desc"]
        n2, m2, o2 = orbx.ORBmatcher(ratio, ori).SearchByProjectionFrame(k2, d2, uR, bounds, pts, occ)
        on2, om2, oo2 = oracle.search_by_projection_frame(k2, d2, uR, bounds, pts, ori, occ)
        ok = ok and n2 == on2 and np.array_equal(m2, om2) and np.array_equal(o2, oo2)
        n1, m1, o1 = orbx.ORBmatcher(ratio, ori).SearchByProjection(k2, d2, uR, bounds, sf, mps, occ, th, True, 40.0)
        on1, om1, oo1 = oracle.search_by_projection(k2, d2, uR, bounds, sf, mps, th, True, 40.0, ratio, occ)
        ok = ok and n1 == on1 and np.array_equal(m1, om1) and np.array_equal(o1, oo1)
        # stereo-fisheye flavours: frame = (k2 | k1) as left | right keypoints, random partner arrays
        nL, nR = len(k2), len(k1)
        kk, dd = np.concatenate([k2, k1]), np.concatenate([d2, d1])
        l2r = np.where(rng.random(nL) < 0.4, rng.integers(0, nR, nL), -1).astype(np.int32)
        r2l = np.where(rng.random(nR) < 0.4, rng.integers(0, nL, nR), -1).astype(np.int32)
        mpr = np.zeros(nmp, orbx.MPR_DTYPE)
        mpr["proj_yr"] = k1["y"] + rng.normal(0, 3.0, nmp)
        mpr["view_cos_r"] = rng.uniform(0.99, 1.0, nmp)
        mpr["predicted_level_r"] = np.where(rng.random(nmp) < 0.1, -1, np.clip(k1["octave"] + rng.integers(-1, 2, nmp), 0, 7))
        mpr["in_view_r"] = rng.random(nmp) < 0.8
        mpsf = mps.copy()
        mpsf["proj_xr"] = k1["x"] + rng.normal(0, 3.0, nmp)   # mTrackProjXR is a right-camera x here
        occf = (rng.random(nL + nR) < 0.1).astype(np.uint8)
        uvr = np.stack([mpsf["proj_xr"], mpr["proj_yr"]], 1).astype(np.float32)
        n3, m3, o3 = orbx.ORBmatcher(ratio, ori).SearchByProjectionFisheye(kk, dd, nL, bounds, sf, mpsf, mpr, l2r, r2l, occf, th, True, 40.0)
        on3, om3, oo3 = oracle.search_by_projection_fisheye(kk, dd, nL, bounds, sf, mpsf.view(oracle.MP_DTYPE), mpr.view(oracle.MPR_DTYPE),
                                                            th, True, 40.0, ratio, l2r, r2l, occf)
        ok = ok and n3 == on3 and np.array_equal(m3, om3) and np.array_equal(o3, oo3)
        n4, m4, o4 = orbx.ORBmatcher(ratio, ori).SearchByProjectionFrameFisheye(kk, dd, nL, bounds, pts, uvr, occf)
        on4, om4, oo4 = oracle.search_by_projection_frame_fisheye(kk, dd, nL, bounds, pts.view(oracle.PP_DTYPE), uvr, ori, occf)
        ok = ok and n4 == on4 and np.array_equal(m4, om4) and np.array_equal(o4, oo4)
        # round-3 matchers: relocalisation flavour, SearchForTriangulation (pinhole), Fuse / SearchBySim3 search, SearchByBoW(KF, KF)
        orb_dist = int(rng.choice([37, 64, 100, 255]))
        occk = (rng.random(len(k2)) < float(rng.choice([0.0, 0.3, 0.7]))).astype(np.uint8)
        n5, m5, o5 = orbx.ORBmatcher(ratio, ori).SearchByProjectionKeyFrame(k2, d2, bounds, pts, occk, orb_dist)
        on5, om5, oo5 = oracle.search_by_projection_keyframe(k2, d2, bounds, pts.view(oracle.PP_DTYPE), orb_dist, ori, occk)
        ok = ok and n5 == on5 and np.array_equal(m5, om5) and np.array_equal(o5, oo5)

        def fvec(desc, nodes):
            node = (desc[:, 0].astype(np.uint32) * 7 + (desc[:, 1] >> 6)) % nodes * 3 + 2
            keep = rng.random(len(desc)) >= 0.05
            ids = np.unique(node[keep])
            ids = ids[rng.random(len(ids)) >= 0.1] if len(ids) > 2 else ids
            start, feats = [0], []
            for nid in ids:
                feats.extend(np.nonzero(keep & (node == nid))[0].tolist())
                start.append(len(feats))
            return ids.astype(np.uint32), np.array(start, np.int32), np.array(feats, np.uint32)
        nodes = int(rng.choice([2, 16, 64, 200]))
        fv1, fv2 = fvec(d1, nodes), fvec(d2, nodes)
        hm1, hm2 = (rng.random(len(k1)) < 0.3).astype(np.uint8), (rng.random(len(k2)) < 0.3).astype(np.uint8)
        ur1 = None if rng.random() < 0.3 else np.where(rng.random(len(k1)) < 0.5, k1["x"] - 5, -1).astype(np.float32)
        ur2 = None if rng.random() < 0.3 else uR
        F12 = (np.array([[0, 0, 0.6], [0, 0, -0.8], [-0.6, 0.8, 0]]) + rng.normal(0, 3e-5, (3, 3)) * [[1, 1, 300], [1, 1, 300], [300, 300, 1]]).astype(np.float32)
        epi = np.array([rng.uniform(0, w), rng.uniform(0, h)], np.float32)
        sigma2 = (sf * sf).astype(np.float32)
        only_st, coarse = bool(rng.random() < 0.2), bool(rng.random() < 0.3)
        n6, _, m6 = orbx.ORBmatcher(ratio, ori).SearchForTriangulation(fv1, k1, d1, hm1, ur1, fv2, k2, d2, hm2, ur2, sf, sigma2, epi, F12,
                                                                       only_st, coarse)
        on6, om6 = oracle.search_for_triangulation(fv1, k1, d1, hm1, ur1, fv2, k2, d2, hm2, ur2, sf, sigma2, epi, F12, only_st, coarse, ori)
        ok = ok and n6 == on6 and np.array_equal(m6, om6)
        fp = np.zeros(nmp, orbx.FP_DTYPE)
        fp["u"], fp["v"], fp["ur"] = mps["proj_x"], mps["proj_y"], mps["proj_xr"]
        fp["predicted_level"] = mps["predicted_level"]
        fp["radius"] = (np.float32(rng.choice([3.0, 7.5])) * sf[mps["predicted_level"]]).astype(np.float32)
        fp["valid"], fp["desc"] = mps["in_view"], mps["desc"]
        gate = (1.0 / sigma2).astype(np.float32) if rng.random() < 0.6 else np.zeros(8, np.float32)
        md = int(rng.choice([50, 100]))
        n7, b7, d7 = orbx.ORBmatcher(ratio, ori).FuseSearch(k2, d2, ur2, bounds, gate, fp, md)
        on7, ob7, od7 = oracle.fuse_search(k2, d2, ur2, bounds, gate, fp.view(oracle.FP_DTYPE), md)
        ok = ok and n7 == on7 and np.array_equal(b7, ob7) and np.array_equal(d7, od7)
        n8, m8 = orbx.SearchByBoWKeyFrames(fv1, k1, d1, 1 - hm1, fv2, k2, d2, 1 - hm2, ratio, ori)
        on8, om8 = oracle.search_by_bow_keyframes(fv1, d1, k1["angle"], 1 - hm1, fv2, d2, k2["angle"], 1 - hm2, ratio, ori)
        ok = ok and n8 == on8 and np.array_equal(m8, om8)
        idx, dist, okk = orbx.bf_knn2(d1[: min(400, len(d1))], d2)
        oidx, odist, ookk = oracle.bf_knn2(d1[: min(400, len(d1))], d2)
        ok = ok and np.array_equal(idx, oidx) and np.array_equal(dist, odist) and np.array_equal(okk, ookk)
        if not ok:
            fails.append((seed0 + case, w, h, nf, win, ratio, ori))
        ex.close()
    assert not fails, "mismatching matcher cases: %r" % fails


def test_fuzz_preprocess(oracle):
    """cv::remap and CLAHE over random sizes, tile grids, clip limits and maps (smooth, seams, magnification, out of range)."""
    ncases = int(os.environ.get("ORBX_FUZZ_CASES", "60"))
    seed0 = int(os.environ.get("ORBX_FUZZ_SEED", "12345")) + 9000
    fails = []
    for case in range(ncases):
        rng = np.random.default_rng(seed0 + case)
        sw, sh = int(rng.integers(9, 700)), int(rng.integers(9, 500))
        img = make_image(rng, max(sw, 160), max(sh, 120), case)[0][:sh, :sw] if rng.random() < 0.6 else \
            rng.integers(0, 256, (sh, sw), dtype=np.uint8)
        img = np.ascontiguousarray(img)
        dw, dh = int(rng.integers(1, 700)), int(rng.integers(1, 500))
        u, v = np.meshgrid(np.arange(dw, dtype=np.float32), np.arange(dh, dtype=np.float32))
        kind = int(rng.integers(0, 5))
        if kind == 0:
            mx, my = synth.rectify_maps(dw, dh, sw, sh, seed=case, k1=float(rng.uniform(-0.4, 0.2)),
                                        rot_deg=tuple(rng.uniform(-2, 2, 3)))
        elif kind == 1:  # magnification / minification with a shear: windows of a thread's taps do not always fit
            sc = float(rng.choice([0.3, 0.7, 1.0, 1.6, 2.5, 5.0]))
            mx, my = (u * sc + v * 0.13 - 3).astype(np.float32), (v * sc - u * 0.21 + 2).astype(np.float32)
        elif kind == 2:
            mx = rng.uniform(-4, sw + 4, (dh, dw)).astype(np.float32)
            my = rng.uniform(-4, sh + 4, (dh, dw)).astype(np.float32)
        elif kind == 3:  # a seam: the right half of the map jumps elsewhere
            mx, my = (u + 0.37).astype(np.float32), (v - 0.62).astype(np.float32)
            mx[:, dw // 2:] = (sw - 1 - u[:, dw // 2:] * 0.5).astype(np.float32)
        else:  # exact 1/32 fractions and integer positions
            mx = (u + rng.integers(0, 33, (dh, dw)) / 32).astype(np.float32)
            my = (v + rng.integers(0, 33, (dh, dw)) / 32).astype(np.float32)
        if rng.random() < 0.3:
            mx[rng.integers(0, dh), rng.integers(0, dw)] = rng.choice([np.nan, np.inf, -np.inf, 1e10, -1e10])
        if not np.array_equal(orbx.remap(img, mx, my), oracle.remap(img, mx, my)):
            fails.append(("remap", case, sw, sh, dw, dh, kind))
        tiles = (int(rng.integers(1, 12)), int(rng.integers(1, 12)))
        clip = float(rng.choice([0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 40.0]))
        if sw > tiles[0] and sh > tiles[1]:
            if not np.array_equal(orbx.CLAHE(clip, tiles).apply(img), oracle.clahe(img, clip, tiles)):
                fails.append(("clahe", case, sw, sh, tiles, clip))
    assert not fails, fails[:10]


def test_fuzz_preproc_plans(oracle):
    """The pre-processing PLANS' round-6 kernels over random geometry: k_remap_lds (tile footprints through LDS; sources whose width is
    a multiple of 16, both forms via the hook, maps from mild rectifications to shears whose tiles do not fit and fall back), the
    single-channel input resize through the pyramid kernel, and the 16-pixel gray conversion with row tails -- against the oracle."""
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    import ctypes as C
    from orb_slam3_fast_amd import hipmem
    ncases = int(os.environ.get("ORBX_FUZZ_CASES", "60")) // 2 + 1
    seed0 = int(os.environ.get("ORBX_FUZZ_SEED", "12345")) + 13000
    fails = []

    def run_plan(pp, frames, w, h):
        n = len(frames)
        dev = DeviceBuffer.from_numpy(frames)
        rowb = frames.shape[2] * (frames.shape[3] if frames.ndim == 4 else 1)
        ptr, ow, oh, rp, ip = pp.run_device(dev.ptr.value, n, rowb, rowb * h)
        got = np.zeros((n, ip), np.uint8)
        hipmem._ck(hipmem.hip().hipMemcpy(got.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), got.nbytes, 2))
        return got[:, :oh * rp].reshape(n, oh, rp)[:, :, :ow]

    try:
        for case in range(ncases):
            rng = np.random.default_rng(seed0 + case)
            sw, sh = 16 * int(rng.integers(2, 50)), int(rng.integers(12, 400))
            dw, dh = int(rng.integers(5, 700)), int(rng.integers(3, 400))
            n = int(rng.integers(1, 20))
            nmaps = int(rng.integers(1, 3))
            frames = rng.integers(0, 256, (n, sh, sw), dtype=np.uint8)
            u, v = np.meshgrid(np.arange(dw, dtype=np.float32), np.arange(dh, dtype=np.float32))
            kind = int(rng.integers(0, 4))
            maps = []
            for m in range(nmaps):
                if kind == 0:
                    maps.append(synth.rectify_maps(dw, dh, sw, sh, seed=case + m, k1=float(rng.uniform(-0.3, 0.1)), rot_deg=tuple(rng.uniform(-1, 1, 3))))
                elif kind == 1:
                    sc = float(rng.choice([0.6, 0.9, 1.0, 1.3, 2.2]))
                    maps.append(((u * sc + v * 0.05 - 2 + m).astype(np.float32), (v * sc - u * 0.03 + 1).astype(np.float32)))
                elif kind == 2:   # a seam and a region outside the source
                    mx, my = (u * (sw / max(dw, 1)) + 0.4).astype(np.float32), (v * (sh / max(dh, 1)) - 0.6).astype(np.float32)
                    mx[:, dw // 2:] += 7.25
                    my[: max(dh // 5, 1)] = -9.0
                    maps.append((mx, my))
                else:
                    maps.append(((u + rng.integers(0, 33, (dh, dw)) / 32).astype(np.float32), (v + rng.integers(0, 33, (dh, dw)) / 32).astype(np.float32)))
            mapsx, mapsy = np.stack([a for a, _ in maps]), np.stack([b for _, b in maps])
            pp = orbx.Preproc(sw, sh, channels=1, maps=(mapsx, mapsy), max_batch=n)
            want = [oracle.remap(frames[i], mapsx[i % nmaps], mapsy[i % nmaps]) for i in range(n)]
            for hook in (1, 0):
                orbx.lib().orbx_debug_set_remap_lds(hook)
                got = run_plan(pp, frames, sw, sh)
                if not all(np.array_equal(got[i], want[i]) for i in range(n)):
                    fails.append(("remap plan", case, hook, sw, sh, dw, dh, n, nmaps, kind))
            # input resize of mono frames (the pyramid kernel on a two-level geometry)
            rw, rh = int(rng.integers(8, 2 * sw)), int(rng.integers(4, 2 * sh))
            if rw * 4 > sw and rh * 4 > sh:     # (scale factors below 4: beyond that the plan keeps the per-pixel kernel anyway)
                pr = orbx.Preproc(sw, sh, channels=1, out_size=(rw, rh), max_batch=n)
                got = run_plan(pr, frames, sw, sh)
                if not all(np.array_equal(got[i], oracle.resize(frames[i], rw, rh)) for i in range(n)):
                    fails.append(("resize plan", case, sw, sh, rw, rh, n))
            # gray: widths with and without a row tail, three / four channels
            cn, gw = int(rng.choice([3, 4])), int(rng.integers(1, 300))
            col = rng.integers(0, 256, (min(n, 3), sh, gw, cn), dtype=np.uint8)
            rgb = bool(rng.integers(0, 2))
            pg = orbx.Preproc(gw, sh, channels=cn, rgb=rgb, max_batch=len(col))
            got = run_plan(pg, col, gw, sh)
            if not all(np.array_equal(got[i], oracle.cvt_gray(col[i], rgb)) for i in range(len(col))):
                fails.append(("gray plan", case, gw, sh, cn, rgb))
    finally:
        orbx.lib().orbx_debug_set_remap_lds(1)
    assert not fails, fails[:10]


def test_fuzz_bow(oracle):
    """ComputeBoW + SearchByBoW over random tree shapes, scoring / weighting types, feature counts and eye splits."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_bow import _scene, _kps
    ncases = int(os.environ.get("ORBX_FUZZ_CASES", "60")) // 2 + 1
    seed0 = int(os.environ.get("ORBX_FUZZ_SEED", "12345")) + 13000
    fails = []
    for case in range(ncases):
        rng = np.random.default_rng(seed0 + case)
        k, L = int(rng.integers(2, 13)), int(rng.integers(1, 5))
        while k ** L > 20000:
            L -= 1
        lu = int(rng.integers(0, L + 2))
        scoring, weighting = int(rng.integers(0, 6)), int(rng.integers(0, 4))
        cols = synth.make_vocabulary(k, L, seed=case, early_leaf_prob=float(rng.choice([0.0, 0.05, 0.3])),
                                     stop_prob=float(rng.choice([0.0, 0.05, 0.5])))
        ovoc = oracle.Vocabulary(k, L, *cols, scoring=scoring, weighting=weighting)
        voc = orbx.ORBVocabulary(k, L, *cols, scoring=scoring, weighting=weighting)
        n_kf, n_f = int(rng.integers(1, 2500)), int(rng.integers(1, 2500))
        n_left = -1 if rng.random() < 0.5 else int(rng.integers(0, n_f + 1))
        kd, ka, kv, fd, fa = _scene(cols, n_kf, n_f, case, n_left)
        got_k, got_f = voc.transform(kd, lu), voc.transform(fd, lu)
        want_k, want_f = ovoc.transform(kd, lu), ovoc.transform(fd, lu)
        same = all(np.array_equal(x, y) for g, wnt in ((got_k, want_k), (got_f, want_f)) for x, y in zip(g[0] + g[1], wnt[0] + wnt[1]))
        same = same and np.array_equal(got_f[0][1].view(np.uint64), want_f[0][1].view(np.uint64))
        if not same:
            fails.append(("transform", case, k, L, lu, scoring, weighting, n_kf, n_f))
            continue
        ratio, ori = float(rng.choice([0.6, 0.7, 0.9])), bool(rng.integers(0, 2))
        n, m = orbx.SearchByBoW(want_k[1], _kps(ka), kd, kv, want_f[1], _kps(fa), fd, n_left, ratio, ori)
        on, om = oracle.search_by_bow(want_k[1], kd, ka, kv, want_f[1], fd, fa, n_left, ratio, ori)
        if n != on or not np.array_equal(m, om):
            fails.append(("search", case, k, L, lu, n_kf, n_f, n_left, ratio, ori))
    assert not fails, fails[:10]
