"""ORBextractor::operator() (src/ORBextractor.cc:1015-1106) and the tail of ComputeKeyPointsOctTree (:972-999) transcribed in
Python on top of the oracle's stage functions (pyramid levels, per-level FAST candidates, DistributeOctTree, IC_Angle,
GaussianBlur, computeOrbDescriptor): border offsets, octave / size fields, orientation on the unblurred level, blur before
describing, coordinate scaling for levels > 0, the mono / lapping slot rule and the returned monoIndex."""
import numpy as np
import pytest

from orb_slam3_fast_amd import synth

f32 = np.float32


def extract_py(oracle, img, nfeatures, nlevels, lap):
    ex = oracle.OracleExtractor(nfeatures, 1.2, nlevels, 20, 7)
    t = ex.tables()
    ex.compute_pyramid(img)
    per_level = []
    for level in range(nlevels):
        im = ex.level(level)
        h, w = im.shape
        minBX, minBY, maxBX, maxBY = 16, 16, w - 16, h - 16  # EDGE_THRESHOLD - 3
        cand = ex.detect_candidates(level)
        keys = ex.distribute(cand, minBX, maxBX, minBY, maxBY, int(t["nfeat"][level]))
        patch = int(f32(31) * t["scale"][level])  # const int scaledPatchSize = PATCH_SIZE * mvScaleFactor[level]
        keys["x"] += minBX
        keys["y"] += minBY
        keys["octave"] = level
        keys["size"] = patch
        for k in keys:
            k["angle"] = oracle.ic_angle(im, int(np.rint(k["x"])), int(np.rint(k["y"])))
        per_level.append(keys)
    n = sum(len(k) for k in per_level)
    out_k = np.zeros(n, oracle.KP_DTYPE)
    out_d = np.zeros((n, 32), np.uint8)
    mono, stereo = 0, n - 1
    for level, keys in enumerate(per_level):
        if len(keys) == 0:
            continue
        blurred = oracle.blur(ex.level(level))
        scale = t["scale"][level]
        for k in keys:
            d = oracle.descriptor(blurred, float(k["x"]), float(k["y"]), float(k["angle"]))
            k = k.copy()
            if level != 0:
                k["x"], k["y"] = f32(k["x"]) * scale, f32(k["y"]) * scale
            if lap[0] <= k["x"] <= lap[1]:
                out_k[stereo], out_d[stereo] = k, d
                stereo -= 1
            else:
                out_k[mono], out_d[mono] = k, d
                mono += 1
    return mono, out_k, out_d


@pytest.mark.parametrize("w,h,nf,nl,lap,stream", [(384, 288, 400, 8, (0, 0), 401), (320, 240, 300, 5, (100, 220), 402),
                                                   (400, 300, 500, 8, (0, 1000), 403)])
def test_python_assembly_of_operator_call_matches_oracle(oracle, w, h, nf, nl, lap, stream):
    img = synth.mono_frame(w, h, stream)
    img[: h // 4, : w // 3] = (img[: h // 4, : w // 3] // 8) + 90  # low-contrast zone: minThFAST cells
    mono, k, d = extract_py(oracle, img, nf, nl, lap)
    om, ok, od = oracle.OracleExtractor(nf, 1.2, nl, 20, 7).extract(img, lap)
    assert mono == om and len(k) == len(ok) > 100
    assert k.tobytes() == ok.tobytes() and np.array_equal(d, od)


def test_threaded_oracle_equals_serial(oracle):
    """oracle `cpu_mt` (2 eye threads x per-level tasks, the fork's structure) returns exactly the serial oracle's
    keypoints, descriptors, uRight and depth -- it is only a timing baseline."""
    import numpy as np
    from orb_slam3_fast_amd import synth
    L, R = synth.stereo_pair(400, 300, 77)
    eL, eR = oracle.OracleExtractor(600), oracle.OracleExtractor(600)
    _, kL, dL = eL.extract(L)
    _, kR, dR = eR.extract(R)
    bf, b = 0.12 * 532.03, 0.12
    u, d = oracle.stereo_match(eL, eR, kL, dL, kR, dR, bf, b)
    fL, fR = oracle.OracleExtractor(600), oracle.OracleExtractor(600)
    for _ in range(2):
        r = oracle.stereo_frame_mt(fL, fR, L, R, bf, b)
        assert kL.tobytes() == r[0].tobytes() and np.array_equal(dL, r[1])
        assert kR.tobytes() == r[2].tobytes() and np.array_equal(dR, r[3])
        assert u.tobytes() == r[4].tobytes() and d.tobytes() == r[5].tobytes()
        assert r[6] > 0 and r[7] > 0
