"""The C++ mirror classes (csrc/ORBextractor.h, ORBmatcher.h) compile with plain g++ against the C ABI, fail
loudly without a GPU, and on the GPU reproduce the oracle when driven like the reference's Frame."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "frame_like")
STUB = os.path.join(ROOT, "tests", "cpp", "opencv_stub")


import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
from build_cpp import build_exe  # noqa: E402  (shared with bench.py's latency leg)


@pytest.mark.parametrize("cv", [False, True])
def test_cpp_mirror_compiles_and_fails_loudly_without_gpu(cv):
    import orb_slam3_fast_amd as orbx
    exe = build_exe(cv)
    r = subprocess.run([exe], capture_output=True, text=True)
    if orbx.device_count() == 0:
        assert r.returncode == 3 and "no-device error" in r.stdout
    else:
        assert r.returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cv", [False, True])
def test_cpp_mirror_matches_oracle(oracle, tmp_path, cv):
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd import synth
    assert orbx.device_count() > 0
    exe = build_exe(cv)
    w, h, nf = 640, 480, 1000
    L, R = synth.stereo_pair(w, h, 61)
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    out = str(tmp_path / "o")
    r = subprocess.run([exe, str(w), str(h), str(nf), str(tmp_path / "L.raw"), str(tmp_path / "R.raw"), out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    monoL, monoR, nL, nR, nm = map(int, r.stdout.split())
    oL, oR = oracle.OracleExtractor(nf), oracle.OracleExtractor(nf)
    omL, okL, odL = oL.extract(L)
    omR, okR, odR = oR.extract(R)
    assert (monoL, monoR, nL, nR) == (omL, omR, len(okL), len(okR))
    assert open(out + ".kL", "rb").read() == okL.tobytes() and open(out + ".dL", "rb").read() == odL.tobytes()
    assert open(out + ".kR", "rb").read() == okR.tobytes() and open(out + ".dR", "rb").read() == odR.tobytes()
    ou, od = oracle.stereo_match(oL, oR, okL, odL, okR, odR, np.float32(0.12) * np.float32(532.03), 0.12)
    assert open(out + ".uR", "rb").read() == ou.tobytes() and open(out + ".depth", "rb").read() == od.tobytes()
    assert open(out + ".pyr3", "rb").read() == oL.level(3).tobytes()
    assert open(out + ".vpyr3", "rb").read() == oL.level(3).tobytes()        # the in-place view of the kept host pyramid
    assert open(out + ".vpyrR5", "rb").read() == oR.level(5).tobytes()       # ExtractStereo: right eye's view
    prev = np.stack([okL["x"], okL["y"]], 1)
    on, om12, _ = oracle.search_init(okL, odL, okR, odR, (0, 0, w, h), prev, 100, 0.9, True)
    assert nm == on and np.fromfile(out + ".m12", np.int32).tolist() == om12.tolist()
    # ORBextractor::ExtractStereo (one batched pipeline for both eyes + ComputeStereoMatches)
    assert open(out + ".pkL", "rb").read() == okL.tobytes() and open(out + ".pdR", "rb").read() == odR.tobytes()
    assert open(out + ".puR", "rb").read() == ou.tobytes() and open(out + ".pdepth", "rb").read() == od.tobytes()


@pytest.mark.gpu
def test_cpp_mirror_fisheye_matches_python_binding(tmp_path):
    """ORB_SLAM3::ComputeStereoFishEyeMatches (csrc/ORBmatcher.h) is the same C-ABI call as the ctypes mirror."""
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd import synth
    exe = build_exe()
    sc = synth.fisheye_stereo_scene(4)
    rig = orbx.kb8_rig(sc["cam1"], sc["cam2"], sc["R12"], sc["t12"])
    names = {}
    for key, arr in (("kL", sc["kL"]), ("dL", sc["dL"]), ("kR", sc["kR"]), ("dR", sc["dR"]), ("rig", rig),
                     ("s2", sc["level_sigma2"])):
        names[key] = str(tmp_path / (key + ".raw"))
        np.ascontiguousarray(arr).tofile(names[key])
    out = str(tmp_path / "f")
    r = subprocess.run([exe, "fisheye", names["kL"], names["dL"], str(sc["mono_left"]), names["kR"], names["dR"],
                        str(sc["mono_right"]), names["rig"], names["s2"], out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    n, nd, l2r, r2l, dep, pts = orbx.ComputeStereoFishEyeMatches(sc["kL"], sc["dL"], sc["mono_left"], sc["kR"], sc["dR"],
                                                                 sc["mono_right"], rig, sc["level_sigma2"])
    assert int(r.stdout.split()[0]) == n > 100
    assert np.array_equal(np.fromfile(out + ".l2r", np.int32), l2r) and np.array_equal(np.fromfile(out + ".r2l", np.int32), r2l)
    assert np.fromfile(out + ".depth", np.float32).tobytes() == dep.tobytes()
    assert np.fromfile(out + ".p3d", np.float32).tobytes() == pts.tobytes()
    assert (np.fromfile(out + ".uR", np.float32) == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("cv", [False, True])
def test_cpp_mirror_rectify_clahe_matches_oracle(oracle, tmp_path, cv):
    """ORB_SLAM3::remap / CLAHE / StereoRectifier (csrc/Preprocess.h) against the oracle's cv::remap / CLAHE restatement;
    cv=True also drives the cv::remap(InputArray, OutputArray, InputArray, InputArray, int) overload (src/System.cc:294)."""
    from orb_slam3_fast_amd import synth
    exe = build_exe(cv)
    sw, sh, dw, dh = 512, 512, 480, 470
    L, R = synth.stereo_pair(sw, sh, 71)
    ml = synth.rectify_maps(dw, dh, sw, sh, seed=3)
    mr = synth.rectify_maps(dw, dh, sw, sh, seed=4, rot_deg=(-0.2, 0.3, 0.1))
    L.tofile(tmp_path / "L.raw")
    R.tofile(tmp_path / "R.raw")
    np.stack([ml[0], ml[1], mr[0], mr[1]]).astype(np.float32).tofile(tmp_path / "maps.raw")
    out = str(tmp_path / "r")
    r = subprocess.run([exe, "rectify", str(sw), str(sh), str(dw), str(dh), str(tmp_path / "L.raw"), str(tmp_path / "R.raw"),
                        str(tmp_path / "maps.raw"), out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    eqL, eqR = oracle.clahe(L, 3.0, (8, 8)), oracle.clahe(R, 3.0, (8, 8))
    wl, wr = oracle.remap(eqL, *ml), oracle.remap(eqR, *mr)
    assert np.array_equal(np.fromfile(out + ".eqL", np.uint8).reshape(sh, sw), eqL)
    for name, want in (("a", wl), ("b", wr), ("c", wl), ("d", wr)):
        assert np.array_equal(np.fromfile(out + "." + name, np.uint8).reshape(dh, dw), want), name


@pytest.mark.gpu
def test_cpp_mirror_bow_matches_oracle(oracle, tmp_path):
    """ORB_SLAM3::ORBVocabulary (loadFromTextFile, transform) and SearchByBoW (csrc/ORBVocabulary.h) against the oracle."""
    from orb_slam3_fast_amd import synth
    from test_bow import _scene
    exe = build_exe()
    cols = synth.make_vocabulary(8, 3, seed=21)
    ovoc = oracle.Vocabulary(8, 3, *cols)
    path = str(tmp_path / "voc.txt")
    ovoc.save(path)
    kd, ka, kv, fd, fa = _scene(cols, 600, 550, 21)
    for name, arr in (("kd", kd), ("ka", ka), ("kv", kv), ("fd", fd), ("fa", fa)):
        np.ascontiguousarray(arr).tofile(tmp_path / (name + ".raw"))
    out = str(tmp_path / "b")
    r = subprocess.run([exe, "bow", path] + [str(tmp_path / (n + ".raw")) for n in ("kd", "ka", "kv", "fd", "fa")] + ["1", "-1", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    (words, values), f_fv = ovoc.transform(fd, 1)
    kf_fv = ovoc.transform(kd, 1)[1]
    n, m = oracle.search_by_bow(kf_fv, kd, ka, kv, f_fv, fd, fa, -1, 0.7, True)
    assert np.array_equal(np.fromfile(out + ".words", np.uint32), words)
    assert np.array_equal(np.fromfile(out + ".values", np.float64).view(np.uint64), values.view(np.uint64))
    assert np.array_equal(np.fromfile(out + ".match", np.int32), m)
    assert ("%d matches" % n) in r.stdout and n > 50


@pytest.mark.gpu
def test_cpp_mirror_relocalisation_and_triangulation_match_oracle(oracle, tmp_path):
    """ORBmatcher::SearchByProjection(CurrentFrame, keyFramePoints, ORBdist, ...) (csrc/ORBmatcher.h) and SearchForTriangulation
    (csrc/ORBVocabulary.h, DBoW2::FeatureVector in, vMatchedPairs out) against the oracle."""
    import orb_slam3_fast_amd as orbx
    import test_reloc_triangulation as T
    exe = build_exe()
    f = T._frames(oracle, 752, 480, 1500, 72)
    rng = np.random.default_rng(55)
    pts = T._kf_points(orbx, rng, f, 10.0)
    occ = (rng.random(len(f["k2"])) < 0.3).astype(np.uint8)
    fv1, mp1, _, fv2, mp2, _, ep, F = T._tri_inputs(f, rng, True)
    pre = str(tmp_path / "rt")
    for ext, arr in (("k1", f["k1"]), ("d1", f["d1"]), ("k2", f["k2"]), ("d2", f["d2"]), ("pts", pts), ("occ", occ), ("n1", fv1[0]),
                     ("s1", fv1[1]), ("f1", fv1[2]), ("n2", fv2[0]), ("s2", fv2[1]), ("f2", fv2[2]), ("mp1", mp1), ("mp2", mp2),
                     ("sf", f["sf"]), ("sg", f["sigma2"]), ("epF", np.concatenate([ep, F.reshape(9)]).astype(np.float32))):
        np.ascontiguousarray(arr).tofile(pre + "." + ext)
    fp = T._fuse_points(orbx, rng, f, 3.0, 1.0)
    inv = (1.0 / f["sigma2"]).astype(np.float32)
    p12, p21 = T._sim3_inputs(orbx, rng, f)
    good1, good2 = (1 - mp1).astype(np.uint8), (1 - mp2).astype(np.uint8)
    for ext, arr in (("fp", fp), ("isg", inv), ("p12", p12), ("p21", p21), ("good1", good1), ("good2", good2)):
        np.ascontiguousarray(arr).tofile(pre + "." + ext)
    r = subprocess.run([exe, "reloc_tri", pre, "752", "480", "100"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    nr, nt, npairs, nfz, nsim, nbow, nbow_rig = map(int, r.stdout.split())
    onr, omatch, oocc = oracle.search_by_projection_keyframe(f["k2"], f["d2"], f["bounds"], pts, 100, True, occ)
    ont, om12 = oracle.search_for_triangulation(fv1, f["k1"], f["d1"], mp1, None, fv2, f["k2"], f["d2"], mp2, None, f["sf"], f["sigma2"],
                                                ep, F, False, False, True)
    assert nr == onr > 100 and np.array_equal(np.fromfile(pre + ".rmatch", np.int32), omatch)
    assert np.array_equal(np.fromfile(pre + ".rocc", np.uint8), oocc)
    assert nt == ont == npairs and ont > 50 and np.array_equal(np.fromfile(pre + ".m12", np.int32), om12)
    # ORBmatcher::Fuse (search), SearchBySim3, SearchByBoW(KeyFrame*, KeyFrame*) through the mirror
    onf, obi, _ = oracle.fuse_search(f["k2"], f["d2"], None, f["bounds"], inv, fp)
    assert nfz == onf > 50 and np.array_equal(np.fromfile(pre + ".fuse", np.int32), obi)
    z = np.zeros(8, np.float32)
    _, m1, _ = oracle.fuse_search(f["k2"], f["d2"], None, f["bounds"], z, p12, 100)
    _, m2, _ = oracle.fuse_search(f["k1"], f["d1"], None, f["bounds"], z, p21, 100)
    agree = m1 >= 0
    agree[agree] = m2[m1[agree]] == np.arange(len(m1))[agree]
    assert nsim == agree.sum() > 100 and np.array_equal(np.fromfile(pre + ".sim3", np.int32), np.where(agree, m1, -1))
    onb, omb = oracle.search_by_bow_keyframes(fv1, f["d1"], f["k1"]["angle"], good1, fv2, f["d2"], f["k2"]["angle"], good2, 0.75, True)
    assert nbow == onb > 20 and np.array_equal(np.fromfile(pre + ".bowkf", np.int32), omb)
    # rig key frames: features past mvKeysUn.size() (here the last quarter) are skipped like src/ORBmatcher.cc:799,816
    g1r, g2r = good1.copy(), good2.copy()
    g1r[len(g1r) * 3 // 4:] = 0
    g2r[len(g2r) * 3 // 4:] = 0
    onr_, ombr = oracle.search_by_bow_keyframes(fv1, f["d1"], f["k1"]["angle"], g1r, fv2, f["d2"], f["k2"]["angle"], g2r, 0.75, True)
    got = np.fromfile(pre + ".bowkf_rig", np.int32)
    assert nbow_rig == onr_ > 10 and len(got) == len(good1) and np.array_equal(got, ombr)
