"""Committed golden fixtures (tests/golden/*.npz, made by tools/gen_golden.py from the oracle in the build
container): the oracle must keep reproducing them (CPU), and the HIP path must reproduce them on the GPU box
without consulting the oracle at all."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXTRACT = sorted(glob.glob(os.path.join(HERE, "golden", "extract_*.npz")))
STEREO = sorted(glob.glob(os.path.join(HERE, "golden", "stereo_*.npz")))


def _kb(k):
    return np.ascontiguousarray(k).view(np.uint8).reshape(len(k), 28)


@pytest.mark.parametrize("path", EXTRACT, ids=os.path.basename)
def test_oracle_reproduces_extract_golden(oracle, path):
    g = np.load(path)
    ex = oracle.OracleExtractor(int(g["nfeatures"]), 1.2, int(g["nlevels"]), 20, 7)
    mono, k, d = ex.extract(g["image"], tuple(int(v) for v in g["lap"]))
    assert mono == int(g["mono"])
    assert np.array_equal(_kb(k), g["keypoints"]) and np.array_equal(d, g["descriptors"])
    assert [list(ex.level(l).shape) for l in range(int(g["nlevels"]))] == g["level_sizes"].tolist()


@pytest.mark.parametrize("path", STEREO, ids=os.path.basename)
def test_oracle_reproduces_stereo_golden(oracle, path):
    g = np.load(path)
    nf = int(g["nfeatures"])
    eL, eR = oracle.OracleExtractor(nf), oracle.OracleExtractor(nf)
    _, kL, dL = eL.extract(g["left"])
    _, kR, dR = eR.extract(g["right"])
    assert np.array_equal(_kb(kL), g["kL"]) and np.array_equal(_kb(kR), g["kR"])
    u, dep = oracle.stereo_match(eL, eR, kL, dL, kR, dR, float(g["bf"]), float(g["b"]))
    assert u.tobytes() == g["uRight"].tobytes() and dep.tobytes() == g["depth"].tobytes()
    idx, dist, ok = oracle.bf_knn2(dL, dR)
    assert np.array_equal(idx, g["knn_idx"]) and np.array_equal(dist, g["knn_dist"]) and np.array_equal(ok, g["knn_ok"])
    h, w = g["left"].shape
    n, m12, prev = oracle.search_init(kL, dL, kR, dR, (0, 0, w, h), np.stack([kL["x"], kL["y"]], 1), 100, 0.9, True)
    assert n == int(g["init_n"]) and np.array_equal(m12, g["init_m12"]) and prev.tobytes() == g["init_prev"].tobytes()


PROJ = sorted(glob.glob(os.path.join(HERE, "golden", "projection_*.npz")))


def _proj_inputs(g, mod):
    kc = np.ascontiguousarray(g["kc"]).view(mod.KP_DTYPE).reshape(-1)
    mps = np.ascontiguousarray(g["mps"]).view(mod.MP_DTYPE).reshape(-1)
    pts = np.ascontiguousarray(g["pts"]).view(mod.PP_DTYPE).reshape(-1)
    return kc, mps, pts, (0.0, 0.0, float(g["w"]), float(g["h"]))


@pytest.mark.parametrize("path", PROJ, ids=os.path.basename)
def test_oracle_reproduces_projection_golden(oracle, path):
    g = np.load(path)
    kc, mps, pts, bounds = _proj_inputs(g, oracle)
    n1, m1, o1 = oracle.search_by_projection(kc, g["dc"], g["uR"], bounds, g["scale"], mps, 3.0, True, 60.0, 0.8, g["occupied"])
    assert n1 == int(g["map_n"]) and np.array_equal(m1, g["map_match"]) and np.array_equal(o1, g["map_occ"])
    n2, m2, o2 = oracle.search_by_projection_frame(kc, g["dc"], g["uR"], bounds, pts, True, g["occupied"])
    assert n2 == int(g["frame_n"]) and np.array_equal(m2, g["frame_match"]) and np.array_equal(o2, g["frame_occ"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", PROJ, ids=os.path.basename)
def test_hip_reproduces_projection_golden(path):
    import orb_slam3_fast_amd as orbx
    g = np.load(path)
    kc, mps, pts, bounds = _proj_inputs(g, orbx)
    n1, m1, o1 = orbx.ORBmatcher(0.8, True).SearchByProjection(kc, g["dc"], g["uR"], bounds, g["scale"], mps, g["occupied"],
                                                              3.0, True, 60.0)
    assert n1 == int(g["map_n"]) and np.array_equal(m1, g["map_match"]) and np.array_equal(o1, g["map_occ"])
    n2, m2, o2 = orbx.ORBmatcher(0.8, True).SearchByProjectionFrame(kc, g["dc"], g["uR"], bounds, pts, g["occupied"])
    assert n2 == int(g["frame_n"]) and np.array_equal(m2, g["frame_match"]) and np.array_equal(o2, g["frame_occ"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", EXTRACT, ids=os.path.basename)
def test_hip_reproduces_extract_golden(path):
    import orb_slam3_fast_amd as orbx
    g = np.load(path)
    h, w = g["image"].shape
    ex = orbx.ORBextractor(int(g["nfeatures"]), 1.2, int(g["nlevels"]), 20, 7, max_width=w, max_height=h)
    mono, k, d = ex(g["image"], tuple(int(v) for v in g["lap"]))
    assert mono == int(g["mono"])
    assert np.array_equal(_kb(k), g["keypoints"]) and np.array_equal(d, g["descriptors"])
    assert [list(ex.image_pyramid(l).shape) for l in range(int(g["nlevels"]))] == g["level_sizes"].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("path", STEREO, ids=os.path.basename)
def test_hip_reproduces_stereo_golden(path):
    import orb_slam3_fast_amd as orbx
    g = np.load(path)
    nf = int(g["nfeatures"])
    h, w = g["left"].shape
    eL = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    eR = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
    _, kL, dL = eL(g["left"])
    _, kR, dR = eR(g["right"])
    assert np.array_equal(_kb(kL), g["kL"]) and np.array_equal(dL, g["dL"])
    assert np.array_equal(_kb(kR), g["kR"]) and np.array_equal(dR, g["dR"])
    u, dep = orbx.ComputeStereoMatches(eL, eR, float(g["bf"]), float(g["b"]))
    n = len(kL)
    assert u[0, :n].tobytes() == g["uRight"].tobytes() and dep[0, :n].tobytes() == g["depth"].tobytes()
    idx, dist, ok = orbx.bf_knn2(dL, dR)
    assert np.array_equal(idx, g["knn_idx"]) and np.array_equal(dist, g["knn_dist"]) and np.array_equal(ok, g["knn_ok"])
    nm, m12, prev = orbx.ORBmatcher(0.9, True).SearchForInitialization(kL, dL, kR, dR, (0.0, 0.0, float(w), float(h)),
                                                                     np.stack([kL["x"], kL["y"]], 1), 100)
    assert nm == int(g["init_n"]) and np.array_equal(m12, g["init_m12"])
    assert prev.reshape(-1).tobytes() == g["init_prev"].reshape(-1).tobytes()
