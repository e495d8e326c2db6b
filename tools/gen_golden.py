#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no golden vectors for this path, so the fixtures are produced here, in the
build container, by oracle/liborb_oracle.so on seeded synthetic inputs; they pin (a) the oracle against
accidental change and (b) the HIP path on the GPU box independently of the oracle build.
Each .npz holds the INPUT image(s) and the expected outputs (keypoints as raw 28-byte records, descriptors,
stereo / kNN / initialisation-match results).   usage: python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402
from orb_slam3_fast_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def extract_case(name, w, h, nf, nl, stream, lap):
    img = synth.mono_frame(w, h, stream)
    img[: h // 4, : w // 3] = (img[: h // 4, : w // 3] // 8) + 90
    ex = O.OracleExtractor(nf, 1.2, nl, 20, 7)
    mono, k, d = ex.extract(img, lap)
    np.savez_compressed(os.path.join(OUT, name), image=img, nfeatures=nf, nlevels=nl, lap=np.array(lap), mono=mono,
                        keypoints=k.view(np.uint8).reshape(len(k), 28), descriptors=d,
                        level_sizes=np.array([ex.level(l).shape for l in range(nl)]))
    print(name, len(k), mono)


def stereo_case(name, w, h, nf, stream):
    L, R = synth.stereo_pair(w, h, stream)
    eL, eR = O.OracleExtractor(nf), O.OracleExtractor(nf)
    _, kL, dL = eL.extract(L)
    _, kR, dR = eR.extract(R)
    bf, b = np.float32(0.12) * np.float32(532.03), np.float32(0.12)
    u, dep = O.stereo_match(eL, eR, kL, dL, kR, dR, bf, b)
    idx, dist, ok = O.bf_knn2(dL, dR)
    prev = np.stack([kL["x"], kL["y"]], 1)
    n, m12, newprev = O.search_init(kL, dL, kR, dR, (0, 0, w, h), prev, 100, 0.9, True)
    np.savez_compressed(os.path.join(OUT, name), left=L, right=R, nfeatures=nf, bf=bf, b=b,
                        kL=kL.view(np.uint8).reshape(len(kL), 28), dL=dL, kR=kR.view(np.uint8).reshape(len(kR), 28),
                        dR=dR, uRight=u, depth=dep, knn_idx=idx, knn_dist=dist, knn_ok=ok, init_n=n, init_m12=m12,
                        init_prev=newprev)
    print(name, len(kL), len(kR), int((u >= 0).sum()), int(ok.sum()), n)


def projection_case(name, w, h, nf, stream):
    """SearchByProjection (local map + frame-to-frame, pinhole): seeded map-point / projected-point views."""
    L0, _ = synth.stereo_pair(w, h, stream, 0)
    L1, R1 = synth.stereo_pair(w, h, stream, 1)
    eP, eL, eR = O.OracleExtractor(nf), O.OracleExtractor(nf), O.OracleExtractor(nf)
    _, kp, dp = eP.extract(L0)
    _, kc, dc = eL.extract(L1)
    _, kr, dr = eR.extract(R1)
    uR, _ = O.stereo_match(eL, eR, kc, dc, kr, dr, np.float32(0.12) * np.float32(532.03), np.float32(0.12))
    rng = np.random.default_rng(2024)
    n = len(kp)
    sf = eL.tables()["scale"]
    flips = rng.random((n, 32, 8)) < 0.04
    desc = dp ^ np.packbits(flips, axis=2).reshape(n, 32)
    mps = np.zeros(n, O.MP_DTYPE)
    mps["proj_x"] = kp["x"] - 4 + rng.normal(0, 3.0, n)
    mps["proj_y"] = kp["y"] - 2 + rng.normal(0, 3.0, n)
    mps["proj_xr"] = mps["proj_x"] - rng.uniform(2, 60, n).astype(np.float32)
    mps["view_cos"] = rng.choice([0.9, 0.9985], n).astype(np.float32)
    mps["track_depth"] = rng.uniform(1, 80, n).astype(np.float32)
    mps["predicted_level"] = np.clip(kp["octave"] + rng.integers(-1, 2, n), 0, 7)
    mps["in_view"] = rng.random(n) < 0.9
    mps["bad"] = rng.random(n) < 0.05
    mps["has_observations"] = rng.random(n) < 0.85
    mps["desc"] = desc
    pts = np.zeros(n, O.PP_DTYPE)
    pts["u"], pts["v"], pts["ur"] = mps["proj_x"], mps["proj_y"], mps["proj_xr"]
    pts["radius"] = (np.float32(15.0) * sf[kp["octave"]]).astype(np.float32)
    pts["angle"] = kp["angle"]
    pts["min_level"], pts["max_level"] = kp["octave"] - 1, kp["octave"] + 1
    pts["valid"] = mps["in_view"]
    pts["has_observations"] = mps["has_observations"]
    pts["desc"] = desc
    occupied = (rng.random(len(kc)) < 0.05).astype(np.uint8)
    bounds = (0.0, 0.0, float(w), float(h))
    n1, m1, o1 = O.search_by_projection(kc, dc, uR, bounds, sf, mps, 3.0, True, 60.0, 0.8, occupied)
    n2, m2, o2 = O.search_by_projection_frame(kc, dc, uR, bounds, pts, True, occupied)
    np.savez_compressed(os.path.join(OUT, name), w=w, h=h, kc=kc.view(np.uint8).reshape(len(kc), 28), dc=dc, uR=uR,
                        scale=sf, mps=mps.view(np.uint8).reshape(n, 60), pts=pts.view(np.uint8).reshape(n, 64),
                        occupied=occupied, map_n=n1, map_match=m1, map_occ=o1, frame_n=n2, frame_match=m2, frame_occ=o2)
    print(name, len(kc), n, n1, n2)


def fisheye_case(name, seed):
    """ComputeStereoFishEyeMatches incl. KB8 triangulation (float: consumers compare with the tolerances stated in
    tests/test_fisheye.py); `gates` keeps the gated quantities so borderline decisions can be told apart."""
    sc = synth.fisheye_stereo_scene(seed)
    rig = O.kb8_rig(sc["cam1"], sc["cam2"], sc["R12"], sc["t12"])
    n, nd, l2r, r2l, dep, pts, gates = O.fisheye_stereo_match(sc["kL"], sc["dL"], sc["mono_left"], sc["kR"], sc["dR"],
                                                              sc["mono_right"], rig, sc["level_sigma2"])
    np.savez_compressed(os.path.join(OUT, name), kL=sc["kL"].view(np.uint8).reshape(-1, 28), dL=sc["dL"],
                        kR=sc["kR"].view(np.uint8).reshape(-1, 28), dR=sc["dR"], mono_left=sc["mono_left"],
                        mono_right=sc["mono_right"], rig=rig, level_sigma2=sc["level_sigma2"], n=n, nd=nd, l2r=l2r, r2l=r2l,
                        depth=dep, p3d=pts, gates=gates)
    print(name, n, nd)


def projection_fisheye_case(name, seed):
    """Both SearchByProjection matchers on a stereo-fisheye frame (Nleft != -1); the frame comes from the helper the
    parity tests use (tests/test_fisheye.py:_fisheye_frame)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_fisheye import _fisheye_frame
    f = _fisheye_frame(O, seed)
    n1, m1, o1 = O.search_by_projection_fisheye(f["kps"], f["desc"], f["nL"], f["bounds"], f["sf"], f["mps"], f["mpr"], 3.0, True, 60.0,
                                                0.8, f["l2r"], f["r2l"], f["occ"])
    n2, m2, o2 = O.search_by_projection_frame_fisheye(f["kps"], f["desc"], f["nL"], f["bounds"], f["pts"], f["uvr"], True, f["occ"])
    np.savez_compressed(os.path.join(OUT, name), kps=f["kps"].view(np.uint8).reshape(-1, 28), desc=f["desc"], n_left=f["nL"],
                        scale=f["sf"], mps=f["mps"].view(np.uint8).reshape(-1, 60), mpr=f["mpr"].view(np.uint8).reshape(-1, 16),
                        pts=f["pts"].view(np.uint8).reshape(-1, 64), uvr=f["uvr"], l2r=f["l2r"], r2l=f["r2l"], occ=f["occ"],
                        bounds=np.array(f["bounds"], np.float32), map_n=n1, map_match=m1, map_occ=o1, frame_n=n2,
                        frame_match=m2, frame_occ=o2)
    print(name, len(f["kps"]), n1, n2)


def undistort_case(name):
    """UndistortKeyPoints / ComputeImageBounds for the EuRoC (4 coefficients) and TUM1 (5 coefficients) cameras."""
    rng = np.random.default_rng(106)
    out = {}
    for tag, K, D, w, h in (("euroc", (458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05), 752, 480),
                            ("tum1", (517.306408, 516.469215, 318.643040, 255.313989), (0.262383, -0.953104, -0.005358, 0.002628, 1.163314), 640, 480)):
        k = np.zeros(400, O.KP_DTYPE)
        k["x"], k["y"] = rng.uniform(0, w, 400), rng.uniform(0, h, 400)
        k["size"], k["angle"], k["response"], k["octave"], k["class_id"] = 31, rng.uniform(0, 360, 400), 30, rng.integers(0, 8, 400), -1
        out[tag + "_K"], out[tag + "_D"], out[tag + "_size"] = np.array(K, np.float32), np.array(D, np.float32), np.array([w, h])
        out[tag + "_kps"] = k.view(np.uint8).reshape(-1, 28)
        out[tag + "_un"] = O.undistort_keypoints(k, K, D).view(np.uint8).reshape(-1, 28)
        out[tag + "_bounds"] = O.image_bounds(w, h, K, D)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, out["euroc_bounds"], out["tum1_bounds"])


def rectify_clahe_case(name):
    """cv::remap rectification (float maps, INTER_LINEAR) and CLAHE(3.0, 8x8) on a small frame, incl. a size that does not
    divide into the tiles and map entries outside the source."""
    from orb_slam3_fast_amd import synth
    img = synth.mono_frame(136, 100, 107)
    mx, my = synth.rectify_maps(120, 90, 136, 100, seed=5, k1=-0.35, rot_deg=(0.8, -1.1, 0.6))
    mx[0, :3], my[0, :3] = (np.nan, -40.0, 1e9), (5.0, 5.0, 5.0)
    eq = O.clahe(img, 3.0, (8, 8))
    out = {"img": img, "map_x": mx, "map_y": my, "clahe": eq, "remap": O.remap(img, mx, my), "chain": O.remap(eq, mx, my),
           "clahe_4x3_clip2": O.clahe(img, 2.0, (4, 3))}
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


def bow_case(name):
    """A small vocabulary tree (k = 4, L = 3, with early leaves and stopped words), ComputeBoW of a keyframe and a frame,
    SearchByBoW between them (mono and Nleft != -1)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_bow import _scene
    from orb_slam3_fast_amd import synth
    cols = synth.make_vocabulary(4, 3, seed=31, early_leaf_prob=0.1, stop_prob=0.05)
    voc = O.Vocabulary(4, 3, *cols)
    kd, ka, kv, fd, fa = _scene(cols, 180, 160, 31)
    (kw, kval), kfv = voc.transform(kd, 1)
    (fw, fval), ffv = voc.transform(fd, 1)
    n0, m0 = O.search_by_bow(kfv, kd, ka, kv, ffv, fd, fa, -1, 0.7, True)
    n1, m1 = O.search_by_bow(kfv, kd, ka, kv, ffv, fd, fa, 100, 0.7, True)
    out = {"parent": cols[0], "is_leaf": cols[1], "node_desc": cols[2], "weight": cols[3], "kf_desc": kd, "kf_angle": ka,
           "kf_valid": kv, "f_desc": fd, "f_angle": fa, "f_words": fw, "f_values": fval, "f_nodes": ffv[0], "f_start": ffv[1],
           "f_feats": ffv[2], "kf_nodes": kfv[0], "kf_start": kfv[1], "kf_feats": kfv[2], "n_mono": np.array(n0), "match_mono": m0,
           "n_fisheye": np.array(n1), "match_fisheye": m1}
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, len(fw), len(ffv[0]), n0, n1)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    extract_case("extract_160x120_L3.npz", 160, 120, 300, 3, 101, (0, 0))
    extract_case("extract_384x288_L8.npz", 384, 288, 500, 8, 102, (0, 0))
    extract_case("extract_384x288_L8_lap.npz", 384, 288, 500, 8, 102, (100, 250))
    stereo_case("stereo_400x300.npz", 400, 300, 600, 103)
    projection_case("projection_480x360.npz", 480, 360, 800, 104)
    fisheye_case("fisheye_stereo.npz", 105)
    undistort_case("undistort.npz")
    projection_fisheye_case("fisheye_projection.npz", 7)
    rectify_clahe_case("rectify_clahe.npz")
    bow_case("bow.npz")
