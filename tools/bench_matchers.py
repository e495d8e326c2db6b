#!/usr/bin/env python3
"""Wall-clock of the matcher entry points of the C ABI (host arrays in, host results out, one call at a time -- the way
Tracking calls them once per frame) next to the CPU oracle on the same inputs.  Inputs: keypoints / descriptors of a
synthetic 1280x720 stereo stream extracted with 1500 features, seeded map-point / projected-point views as in
tools/gen_golden.py.  usage: python tools/bench_matchers.py [reps]   (GPU box; the oracle is only the comparison)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth
from oracle import oracle_py as O

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w, h, nf = 1280, 720, 1500
L0, _ = synth.stereo_pair(w, h, 300, 0)
L1, R1 = synth.stereo_pair(w, h, 300, 1)
eP, eL, eR = (orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h) for _ in range(3))
_, kp, dp = eP(L0)
_, kc, dc = eL(L1)
_, kr, dr = eR(R1)
uR = orbx.ComputeStereoMatches(eL, eR, 0.12 * 532.03, 0.12)[0][0, : len(kc)]
sf = eL.GetScaleFactors()
rng = np.random.default_rng(2024)
n = len(kp)
flips = rng.random((n, 32, 8)) < 0.04
desc = dp ^ np.packbits(flips, axis=2).reshape(n, 32)
mps = np.zeros(n, orbx.MP_DTYPE)
mps["proj_x"], mps["proj_y"] = kp["x"] - 4 + rng.normal(0, 3.0, n), kp["y"] - 2 + rng.normal(0, 3.0, n)
mps["proj_xr"] = mps["proj_x"] - rng.uniform(2, 60, n).astype(np.float32)
mps["view_cos"], mps["track_depth"] = rng.choice([0.9, 0.9985], n), rng.uniform(1, 80, n)
mps["predicted_level"] = np.clip(kp["octave"] + rng.integers(-1, 2, n), 0, 7)
mps["in_view"], mps["bad"], mps["has_observations"] = rng.random(n) < 0.9, rng.random(n) < 0.05, rng.random(n) < 0.85
mps["desc"] = desc
pts = np.zeros(n, orbx.PP_DTYPE)
pts["u"], pts["v"], pts["ur"] = mps["proj_x"], mps["proj_y"], mps["proj_xr"]
pts["radius"], pts["angle"] = (np.float32(15.0) * sf[kp["octave"]]), kp["angle"]
pts["min_level"], pts["max_level"] = kp["octave"] - 1, kp["octave"] + 1
pts["valid"], pts["has_observations"], pts["desc"] = mps["in_view"], mps["has_observations"], desc
occ = (rng.random(len(kc)) < 0.05).astype(np.uint8)
bounds = (0.0, 0.0, float(w), float(h))
prev = np.stack([kp["x"], kp["y"]], 1)
sc = synth.fisheye_stereo_scene(0, 1500, 1500, 500, 500)
rig = orbx.kb8_rig(sc["cam1"], sc["cam2"], sc["R12"], sc["t12"])
omps, opts = mps.view(O.MP_DTYPE), pts.view(O.PP_DTYPE)
m = orbx.ORBmatcher(0.8, True)
mi = orbx.ORBmatcher(0.9, True)
cases = [
    ("SearchByProjection(F, MapPoints)  %d pts x %d kps" % (n, len(kc)),
     lambda: m.SearchByProjection(kc, dc, uR, bounds, sf, mps, occ, 3.0, True, 60.0),
     lambda: O.search_by_projection(kc, dc, uR, bounds, sf, omps, 3.0, True, 60.0, 0.8, occ)),
    ("SearchByProjection(Cur, Last)     %d pts x %d kps" % (n, len(kc)),
     lambda: m.SearchByProjectionFrame(kc, dc, uR, bounds, pts, occ),
     lambda: O.search_by_projection_frame(kc, dc, uR, bounds, opts, True, occ)),
    ("SearchForInitialization           %d x %d kps, window 100" % (n, len(kc)),
     lambda: mi.SearchForInitialization(kp, dp, kc, dc, bounds, prev, 100),
     lambda: O.search_init(kp, dp, kc, dc, bounds, prev, 100, 0.9, True)),
    ("BFMatcher kNN-2 + ratio           %d x %d" % (len(kc), len(kr)),
     lambda: orbx.bf_knn2(dc, dr), lambda: O.bf_knn2(dc, dr)),
    ("ComputeStereoFishEyeMatches       1000 x 1000 lapping rows",
     lambda: orbx.ComputeStereoFishEyeMatches(sc["kL"], sc["dL"], 500, sc["kR"], sc["dR"], 500, rig, sc["level_sigma2"]),
     lambda: O.fisheye_stereo_match(sc["kL"], sc["dL"], 500, sc["kR"], sc["dR"], 500, rig, sc["level_sigma2"])),
]
# stereo-fisheye flavours: frame = current-left | current-right keypoints, partners from the brute-force association
kk, dd = np.concatenate([kc, kr]), np.concatenate([dc, dr])
nLk, nRk = len(kc), len(kr)
bi, bd, bok = orbx.bf_knn2(dc, dr)
l2r = np.where(bok.astype(bool), bi[:, 0], -1).astype(np.int32)
r2l = np.full(nRk, -1, np.int32)
r2l[l2r[l2r >= 0]] = np.nonzero(l2r >= 0)[0]
mpr = np.zeros(n, orbx.MPR_DTYPE)
mpr["proj_yr"], mpr["view_cos_r"] = mps["proj_y"], mps["view_cos"]
mpr["predicted_level_r"], mpr["in_view_r"] = mps["predicted_level"], mps["in_view"]
occf = (rng.random(nLk + nRk) < 0.05).astype(np.uint8)
uvr = np.stack([mps["proj_xr"], mps["proj_y"]], 1).astype(np.float32)
cases += [
    ("SearchByProjection(F, MapPoints), Nleft != -1  %d pts x %d+%d kps" % (n, nLk, nRk),
     lambda: m.SearchByProjectionFisheye(kk, dd, nLk, bounds, sf, mps, mpr, l2r, r2l, occf, 3.0, True, 60.0),
     lambda: O.search_by_projection_fisheye(kk, dd, nLk, bounds, sf, omps, mpr.view(O.MPR_DTYPE), 3.0, True, 60.0, 0.8, l2r, r2l, occf)),
    ("SearchByProjection(Cur, Last), Nleft != -1     %d pts x %d+%d kps" % (n, nLk, nRk),
     lambda: m.SearchByProjectionFrameFisheye(kk, dd, nLk, bounds, pts, uvr, occf),
     lambda: O.search_by_projection_frame_fisheye(kk, dd, nLk, bounds, opts, uvr, True, occf)),
]
# bag of words: a full 10^6-word tree (the size of ORBvoc.txt), ComputeBoW of the current frame, SearchByBoW keyframe -> frame
vcols = synth.make_vocabulary_bfs(10, 6, seed=1)
voc, ovoc = orbx.ORBVocabulary(10, 6, *vcols), O.Vocabulary(10, 6, *vcols)
kf_fv, f_fv = voc.transform(dp, 4)[1], voc.transform(dc, 4)[1]
kvalid = (rng.random(n) < 0.8).astype(np.uint8)
mb = orbx.ORBmatcher(0.7, True)
cases += [
    ("Frame::ComputeBoW (transform, 10^6 words, levelsup 4)  %d descriptors" % len(dc),
     lambda: voc.transform(dc, 4), lambda: ovoc.transform(dc, 4)),
    ("SearchByBoW(KeyFrame, Frame)      %d x %d kps, %d / %d nodes" % (n, len(kc), len(kf_fv[0]), len(f_fv[0])),
     lambda: orbx.SearchByBoW(kf_fv, kp, dp, kvalid, f_fv, kc, dc, -1, 0.7, True),
     lambda: O.search_by_bow(kf_fv, dp, kp["angle"], kvalid, f_fv, dc, kc["angle"], -1, 0.7, True)),
]
# round 3: relocalisation search, SearchForTriangulation on the two frames' feature vectors, Fuse's search
occk = (rng.random(len(kc)) < 0.3).astype(np.uint8)
hm1, hm2 = (rng.random(n) < 0.25).astype(np.uint8), (rng.random(len(kc)) < 0.25).astype(np.uint8)
sigma2 = (sf * sf).astype(np.float32)
F12 = np.array([[1e-6, 2e-6, 0.45], [-2e-6, 1e-6, -0.9], [-0.45, 0.9, 3.0]], np.float32)
epi = np.array([0.5 * w, 0.45 * h], np.float32)
fpts = np.zeros(n, orbx.FP_DTYPE)
fpts["u"], fpts["v"], fpts["ur"] = mps["proj_x"], mps["proj_y"], mps["proj_xr"]
fpts["predicted_level"] = mps["predicted_level"]
fpts["radius"] = np.float32(3.0) * sf[mps["predicted_level"]]
fpts["valid"], fpts["desc"] = mps["in_view"], desc
inv_sigma2 = (1.0 / sigma2).astype(np.float32)
cases += [
    ("SearchByProjection(Cur, KeyFrame, ORBdist) (relocalisation)  %d pts x %d kps" % (n, len(kc)),
     lambda: m.SearchByProjectionKeyFrame(kc, dc, bounds, pts, occk, 100),
     lambda: O.search_by_projection_keyframe(kc, dc, bounds, opts, 100, True, occk)),
    ("SearchForTriangulation(KF1, KF2)  %d x %d kps, %d / %d nodes" % (n, len(kc), len(kf_fv[0]), len(f_fv[0])),
     lambda: mb.SearchForTriangulation(kf_fv, kp, dp, hm1, None, f_fv, kc, dc, hm2, uR, sf, sigma2, epi, F12),
     lambda: O.search_for_triangulation(kf_fv, kp, dp, hm1, None, f_fv, kc, dc, hm2, uR, sf, sigma2, epi, F12)),
    ("Fuse(KF, MapPoints): the search   %d pts x %d kps" % (n, len(kc)),
     lambda: m.FuseSearch(kc, dc, uR, bounds, inv_sigma2, fpts),
     lambda: O.fuse_search(kc, dc, uR, bounds, inv_sigma2, fpts.view(O.FP_DTYPE))),
    ("SearchByBoW(KeyFrame, KeyFrame)   %d x %d kps, %d / %d nodes" % (n, len(kc), len(kf_fv[0]), len(f_fv[0])),
     lambda: orbx.SearchByBoWKeyFrames(kf_fv, kp, dp, kvalid, f_fv, kc, dc, 1 - hm2, 0.75, True),
     lambda: O.search_by_bow_keyframes(kf_fv, dp, kp["angle"], kvalid, f_fv, dc, kc["angle"], 1 - hm2, 0.75, True)),
    ("SearchBySim3(KF1, KF2): two searches + agreement  %d + %d pts" % (n, n),
     lambda: m.SearchBySim3(kc, dc, bounds, kc, dc, bounds, fpts, fpts),
     lambda: (O.fuse_search(kc, dc, None, bounds, np.zeros(8, np.float32), fpts.view(O.FP_DTYPE), 100),
              O.fuse_search(kc, dc, None, bounds, np.zeros(8, np.float32), fpts.view(O.FP_DTYPE), 100))),
]
import gc
gc.collect()
gc.disable()  # a generation-2 collection inside a millisecond timing window would dominate it
print("%-62s %12s %12s %8s" % ("entry point (host API, one call)", "MI355X ms", "oracle ms", "ratio"))
for name, fg, fo in cases:
    for _ in range(3):
        fg()
    t0 = time.perf_counter()
    for _ in range(reps):
        fg()
    tg = (time.perf_counter() - t0) / reps * 1e3
    ro = max(3, reps // 6)
    t0 = time.perf_counter()
    for _ in range(ro):
        fo()
    to = (time.perf_counter() - t0) / ro * 1e3
    print("%-62s %12.3f %12.3f %8.1f" % (name, tg, to, to / tg))

# batched ComputeBoW on device-resident extraction results (64 images, one launch pair)
from orb_slam3_fast_amd.hipmem import DeviceBuffer
B = 64
imgs = DeviceBuffer.from_numpy(np.stack([L1, R1] * (B // 2)))
exb = orbx.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
exb.extract_batch_device(imgs.ptr.value, B, w, h, w, w * h)
exb.sync()
for _ in range(3):
    voc.transform_batch(exb, 4)
    exb.sync()
t0 = time.perf_counter()
for _ in range(reps):
    voc.transform_batch(exb, 4)
    exb.sync()
tb = (time.perf_counter() - t0) / reps * 1e3
print("%-62s %12.3f   (%.1f us per image, results stay in HBM)" % ("ComputeBoW, batch of %d extracted images" % B, tb, tb * 1e3 / B))

# batched SearchByProjection on the frames of an extraction batch (round 4): 32 frames, ~1500 points each, keypoints and
# descriptors device-resident; one call = one upload of the points, one launch per kernel of the chain, one download
FB = 32
stride = len(pts)
ptsB = np.stack([pts] * FB)
mpsB = np.stack([mps] * FB)
nptsB = np.full(FB, len(pts), np.int32)
for i in range(FB):   # (different points per frame: shift the projections a little)
    ptsB[i]["u"] += np.float32(0.25 * i)
    mpsB[i]["proj_x"] += np.float32(0.25 * i)
occB = np.stack([np.pad(occ, (0, exb.capacity - len(occ)))] * FB)
for name, fn in (("SearchByProjection(Cur, Last), batch of %d frames" % FB,
                  lambda: m.SearchByProjectionFrameBatch(exb, 0, FB, bounds, ptsB, nptsB, occB)),
                 ("SearchByProjection(F, MapPoints), batch of %d frames" % FB,
                  lambda: m.SearchByProjectionBatch(exb, 0, FB, bounds, mpsB, nptsB, occB, 3.0, True, 60.0))):
    for _ in range(3):
        r = fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    tb = (time.perf_counter() - t0) / reps * 1e3
    print("%-62s %12.3f   (%.3f ms per frame, %d matches in frame 0; one-shot call above: per frame)" % (name, tb, tb / FB, int(r[0][0])))

# the same local-map matcher with the projection ON THE DEVICE (round 6): the map (~1500 points) is uploaded once as SoA, a call sends
# 32 poses (80 B each), isInFrustum + PredictScale run in k_project_map, the views never exist on the host
rngp = np.random.default_rng(5)
nmap = len(mps)
fxp = 420.0
posM = np.stack([(mps["proj_x"] - w / 2) / fxp * 8.0, (mps["proj_y"] - h / 2) / fxp * 8.0, np.full(nmap, 8.0)], 1).astype(np.float32)
nrmM = np.tile(np.array([0, 0, 1], np.float32), (nmap, 1))
maxM = (8.0 * 1.2 ** np.clip(mps["predicted_level"], 0, 7)).astype(np.float32)
minM = (maxM / 1.2 ** 7).astype(np.float32)
flagsM = (mps["bad"] | (mps["has_observations"] << 1)).astype(np.uint8)
posesB = np.stack([np.concatenate([np.eye(3).reshape(-1), [0.002 * i, 0, 0], [-0.002 * i, 0, 0], [fxp, fxp, w / 2, h / 2, 0.12 * fxp]])
                   for i in range(FB)]).astype(np.float32)
exb.map_upload(posM, nrmM, minM, maxM, mps["desc"], flagsM)
def dev_step():
    exb.project_map_points(posesB, bounds, 0.5)
    return m.SearchByProjectionBatchDevice(exb, 0, FB, bounds, occB, 3.0, True, 60.0)
for _ in range(3):
    r = dev_step()
t0 = time.perf_counter()
for _ in range(reps):
    exb.project_map_points(posesB, bounds, 0.5)
tp = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
for _ in range(reps):
    r = dev_step()
tb = (time.perf_counter() - t0) / reps * 1e3
print("%-62s %12.3f   (%.3f ms per frame, of which isInFrustum + PredictScale of %d points x %d poses on the device %.3f ms; %d matches "
      "in frame 0; per call %d B of poses instead of %d B of host-projected views)"
      % ("SearchByProjection(F, MapPoints), device-side projection, %d frames" % FB, tb, tb / FB, nmap, FB, tp, int(r[0][0]), posesB.nbytes,
         mpsB.nbytes))

# the frame-to-frame matcher with the projection ON THE DEVICE (round 6b): the LastFrames' points (world position, octave, angle,
# descriptor) are uploaded once per step as SoA, a call sends 32 poses (52 B each), the projection block of
# SearchByProjection(CurrentFrame, LastFrame) runs in k_project_last, the views never exist on the host
npl = len(pts)
zl = 8.0
posL = np.stack([np.stack([(pts["u"] + np.float32(0.25 * i) - w / 2) / fxp * zl, (pts["v"] - h / 2) / fxp * zl, np.full(npl, zl)], 1)
                 for i in range(FB)]).astype(np.float32)
octL = np.stack([np.clip(pts["min_level"] + 1, 0, 7)] * FB).astype(np.int32)
angL = np.stack([pts["angle"]] * FB).astype(np.float32)
descL = np.stack([pts["desc"]] * FB)
flagsL = np.stack([(pts["valid"] | (pts["has_observations"] << 1)).astype(np.uint8)] * FB)
posesQ = np.stack([np.concatenate([[0, 0, 0, 1], [0, 0, 0], [fxp, fxp, w / 2, h / 2, 0.12 * fxp]]) for i in range(FB)]).astype(np.float32)
dirsQ = np.zeros(FB, np.int32)
exb.last_frames_upload(np.full(FB, npl, np.int32), posL, octL, angL, descL, flagsL)
thL = float(pts["radius"][pts["valid"] != 0][0] / exb.GetScaleFactors()[int(octL[0][pts["valid"] != 0][0])]) if (pts["valid"] != 0).any() else 7.0
def dev_step_frame():
    exb.project_last_frames(posesQ, dirsQ, bounds, thL)
    return m.SearchByProjectionFrameBatchDevice(exb, 0, FB, bounds, occB)
for _ in range(3):
    r = dev_step_frame()
t0 = time.perf_counter()
for _ in range(reps):
    exb.project_last_frames(posesQ, dirsQ, bounds, thL)
tp = (time.perf_counter() - t0) / reps * 1e3
t0 = time.perf_counter()
for _ in range(reps):
    r = dev_step_frame()
tb = (time.perf_counter() - t0) / reps * 1e3
print("%-62s %12.3f   (%.3f ms per frame, of which the projection of %d points x %d poses on the device %.3f ms; %d matches in frame 0; "
      "per call %d B of poses instead of %d B of host-projected points)"
      % ("SearchByProjection(Cur, Last), device-side projection, %d frames" % FB, tb, tb / FB, npl, FB, tp, int(r[0][0]),
         posesQ.nbytes + dirsQ.nbytes, ptsB.nbytes))

# batched SearchForInitialization on the frames of an extraction batch (round 5): F1 = the previous frame's keypoints from the
# host for every pair, F2 = the batch's images (exb holds L1, R1 alternating: even images are the frame `kc` came from)
k1B, d1B, prevB = [kp] * FB, [dp] * FB, [prev] * FB
fn = lambda: mi.SearchForInitializationBatch(exb, 0, k1B, d1B, bounds, prevB, 100)
for _ in range(3):
    r = fn()
t0 = time.perf_counter()
for _ in range(reps):
    r = fn()
tb = (time.perf_counter() - t0) / reps * 1e3
print("%-62s %12.3f   (%.3f ms per pair, %d matches in pair 0; Python packing of the %d host arrays included)"
      % ("SearchForInitialization, batch of %d pairs" % FB, tb, tb / FB, int(r[0][0]), FB))

# batched SearchByBoW(KeyFrame, Frame) on the frames of an extraction batch (round 5): the frames' feature vectors were computed on
# the device above (voc.transform_batch(exb, 4)); the key frame of every pair comes from the host
voc.transform_batch(exb, 4)
exb.sync()
fn = lambda: orbx.SearchByBoWBatch(exb, 0, [kf_fv] * FB, [kp] * FB, [dp] * FB, [kvalid] * FB, -1, 0.7, True)
for _ in range(3):
    r = fn()
t0 = time.perf_counter()
for _ in range(reps):
    r = fn()
tb = (time.perf_counter() - t0) / reps * 1e3
print("%-62s %12.3f   (%.3f ms per pair, %d matches in pair 0; Python packing of the %d key frames included)"
      % ("SearchByBoW(KeyFrame, Frame), batch of %d pairs" % FB, tb, tb / FB, int(r[0][0]), FB))

# batched stereo-fisheye SearchByProjection (round 5): 32 two-camera frames, left images 0..31, right images 32..63 of the batch
imgs2 = DeviceBuffer.from_numpy(np.stack([L1] * FB + [R1] * FB))
exb.extract_batch_device(imgs2.ptr.value, 2 * FB, w, h, w, w * h)
exb.sync()
capb = exb.capacity
l2rB, r2lB = np.full((FB, capb), -1, np.int32), np.full((FB, capb), -1, np.int32)
l2rB[:, :nLk], r2lB[:, :nRk] = l2r, r2l
occfB = np.zeros((FB, 2 * capb), np.uint8)
occfB[:, :nLk + nRk] = occf
mprB, uvrB = np.stack([mpr] * FB), np.stack([uvr] * FB)
for name, fn in (("SearchByProjection(F, MapPoints), Nleft != -1, batch of %d frames" % FB,
                  lambda: m.SearchByProjectionFisheyeBatch(exb, 0, FB, FB, bounds, mpsB, mprB, nptsB, l2rB, r2lB, occfB, 3.0, True, 60.0)),
                 ("SearchByProjection(Cur, Last), Nleft != -1, batch of %d frames" % FB,
                  lambda: m.SearchByProjectionFrameFisheyeBatch(exb, 0, FB, FB, bounds, ptsB, uvrB, nptsB, occfB))):
    for _ in range(3):
        r = fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    tb = (time.perf_counter() - t0) / reps * 1e3
    print("%-62s %12.3f   (%.3f ms per frame, %d matches in frame 0)" % (name, tb, tb / FB, int(r[0][0])))
