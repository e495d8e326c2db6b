"""Property tests (hypothesis) of the oracle's kernels: invariants that hold for any input, complementing the KATs."""
import numpy as np
import pytest

hyp = pytest.importorskip("hypothesis")
from hypothesis import given, settings  # noqa: E402
from hypothesis import strategies as st  # noqa: E402
from hypothesis.extra import numpy as hnp  # noqa: E402

SET = settings(max_examples=40, deadline=None)
desc32 = hnp.arrays(np.uint8, (32,))


@SET
@given(desc32, desc32, desc32)
def test_hamming_is_a_metric(oracle, a, b, c):
    h = oracle.hamming
    assert h(a, a) == 0 and h(a, b) == h(b, a) == int(np.unpackbits(a ^ b).sum())
    assert h(a, c) <= h(a, b) + h(b, c) and 0 <= h(a, b) <= 256


@SET
@given(st.integers(0, 255), st.integers(8, 60), st.integers(8, 60), st.integers(5, 70), st.integers(5, 70))
def test_resize_and_blur_keep_constants(oracle, v, w, h, dw, dh):
    img = np.full((h, w), v, np.uint8)
    assert (oracle.resize(img, dw, dh) == v).all() and (oracle.blur(img) == v).all()


@SET
@given(hnp.arrays(np.uint8, st.tuples(st.integers(9, 30), st.integers(9, 30))))
def test_blur_stays_within_the_local_range(oracle, img):
    out = oracle.blur(img).astype(int)
    pad = np.pad(img.astype(int), 3, mode="reflect")
    h, w = img.shape
    win = np.lib.stride_tricks.sliding_window_view(pad, (7, 7))
    assert (out >= win.min((2, 3))).all() and (out <= win.max((2, 3))).all()


@SET
@given(hnp.arrays(np.uint8, st.tuples(st.integers(12, 28), st.integers(12, 28))), st.integers(5, 60), st.integers(1, 40))
def test_fast_threshold_monotonicity_and_scores(oracle, img, t, dt):
    lo = oracle.fast(img, t, nms=False)
    hi = oracle.fast(img, t + dt, nms=False)
    s_lo = {(x, y) for x, y, _ in lo.tolist()}
    s_hi = {(x, y) for x, y, _ in hi.tolist()}
    assert s_hi <= s_lo                                   # raising the threshold never creates a corner
    nms = oracle.fast(img, t, nms=True)
    assert {(x, y) for x, y, _ in nms.tolist()} <= s_lo   # suppression only removes
    assert all(s >= t for _, _, s in nms.tolist())         # cornerScore = M - 1 >= t  <=>  M > t
    sc = {(x, y): s for x, y, s in nms.tolist()}
    assert all(((x, y) in s_hi) == (s >= t + dt) for (x, y), s in sc.items())  # score >= t'  <=>  corner at t'
    h, w = img.shape
    assert all(3 <= x < w - 3 and 3 <= y < h - 3 for x, y in s_lo)


@SET
@given(hnp.arrays(np.uint8, st.tuples(st.integers(1, 12), st.just(32))), hnp.arrays(np.uint8, st.tuples(st.integers(0, 12), st.just(32))))
def test_bf_knn2_properties(oracle, q, t):
    idx, dist, ok = oracle.bf_knn2(q, t)
    for i in range(len(q)):
        d = [oracle.hamming(q[i], t[j]) for j in range(len(t))]
        if len(t) >= 1:
            assert dist[i, 0] == min(d) and idx[i, 0] == d.index(min(d))   # first minimum
        else:
            assert idx[i, 0] == -1
        if len(t) >= 2:
            rest = d[: idx[i, 0]] + d[idx[i, 0] + 1:]
            assert dist[i, 1] == min(rest) and dist[i, 0] <= dist[i, 1]
            assert bool(ok[i]) == (np.float32(dist[i, 0]) < np.float32(dist[i, 1]) * 0.7)
        else:
            assert idx[i, 1] == -1 and not ok[i]


@SET
@given(st.integers(0, 2 ** 31 - 1), st.floats(-30, 700), st.floats(-30, 500), st.sampled_from([3.0, 11.0, 40.0, 150.0]),
       st.integers(-1, 3), st.integers(-1, 7))
def test_features_in_area_equals_brute_force_over_the_cells_it_visits(oracle, seed, x, y, r, lo, hi):
    rng = np.random.default_rng(seed)
    n = 300
    k = np.zeros(n, oracle.KP_DTYPE)
    k["x"], k["y"], k["octave"] = rng.uniform(0, 640, n), rng.uniform(0, 480, n), rng.integers(0, 8, n)
    got = oracle.features_in_area(k, (0.0, 0.0, 640.0, 480.0), x, y, r, lo, hi).tolist()
    x32, y32, r32 = np.float32(x), np.float32(y), np.float32(r)
    inside = (np.abs(k["x"] - x32) < r32) & (np.abs(k["y"] - y32) < r32)
    if lo > 0 or hi >= 0:
        inside &= k["octave"] >= lo
        if hi >= 0:
            inside &= k["octave"] <= hi
    # every reported keypoint satisfies the window + level test; keypoints inside the window are reported unless the
    # grid rounding (PosInGrid rounds to the NEAREST cell) puts them in a cell outside the scanned block
    assert all(inside[i] for i in got) and len(set(got)) == len(got)
    assert len(got) >= 0.7 * int(inside.sum()) - 2


# ---- pre-processing and bag of words ----------------------------------------------------------------
@SET
@given(st.integers(0, 255), st.integers(9, 40), st.integers(9, 40), st.integers(0, 2 ** 31 - 1))
def test_remap_keeps_constants_inside_and_zero_outside(oracle, v, w, h, seed):
    """The four weights of every table entry sum to 2^15, so a constant image stays constant wherever all four taps are
    inside the source; a map that points beyond the border by more than a pixel gives the border value 0."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), v, np.uint8)
    mx = rng.uniform(0, w - 1.0001, (11, 13)).astype(np.float32)
    my = rng.uniform(0, h - 1.0001, (11, 13)).astype(np.float32)
    sx, sy = np.rint(mx * np.float32(32)).astype(int) >> 5, np.rint(my * np.float32(32)).astype(int) >> 5
    inside = (sx + 1 < w) & (sy + 1 < h)  # all four taps of the pixel lie in the source
    out = oracle.remap(img, mx, my)
    assert (out[inside] == v).all()
    far = oracle.remap(img, mx + np.float32(w + 2), my)
    assert not far.any()


@SET
@given(hnp.arrays(np.uint8, st.tuples(st.integers(10, 40), st.integers(10, 40))), st.integers(-6, 6), st.integers(-6, 6))
def test_remap_integer_translation_is_a_shift(oracle, img, dx, dy):
    h, w = img.shape
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    out = oracle.remap(img, u + np.float32(dx), v + np.float32(dy))
    want = np.zeros_like(img)
    ys, xs = np.arange(h) + dy, np.arange(w) + dx
    oky, okx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
    want[np.ix_(oky, okx)] = img[np.ix_(ys[oky], xs[okx])]
    assert np.array_equal(out, want)


@SET
@given(hnp.arrays(np.uint8, st.tuples(st.integers(16, 48), st.integers(16, 48))), st.sampled_from([0.0, 1.0, 3.0, 40.0]),
       st.integers(1, 4), st.integers(1, 4))
def test_clahe_flat_images_stay_flat_and_global_equalisation_preserves_rank(oracle, img, clip, tx, ty):
    """A flat image has one occupied bin in every tile, so all luts agree on it; with one tile and no clipping CLAHE is
    plain histogram equalisation, a non-decreasing function of the grey value."""
    out = oracle.clahe(img, clip, (tx, ty))
    assert out.shape == img.shape and out.dtype == np.uint8
    flat = np.full_like(img, int(img[0, 0]))
    of = oracle.clahe(flat, clip, (tx, ty))
    assert (of == of[0, 0]).all()
    if clip == 0.0 and tx == 1 and ty == 1:  # plain global equalisation: rank preserving
        order = np.argsort(img.ravel(), kind="stable")
        assert (np.diff(out.ravel()[order].astype(int)) >= 0).all()


@SET
@given(st.integers(0, 2 ** 31 - 1), st.integers(2, 6), st.integers(1, 3), st.integers(0, 4))
def test_bow_vector_is_invariant_under_feature_permutation(oracle, seed, k, L, levelsup):
    """A word's value is n sequential additions of the same weight and the norm runs over ascending word ids, so the
    BowVector does not depend on the order of the features -- bit for bit; the FeatureVector permutes with them."""
    from orb_slam3_fast_amd import synth
    cols = synth.make_vocabulary(k, L, seed=seed % 1000, early_leaf_prob=0.1, stop_prob=0.1)
    voc = oracle.Vocabulary(k, L, *cols)
    feats = synth.vocabulary_features(cols, 60, seed=seed % 997)
    perm = np.random.default_rng(seed).permutation(len(feats))
    (w0, v0), (n0, s0, f0) = voc.transform(feats, levelsup)
    (w1, v1), (n1, s1, f1) = voc.transform(feats[perm], levelsup)
    assert np.array_equal(w0, w1) and np.array_equal(v0.view(np.uint64), v1.view(np.uint64))
    assert np.array_equal(n0, n1) and np.array_equal(s0, s1)
    for j in range(len(n0)):
        assert sorted(perm[f1[s1[j]:s1[j + 1]]].tolist()) == f0[s0[j]:s0[j + 1]].tolist()
    if len(v0):
        assert abs(v0.sum() - 1.0) < 1e-12 and (v0 > 0).all()


@SET
@given(st.integers(0, 2 ** 31 - 1), st.sampled_from([-1, 0, 30, 80]), st.booleans())
def test_search_by_bow_invariants(oracle, seed, n_left, check_ori):
    """Whatever the inputs: a matched pair shares its vocabulary node, the keyframe feature holds a good map point and passed
    TH_LOW, and the returned count is the number of matched frame features."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_bow import _scene
    from orb_slam3_fast_amd import synth
    cols = synth.make_vocabulary(4, 3, seed=seed % 50)
    voc = oracle.Vocabulary(4, 3, *cols)
    kd, ka, kv, fd, fa = _scene(cols, 90, 80, seed % 1000, n_left)
    kf_fv, f_fv = voc.transform(kd, 1)[1], voc.transform(fd, 1)[1]
    n, m = oracle.search_by_bow(kf_fv, kd, ka, kv, f_fv, fd, fa, n_left, 0.7, check_ori)
    assert n == int((m >= 0).sum())
    node_of_kf = {int(i): int(kf_fv[0][j]) for j in range(len(kf_fv[0])) for i in kf_fv[2][kf_fv[1][j]:kf_fv[1][j + 1]]}
    node_of_f = {int(i): int(f_fv[0][j]) for j in range(len(f_fv[0])) for i in f_fv[2][f_fv[1][j]:f_fv[1][j + 1]]}
    for i_f in np.flatnonzero(m >= 0):
        ikf = int(m[i_f])
        assert kv[ikf] and node_of_kf[ikf] == node_of_f[int(i_f)]
        d = int(np.unpackbits(kd[ikf] ^ fd[i_f]).sum())
        assert d <= 50  # TH_LOW gates both eyes (the right eye additionally needs the LEFT best below it, :349-363)
