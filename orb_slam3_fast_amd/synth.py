"""Deterministic synthetic camera streams for tests and bench.py (SURVEY.md §8d image spec).

A frame is a mid-grey canvas with 3-octave value noise, random filled rectangles / discs / triangles and
8x8 checker patches.  Every object carries an integer disparity, so the right view of a rectified stereo
pair is the same scene with each object shifted left by its disparity (nearer objects occlude farther
ones in both views).  Frame f of a stream is frame 0 panned by an integer offset (for
SearchForInitialization inputs).

Disparity follows a tilted ground plane (4 + 60*y/h px, +0..3 px jitter per object: lower = nearer), the
background is sheared by the same plane, and every view gets independent +-2 sensor noise, so most
corners are physical points that re-appear in the right view while FAST/NMS ties are broken as in real
camera images.

All randomness comes from splitmix64 seeded with 0x20220131 ^ (stream << 32 | frame-independent id), so
images are identical on every machine (no dependence on numpy's RNG streams).
"""
import numpy as np

SEED0 = 0x20220131
_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed):
        self.s = seed & _M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def randint(self, lo, hi):
        """uniform integer in [lo, hi]"""
        return lo + self.next() % (hi - lo + 1)


def _hash_u64(a):
    """vectorised splitmix64 finaliser on a uint64 array"""
    with np.errstate(over="ignore"):
        z = a + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def plane_disp(y, h):
    """integer ground-plane disparity of image row y"""
    return 4 + (60 * min(max(int(y), 0), h - 1)) // h


def _value_noise(w, h, seed, xoff):
    """3-octave bilinear value noise in [-12, 12]; row y is sampled at x + xoff[y] (views are sheared)."""
    out = np.zeros((h, w), np.float64)
    amp = [7.0, 3.5, 1.5]
    for o, cell in enumerate((48, 16, 5)):
        gx = (np.arange(w)[None, :] + np.asarray(xoff)[:, None] + 4096) / cell
        gy = ((np.arange(h) + 4096) / cell)[:, None] + np.zeros((1, w))
        x0 = np.floor(gx).astype(np.int64)
        y0 = np.floor(gy).astype(np.int64)
        fx = gx - x0
        fy = gy - y0

        def lat(yy, xx):
            k = (yy.astype(np.uint64) * np.uint64(1000003) + xx.astype(np.uint64)
                 + np.uint64((seed * 31 + o * 7919) & _M64))
            return (_hash_u64(k) >> np.uint64(40)).astype(np.float64) / float(1 << 24) * 2.0 - 1.0

        v = (lat(y0, x0) * (1 - fx) * (1 - fy) + lat(y0, x0 + 1) * fx * (1 - fy)
             + lat(y0 + 1, x0) * (1 - fx) * fy + lat(y0 + 1, x0 + 1) * fx * fy)
        out += amp[o] * v
    return out


def make_scene(w, h, stream=0, n_objects=None):
    """Object list of a stream: dicts with kind, geometry, intensity, disparity (sorted far -> near)."""
    rng = SplitMix64(SEED0 ^ (stream << 32))
    if n_objects is None:
        n_objects = int(400 + (800 * w * h) // (1280 * 720))
        n_objects = max(120, min(1200, n_objects))
    objs = []
    for _ in range(n_objects):
        kind = rng.randint(0, 3)  # 0 rect, 1 disc, 2 triangle, 3 checker
        size = rng.randint(6, 80)
        o = dict(kind=kind, cx=rng.randint(-20, w + 100), cy=rng.randint(-20, h + 20), sx=size,
                 sy=rng.randint(6, 80), val=rng.randint(20, 235), val2=rng.randint(20, 235),
                 disp=rng.randint(0, 3), t=[rng.randint(-40, 40) for _ in range(4)])
        o["disp"] += plane_disp(o["cy"] + o["sy"] // 2, h)
        objs.append(o)
    objs.sort(key=lambda o: o["disp"])  # stable: far first
    return objs


def render(w, h, objs, stream=0, right=False, pan=(0, 0)):
    """Render the left (or right) view of a scene as uint8 (h, w).  pan = integer (dx, dy) camera pan."""
    px, py = pan
    rows = np.arange(h) + py
    xoff = np.array([px + ((plane_disp(y, h) - 2) if right else 0) for y in rows])
    img = 110.0 + _value_noise(w, h, SEED0 ^ (stream << 32), xoff)
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    for o in objs:
        cx = o["cx"] - px - (o["disp"] if right else 0)
        cy = o["cy"] - py
        sx, sy = o["sx"], o["sy"]
        x0, x1 = max(cx, 0), min(cx + sx, w)
        y0, y1 = max(cy, 0), min(cy + sy, h)
        if x0 >= x1 or y0 >= y1:
            continue
        k = o["kind"]
        if k == 0:
            img[y0:y1, x0:x1] = o["val"]
        else:
            yy, xx = np.mgrid[y0:y1, x0:x1]
            if k == 1:
                r = sx / 2.0
                m = (xx - (cx + r)) ** 2 + (yy - (cy + r)) ** 2 <= r * r
                m &= yy < cy + sx
            elif k == 2:
                t = o["t"]
                ax, ay = cx, cy
                bx, by = cx + sx, cy + (t[0] % (sy + 1))
                qx, qy = cx + (t[1] % (sx + 1)), cy + sy
                def side(x1_, y1_, x2_, y2_):
                    return (xx - x1_) * (y2_ - y1_) - (yy - y1_) * (x2_ - x1_)
                s1, s2, s3 = side(ax, ay, bx, by), side(bx, by, qx, qy), side(qx, qy, ax, ay)
                m = ((s1 >= 0) & (s2 >= 0) & (s3 >= 0)) | ((s1 <= 0) & (s2 <= 0) & (s3 <= 0))
            else:
                m = None
                chk = (((xx - cx) // 8) + ((yy - cy) // 8)) & 1
                img[y0:y1, x0:x1] = np.where(chk == 0, o["val"], o["val2"]).astype(np.uint8)
            if m is not None:
                sub = img[y0:y1, x0:x1]
                sub[m] = o["val"]
    # independent +-2 sensor noise per view / pan
    yy, xx = np.mgrid[0:h, 0:w]
    k = (yy.astype(np.uint64) * np.uint64(8191) + xx.astype(np.uint64)
         + np.uint64((SEED0 * 977 + stream * 131 + (7 if right else 3) + px * 17 + py * 29) & _M64))
    nz = (_hash_u64(k) >> np.uint64(33)).astype(np.int64) % 5 - 2
    return np.clip(img.astype(np.int64) + nz, 0, 255).astype(np.uint8)


def stereo_pair(w, h, stream=0, frame=0):
    """(left, right) uint8 images of frame `frame` of stream `stream`."""
    objs = make_scene(w, h, stream)
    rng = SplitMix64(SEED0 ^ (stream << 32) ^ 0xABCDEF)
    pan = (0, 0)
    for _ in range(frame):
        pan = (pan[0] + rng.randint(0, 8), pan[1] + rng.randint(0, 4))
    return render(w, h, objs, stream, False, pan), render(w, h, objs, stream, True, pan)


def _ring_job(args):
    w, h, stream, frames = args
    return [stereo_pair(w, h, stream, f) for f in frames]


def stereo_ring(w, h, streams, frames, workers=None):
    """Frames `frames` of every stream in `streams`, rendered by a pool of worker processes (a 1280x720 pair takes
    ~1-2 s of numpy).  Returns (left, right) uint8 arrays of shape [len(frames), len(streams), h, w].
    Workers are spawned (not forked): safe after the HIP runtime has been initialised in the calling process."""
    import concurrent.futures as cf
    import multiprocessing as mp
    import os
    streams, frames = list(streams), list(frames)
    if workers is None:
        workers = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(workers, len(streams)))
    jobs = [(w, h, s, frames) for s in streams]
    if workers == 1:
        res = [_ring_job(j) for j in jobs]
    else:
        with cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(_ring_job, jobs))
    left = np.stack([np.stack([res[i][f][0] for i in range(len(streams))]) for f in range(len(frames))])
    right = np.stack([np.stack([res[i][f][1] for i in range(len(streams))]) for f in range(len(frames))])
    return left, right


def mono_frame(w, h, stream=0, frame=0):
    return stereo_pair(w, h, stream, frame)[0]


# ---- fisheye stereo rig (config C4: TUM-VI-like KannalaBrandt8 pair) ---------------------------------------------------
TUMVI_CAM1 = (190.97847715128717, 190.9733070521226, 254.93170605935475, 256.8974428996504,
              0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182)
TUMVI_CAM2 = (190.44236969414825, 190.4344384721956, 252.59949716835982, 254.91723064636983,
              0.0034003170790442797, 0.001766278153469831, -0.00266312569781606, 0.0003299517423931039)


def kb8_project_np(cam, X):
    """KannalaBrandt8::project in float64 numpy (scene construction only; the checked arithmetic is the oracle's)."""
    X = np.asarray(X, np.float64)
    th = np.arctan2(np.hypot(X[..., 0], X[..., 1]), X[..., 2])
    psi = np.arctan2(X[..., 1], X[..., 0])
    r = th + cam[4] * th ** 3 + cam[5] * th ** 5 + cam[6] * th ** 7 + cam[7] * th ** 9
    return np.stack([cam[0] * r * np.cos(psi) + cam[2], cam[1] * r * np.sin(psi) + cam[3]], -1)


def fisheye_stereo_scene(seed=0, n_left=900, n_right=850, mono_left=300, mono_right=280, n_levels=8):
    """Keypoints + descriptors of a synthetic fisheye stereo frame for ComputeStereoFishEyeMatches: 3-D points seen by
    both KB8 cameras (0.3-40 m, so the parallax gate fires on the far ones), pixel noise that grows with the octave
    (chi-square gates), wrong descriptor pairings (cheirality / reprojection gates), duplicated left keypoints
    (right keypoint claimed twice) and unmatched clutter.  Returns a dict of arrays; kps use KP_DTYPE records."""
    rng = np.random.default_rng(0xF15E + seed)
    kp_dtype = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                         ("octave", "<i4"), ("class_id", "<i4")])
    nQ, nT = n_left - mono_left, n_right - mono_right
    ang = np.deg2rad(rng.normal(0, 0.6, 3))
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    R12 = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
           np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t12 = np.array([0.1011, 0.0019, 0.0012]) + rng.normal(0, 2e-4, 3)
    sigma2 = (1.2 ** np.arange(n_levels)) ** 2
    kL, kR = np.zeros(n_left, kp_dtype), np.zeros(n_right, kp_dtype)
    dL = rng.integers(0, 256, (n_left, 32), dtype=np.uint8)
    dR = rng.integers(0, 256, (n_right, 32), dtype=np.uint8)
    for k in (kL, kR):
        k["x"], k["y"] = rng.uniform(20, 490, len(k)), rng.uniform(20, 490, len(k))
        k["octave"] = rng.integers(0, n_levels, len(k))
        k["size"], k["angle"], k["response"], k["class_id"] = 31.0, rng.uniform(0, 360, len(k)), 30.0, -1
    n_true = min(nQ, nT) * 3 // 4
    depth = np.exp(rng.uniform(np.log(0.3), np.log(40.0), n_true))
    th, ph = rng.uniform(0, 1.0, n_true), rng.uniform(0, 2 * np.pi, n_true)
    X1 = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], 1) * depth[:, None]
    X2 = (X1 - t12) @ R12            # x2 = R12^T (x1 - t12)
    ql = mono_left + rng.permutation(nQ)[:n_true]
    qr = mono_right + rng.permutation(nT)[:n_true]
    octv = rng.integers(0, n_levels, n_true)
    noise = rng.normal(0, 0.45, (n_true, 2, 2)) * (1.2 ** octv)[:, None, None] * rng.choice([1.0, 3.0], n_true, p=[0.8, 0.2])[:, None, None]
    uv1, uv2 = kb8_project_np(TUMVI_CAM1, X1) + noise[:, 0], kb8_project_np(TUMVI_CAM2, X2) + noise[:, 1]
    kL["x"][ql], kL["y"][ql], kL["octave"][ql] = uv1[:, 0], uv1[:, 1], octv
    kR["x"][qr], kR["y"][qr], kR["octave"][qr] = uv2[:, 0], uv2[:, 1], octv
    flips = rng.random((n_true, 32, 8)) < 0.03
    dR[qr] = dL[ql] ^ np.packbits(flips, axis=2).reshape(n_true, 32)
    # wrong pairings: descriptor says match, geometry says no
    wrong = rng.permutation(n_true)[: n_true // 5]
    kR["x"][qr[wrong]] += rng.normal(0, 60, len(wrong)).astype(np.float32)
    kR["y"][qr[wrong]] += rng.normal(0, 60, len(wrong)).astype(np.float32)
    # loose left gate (octave 7), tight right gate (octave 0) with a displaced right point: the -5 exit
    tight = rng.permutation(n_true)[: n_true // 10]
    kL["octave"][ql[tight]], kR["octave"][qr[tight]] = n_levels - 1, 0
    kR["x"][qr[tight]] += rng.normal(0, 2.5, len(tight)).astype(np.float32)
    kR["y"][qr[tight]] += rng.normal(0, 2.5, len(tight)).astype(np.float32)
    # duplicated left keypoints: two lapping rows, same pixel + descriptor -> the same right keypoint claimed twice
    free = np.setdiff1d(np.arange(mono_left, n_left), ql)
    dup = rng.permutation(n_true)[: min(len(free), 24)]
    kL[free[: len(dup)]] = kL[ql[dup]]
    dL[free[: len(dup)]] = dL[ql[dup]]
    return dict(kL=kL, dL=dL, kR=kR, dR=dR, mono_left=mono_left, mono_right=mono_right, cam1=np.array(TUMVI_CAM1, np.float32),
                cam2=np.array(TUMVI_CAM2, np.float32), R12=R12.astype(np.float32), t12=t12.astype(np.float32),
                level_sigma2=sigma2.astype(np.float32), true_left=ql, true_right=qr, true_depth=depth.astype(np.float32))


def rectify_maps(w, h, src_w=None, src_h=None, k1=-0.28, k2=0.07, p1=3e-4, p2=-2e-4, rot_deg=(0.4, -0.6, 0.25), seed=0):
    """Float rectification maps (map_x, map_y) of the kind cv::initUndistortRectifyMap(K, D, R, P, size, CV_32F) hands
    to System::TrackStereo (src/Settings.cc:557-572, src/System.cc:294-295): for every rectified pixel the raw-image
    position, radial-tangential distortion + a small rectifying rotation.  Test / bench input only; the product takes
    the maps as given."""
    src_w = src_w or w
    src_h = src_h or h
    rng = np.random.RandomState(seed)
    fx = 0.61 * src_w * (1 + 0.01 * rng.randn())
    fy = fx * (1 + 0.002 * rng.randn())
    cx, cy = 0.5 * src_w + 3.1 * rng.randn(), 0.5 * src_h + 2.3 * rng.randn()
    nfx, nfy, ncx, ncy = 0.56 * src_w * w / src_w, 0.56 * src_w * w / src_w, 0.5 * w + 1.7, 0.5 * h - 0.9
    ax, ay, az = np.deg2rad(rot_deg)
    Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
    Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
    Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    P = np.array([[nfx, 0, ncx], [0, nfy, ncy], [0, 0, 1.0]])
    iR = np.linalg.inv(P @ R)
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X = iR[0, 0] * u + iR[0, 1] * v + iR[0, 2]
    Y = iR[1, 0] * u + iR[1, 1] * v + iR[1, 2]
    W = iR[2, 0] * u + iR[2, 1] * v + iR[2, 2]
    x, y = X / W, Y / W
    r2 = x * x + y * y
    kr = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return (fx * xd + cx).astype(np.float32), (fy * yd + cy).astype(np.float32)


def make_vocabulary(k=10, L=4, seed=0, early_leaf_prob=0.03, stop_prob=0.02):
    """Synthetic DBoW2-style vocabulary tree in the node order of TemplatedVocabulary::HKmeansStep / the ORBvoc.txt file
    (the k children of a node are created together, then each child is expanded): per-node columns parent, is_leaf,
    descriptor (32 B), weight.  Children are their parent's descriptor with random bit flips (fewer at deeper levels), so a
    descriptor close to a leaf descends to it; a few inner positions become early leaves and a few words are stopped
    (weight 0).  The real ORBvoc.txt (k = 10, L = 6) is not part of the reference tree."""
    rng = np.random.RandomState(seed)
    parent, leaf, desc, weight = [0], [0], [np.zeros(32, np.uint8)], [0.0]

    def flip(d, nbits):
        bits = np.unpackbits(d)
        idx = rng.choice(256, nbits, replace=False)
        bits[idx] ^= 1
        return np.packbits(bits)

    def expand(pid, level):
        first = len(parent)
        base = desc[pid] if pid else rng.randint(0, 256, 32).astype(np.uint8)
        for _ in range(k):
            parent.append(pid)
            is_leaf = level == L or (level >= 2 and rng.rand() < early_leaf_prob)
            leaf.append(int(is_leaf))
            desc.append(flip(base, max(96 >> (level - 1), 6)))
            weight.append(0.0 if (is_leaf and rng.rand() < stop_prob) else (float(rng.uniform(0.5, 9.0)) if is_leaf else 0.0))
        for c in range(first, first + k):
            if not leaf[c]:
                expand(c, level + 1)

    expand(0, 1)
    return (np.array(parent, np.int32), np.array(leaf, np.uint8), np.stack(desc).astype(np.uint8), np.array(weight, np.float64))


def vocabulary_features(voc_cols, n, seed=0, noise_bits=10):
    """n descriptors near random leaves of a make_vocabulary() tree (plus a few purely random ones)."""
    rng = np.random.RandomState(seed)
    parent, leaf, desc, weight = voc_cols
    leaves = np.flatnonzero(leaf)
    pick = desc[rng.choice(leaves, n)].copy()
    bits = np.unpackbits(pick, axis=1)
    for i in range(n):
        bits[i, rng.choice(256, rng.randint(0, noise_bits + 1), replace=False)] ^= 1
    out = np.packbits(bits, axis=1)
    rnd = rng.rand(n) < 0.05
    out[rnd] = rng.randint(0, 256, (int(rnd.sum()), 32))
    return out


def make_vocabulary_bfs(k=10, L=6, seed=0):
    """A full k^L-word tree (ORBvoc.txt has k = 10, L = 6: 1 111 111 nodes) generated level by level with numpy: node order
    is breadth first (parent[i] < i still holds, which is all the loaders need).  Same column layout as make_vocabulary."""
    rng = np.random.RandomState(seed)
    parent, leaf, desc, weight = [np.zeros(1, np.int32)], [np.zeros(1, np.uint8)], [np.zeros((1, 32), np.uint8)], [np.zeros(1)]
    prev_ids, prev_desc, nxt = np.zeros(1, np.int64), rng.randint(0, 256, (1, 32)).astype(np.uint8), 1
    for level in range(1, L + 1):
        n = len(prev_ids) * k
        par = np.repeat(prev_ids, k)
        base = np.repeat(prev_desc, k, axis=0)
        nb = max(96 >> (level - 1), 6)
        flips = np.zeros((n, 256), np.uint8)
        cols = rng.randint(0, 256, (n, nb))
        flips[np.arange(n)[:, None], cols] = 1
        d = base ^ np.packbits(flips, axis=1)
        parent.append(par.astype(np.int32))
        leaf.append(np.full(n, int(level == L), np.uint8))
        desc.append(d)
        weight.append(rng.uniform(0.5, 9.0, n) if level == L else np.zeros(n))
        prev_ids, prev_desc, nxt = np.arange(nxt, nxt + n), d, nxt + n
    return np.concatenate(parent), np.concatenate(leaf), np.concatenate(desc), np.concatenate(weight)
