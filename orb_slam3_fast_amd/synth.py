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


def mono_frame(w, h, stream=0, frame=0):
    return stereo_pair(w, h, stream, frame)[0]
