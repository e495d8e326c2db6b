#!/opt/conda/bin/python3.9
"""Third-party pin of the FAST-9-16 corner criterion, the FAST score and the intensity-centroid orientation.

The reference's arithmetic for these lives in OpenCV (`cv::FAST` src/ORBextractor.cc:810-826, `IC_Angle` :75-99), which
is not installed here.  scikit-image 0.18.3 IS (only under /opt/conda/bin/python3.9) and holds independent
implementations of the same published definitions:

  * `skimage.feature.corner_fast(img, n=9, threshold=t) > 0`  -- the segment test: >= 9 contiguous ring pixels all
    > I + t or all < I - t (Rosten & Drummond), the criterion of cv::FAST TYPE_9_16;
  * the same call swept over t = 1..254 -- the FAST score of OpenCV's cornerScore<16> is the largest threshold at
    which the pixel is still a corner, so  score(p) = #{t : p is a corner at t}  (0 = never);
  * `corner_orientations(img, corners, OFAST_MASK)` -- atan2(m01, m10) over the 31x31 circular patch (749 px, the
    same u_max table as ORBextractor.cc:456-468), the exact angle that cv::fastAtan2 approximates to 0.3 deg;
  * `skimage.feature.orb_cy._orb_loop(image, keypoints, orientations)` -- the steered-BRIEF sampling of skimage's own ORB:
    256 tests  I(p + R(angle) a_j) < I(p + R(angle) b_j)  on its copy of the 31x31 pattern, row offset
    round(sin x + cos y), column offset round(cos x - sin y) in double: an independent implementation of
    computeOrbDescriptor (ORBextractor.cc:102-147) -- pattern pairing, rotation sense, comparison direction, bit order.
    (Run on the raw frame: what is sampled does not matter for the sampling conventions.)

Run in THIS container only (the GPU box has no skimage):   /opt/conda/bin/python3.9 tools/pin_skimage.py
Writes tests/golden/fast9_skimage.npz (inputs + skimage's outputs, data only).  tests/test_pin_skimage.py checks the
oracle against it on the CPU and the HIP path against it on the GPU, neither through the other.
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.feature import corner_fast, corner_orientations  # noqa: E402
from skimage.feature.orb import OFAST_MASK  # noqa: E402
from skimage.feature.orb_cy import _orb_loop  # noqa: E402
import skimage  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def corner_mask(img, t):
    return corner_fast(img.astype(np.float64), n=9, threshold=float(t)) > 0


def score_map(img):
    """score(p) = number of thresholds 1..254 at which p passes the segment test = the largest such threshold."""
    s = np.zeros(img.shape, np.uint8)
    f = img.astype(np.float64)
    for t in range(1, 255):
        m = corner_fast(f, n=9, threshold=float(t)) > 0
        if not m.any():
            break
        assert not (m & (s != t - 1)).any(), "segment test is not monotone in t"
        s[m] = t
    return s


def orientations(img, mask, limit, rng):
    ys, xs = np.nonzero(mask[15:-15, 15:-15])
    ys, xs = ys + 15, xs + 15
    if len(ys) > limit:
        sel = np.sort(rng.choice(len(ys), limit, replace=False))
        ys, xs = ys[sel], xs[sel]
    # skimage reads the image as float (uint8 / 255): the moment RATIO, hence the angle, is unchanged
    ang = corner_orientations(img, np.stack([ys, xs], 1), OFAST_MASK)
    return np.stack([ys, xs], 1).astype(np.int32), ang.astype(np.float64)


def main():
    rng = np.random.default_rng(20220131)
    imgs = {
        "g384": np.load(os.path.join(G, "extract_384x288_L8.npz"))["image"],
        "g160": np.load(os.path.join(G, "extract_160x120_L3.npz"))["image"],
        "g400L": np.load(os.path.join(G, "stereo_400x300.npz"))["left"],
        "g400R": np.load(os.path.join(G, "stereo_400x300.npz"))["right"],
        # natural photographs (tools/gen_natural_fixture.py): corner density, ties and low-contrast regions of real texture
        "nat_cam": np.load(os.path.join(G, "natural_images.npz"))["camera"],
        "nat_moto": np.load(os.path.join(G, "natural_images.npz"))["moto_left"],
    }
    out = {"skimage_version": np.array(skimage.__version__), "names": np.array(sorted(imgs))}
    for name, im in sorted(imgs.items()):
        out["img_" + name] = im
        for t in (20, 7):
            m = corner_mask(im, t)
            out["mask%d_%s" % (t, name)] = np.packbits(m, axis=1)
            print(name, im.shape, "t=%d" % t, int(m.sum()), "corners")
        pos, ang = orientations(im, corner_mask(im, 20), 1500, rng)
        out["ori_pos_" + name] = pos
        out["ori_rad_" + name] = ang
        # descriptors at (up to) 600 of those corners that lie >= 20 px inside, for the angle the extractor would carry:
        # float32 degrees in [0, 360) (the radians handed to skimage are that float32 value times pi / 180 in double),
        # plus the exact quadrant angles on the first corners
        h_, w_ = im.shape
        ok = (pos[:, 0] >= 20) & (pos[:, 0] < h_ - 20) & (pos[:, 1] >= 20) & (pos[:, 1] < w_ - 20)
        dpos = pos[ok][:600]
        ddeg = np.float32(np.degrees(ang[ok][:600]) % 360.0)
        ddeg[:8] = np.float32([0, 90, 180, 270, 45, 135, 225, 315])[:len(ddeg[:8])]
        out["desc_pos_" + name] = dpos
        out["desc_deg_" + name] = ddeg
        out["desc_bits_" + name] = np.packbits(_orb_loop(im.astype(np.float64), dpos.astype(np.intp),
                                                         np.deg2rad(ddeg.astype(np.float64))).astype(bool), axis=1, bitorder="little")
    # full FAST-score maps (threshold sweep) on small inputs: crops of the synthetic frames, noise, flat rectangles
    # (equal neighbouring scores), extremes 0 / 255
    crops = []
    big = imgs["g384"]
    for _ in range(4):
        y, x = int(rng.integers(0, big.shape[0] - 64)), int(rng.integers(0, big.shape[1] - 64))
        crops.append(big[y:y + 64, x:x + 64].copy())
    crops.append(rng.integers(0, 256, (64, 64), dtype=np.uint8))
    crops.append((rng.integers(0, 2, (64, 64)) * 255).astype(np.uint8))
    rects = np.full((64, 64), 90, np.uint8)   # flat rectangles: runs of equal scores along their corners (NMS ties)
    for _ in range(14):
        y, x, hh, ww = (int(v) for v in rng.integers(2, 44, 4))
        rects[y:y + 3 + hh // 3, x:x + 3 + ww // 3] = int(rng.integers(0, 256))
    crops.append(rects)
    crops.append(np.clip(110 + 18 * rng.standard_normal((64, 64)), 0, 255).astype(np.uint8))
    crops = np.stack(crops)
    out["score_inputs"] = crops
    out["score_maps"] = np.stack([score_map(c) for c in crops])
    print("score maps:", [int((s > 0).sum()) for s in out["score_maps"]], "corners at t=1; max score", int(out["score_maps"].max()))
    np.savez_compressed(os.path.join(G, "fast9_skimage.npz"), **out)
    print("wrote tests/golden/fast9_skimage.npz", os.path.getsize(os.path.join(G, "fast9_skimage.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
