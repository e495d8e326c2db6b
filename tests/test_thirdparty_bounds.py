"""Third-party BOUNDS (not bit pins) for the OpenCV rules that no library in this image restates bit for bit.

The fixed-point arithmetic of cv::resize / cv::GaussianBlur / cv::remap can only be pinned by real OpenCV
(tools/gen_golden_opencv.py).  What CAN be checked here against independent float implementations is everything a wrong
memory would break grossly -- the sampling geometry (half-pixel centres, which source rows / columns a destination pixel
blends), the kernel (7 taps, sigma 2), the border rule (reflect-101) and the rounding direction:

  * cv::resize(INTER_LINEAR)     vs torch.nn.functional.interpolate(mode="bilinear", align_corners=False)   |diff| <= 1
    (OpenCV's 11-bit coefficients and two truncating shifts stay within one grey level of the exact bilinear value)
  * cv::GaussianBlur(7x7, 2)     vs scipy.ndimage.correlate1d with the exact normalised Gaussian, mode="mirror"  |diff| <= 1.5,
    EQUAL to scipy's integer correlation with the taps 18,34,48,56,48,34,18 (+ 32768 >> 16); the taps are the 8-bit
    rounding of that kernel that sums to 256
  * cv::remap(INTER_LINEAR)      vs torch grid_sample(bilinear, zeros padding, align_corners=True) on maps quantised to
    1/32 px (OpenCV's 5 fractional bits)                                                                   |diff| <= 1
  * cv::fastAtan2                vs numpy.arctan2 (0.3 deg)   -- tests/test_oracle_kernels.py
Reference call sites: src/ORBextractor.cc:1122 (resize), :1075 (GaussianBlur), src/System.cc:294 (remap)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def frames():
    pin = np.load(os.path.join(G, "fast9_skimage.npz"))
    return [pin["img_g384"], pin["img_g400L"], pin["img_g160"]]


def test_resize_geometry_matches_torch_bilinear(oracle, frames):
    import torch
    import torch.nn.functional as F
    for im in frames:
        h, w = im.shape
        for dw, dh in ((int(round(w / 1.2)), int(round(h / 1.2))), (w // 2, h // 2), (int(w * 0.37), int(h * 0.61)), (w + 57, h + 31)):
            got = oracle.resize(im, dw, dh).astype(np.int32)
            t = torch.from_numpy(im.astype(np.float64))[None, None]
            ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
            d = np.abs(got - ref)
            assert d.max() <= 1.0 + 1e-9, (dw, dh, d.max())
            assert (d > 0.75).mean() < 0.02          # and it rounds to nearest, not down: large errors are rare
            # the two truncating shifts of the fixed-point formula bias the result by about -0.1 grey levels; a geometric
            # error (half a pixel, align-corners convention) would break the |diff| <= 1 bound above on these textured frames
            assert -0.2 < (got - ref).mean() < 0.2   # (exact 2x takes the (a+b+c+d+2)>>2 path: +0.12)


def test_blur_matches_exact_gaussian(oracle, frames):
    from scipy import ndimage
    x = np.arange(-3, 4, dtype=np.float64)
    k = np.exp(-x * x / 8.0)
    k /= k.sum()
    taps = np.array([18, 34, 48, 56, 48, 34, 18])
    assert taps.sum() == 256 and np.abs(taps - 256 * k).max() < 0.85       # the 8.8 kernel of sigma = 2 (nearest rounding sums to 257: 48.8 -> 48, 55.3 -> 56 keep the sum at 256)
    for im in frames:
        ref = ndimage.correlate1d(ndimage.correlate1d(im.astype(np.float64), k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
        got = oracle.blur(im).astype(np.float64)
        assert np.abs(got - ref).max() <= 1.5        # taps differ from the exact kernel by up to 0.82 / 256
        assert abs((got - ref).mean()) < 0.1
        # the border rows / columns are where a wrong border rule (reflect vs reflect-101 vs replicate) would show
        edge = np.ones(im.shape, bool)
        edge[3:-3, 3:-3] = False
        assert np.abs(got - ref)[edge].max() <= 1.5
        fixed = ndimage.correlate1d(ndimage.correlate1d(im.astype(np.int64), taps, axis=0, mode="mirror"), taps, axis=1, mode="mirror")
        assert np.array_equal(got.astype(np.int64), (fixed + 32768) >> 16)       # == the integer definition, one rounding


def test_remap_matches_torch_grid_sample(oracle):
    import torch
    import torch.nn.functional as F
    rc = np.load(os.path.join(G, "rectify_clahe.npz"))
    img, mx, my = rc["img"], rc["map_x"], rc["map_y"]
    h, w = img.shape
    ok = np.isfinite(mx) & np.isfinite(my)
    qx = np.where(ok, np.rint(np.where(ok, mx, 0) * 32.0) / 32.0, -10.0)      # OpenCV: 5 fractional bits, cvRound
    qy = np.where(ok, np.rint(np.where(ok, my, 0) * 32.0) / 32.0, -10.0)
    gx = 2.0 * qx / (w - 1) - 1.0
    gy = 2.0 * qy / (h - 1) - 1.0
    grid = torch.from_numpy(np.stack([gx, gy], -1).astype(np.float64))[None]
    ref = F.grid_sample(torch.from_numpy(img.astype(np.float64))[None, None], grid, mode="bilinear", padding_mode="zeros",
                        align_corners=True)[0, 0].numpy()
    got = oracle.remap(img, mx, my).astype(np.float64)
    inside = ok & (qx >= 0) & (qx <= w - 1) & (qy >= 0) & (qy <= h - 1)
    assert inside.mean() > 0.5
    assert np.abs(got - ref)[inside].max() <= 1.0 + 1e-9


def test_gray_conversion_matches_pillow(oracle):
    """cv::cvtColor(RGB2GRAY) (14-bit fixed point 4899 / 9617 / 1868, src/Tracking.cc:1394-1412) vs Pillow's ITU-R 601
    conversion (16-bit fixed point 19595 / 38470 / 7471): the same luma weights to 4 decimals, so the two agree within one
    grey level, and exactly on greys."""
    from PIL import Image
    rng = np.random.default_rng(9)
    rgb = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(rgb, "RGB").convert("L")).astype(np.int32)
    got = oracle.cvt_gray(rgb, rgb=True).astype(np.int32)
    assert np.abs(got - ref).max() <= 1 and (got != ref).mean() < 0.2
    bgr = oracle.cvt_gray(rgb[:, :, ::-1].copy(), rgb=False).astype(np.int32)
    assert np.array_equal(bgr, got)
    grey = np.repeat(rng.integers(0, 256, (32, 32, 1), dtype=np.uint8), 3, axis=2)
    assert np.array_equal(oracle.cvt_gray(grey, rgb=True), grey[:, :, 0])
