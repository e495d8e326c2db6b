#!/usr/bin/env python3
"""Per-cell statistics of k_detect's work on the benchmark frames (run on the GPU box):
pre-NMS corners per FAST cell at iniThFAST and compass-test survivors per cell, per level.
    python tools/cell_stats.py [streams]
"""
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import orb_slam3_fast_amd as orbx  # noqa: E402
from orb_slam3_fast_amd import synth  # noqa: E402


def compass(img, t):
    """survivor mask of k_detect's stage 1: two cyclically adjacent compass points of one polarity."""
    c = img.astype(np.int16)
    h, w = c.shape
    pad = np.pad(c, 3, mode="edge")
    v0 = pad[6:6 + h, 3:3 + w]    # (0, +3)
    v4 = pad[3:3 + h, 6:6 + w]    # (+3, 0)
    v8 = pad[0:h, 3:3 + w]        # (0, -3)
    v12 = pad[3:3 + h, 0:w]       # (-3, 0)
    hi = np.minimum(np.maximum(v0, v8), np.maximum(v4, v12)) > c + t
    lo = np.maximum(np.minimum(v0, v8), np.minimum(v4, v12)) < c - t
    return hi | lo


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
        (-2, 2), (-1, 3)]


def even_ring(img, t):
    """the strongest necessary test on the EVEN ring positions only (round-5 experiment, -DORBX_EVEN_FILTER=1): a 9-arc covers at
    least four cyclically consecutive even positions, all > c + t or all < c - t."""
    c = img.astype(np.int16)
    h, w = c.shape
    pad = np.pad(c, 3, mode="edge")
    e = [pad[3 + dy:3 + dy + h, 3 + dx:3 + dx + w] for (dx, dy) in RING[::2]]
    hi = np.zeros((h, w), bool)
    lo = np.zeros((h, w), bool)
    for q in range(8):
        run = [e[(q + i) & 7] for i in range(4)]
        hi |= np.minimum(np.minimum(run[0], run[1]), np.minimum(run[2], run[3])) > c + t
        lo |= np.maximum(np.maximum(run[0], run[1]), np.maximum(run[2], run[3])) < c - t
    return hi | lo


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    w, h = 1280, 720
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    ex.debug_score_map(True)
    corners, surv, surv2 = [], [], []
    for s in range(streams):
        img = synth.stereo_pair(w, h, s)[0]
        ex(img, (0, 0))
        for l in range(8):
            sc = ex.debug_score_level(l)
            lv = ex.image_pyramid(l)
            H, W = lv.shape
            width, height = W - 32, H - 32
            nc, nr = int(width / 35.0), int(height / 35.0)
            wc, hc = int(np.ceil(width / nc)), int(np.ceil(height / nr))
            cm = compass(lv, 20)
            em = cm & even_ring(lv, 20)
            for i in range(nr):
                for j in range(nc):
                    y0, x0 = 19 + i * hc, 19 + j * wc
                    y1, x1 = min(y0 + hc, H - 19), min(x0 + wc, W - 19)
                    if y1 <= y0 or x1 <= x0:
                        continue
                    corners.append(int((sc[y0:y1, x0:x1] > 0).sum()))
                    surv.append(int(cm[y0:y1, x0:x1].sum()))
                    surv2.append(int(em[y0:y1, x0:x1].sum()))
    corners, surv, surv2 = np.array(corners), np.array(surv), np.array(surv2)
    q = [10, 25, 50, 75, 90, 95, 99, 100]
    print("cells", len(corners))
    print("pre-NMS corners / cell: mean %.1f  percentiles %s = %s" % (corners.mean(), q, np.percentile(corners, q).astype(int).tolist()))
    print("compass survivors / cell: mean %.1f  percentiles %s = %s" % (surv.mean(), q, np.percentile(surv, q).astype(int).tolist()))
    print("contrast passes of 128: mean %.2f; cells with > 256 corners %.1f %%, > 448 %.2f %%; survivors > 448: %.1f %%, > 640: %.1f %%" % (
        np.ceil(surv / 128.0).mean(), 100.0 * (corners > 256).mean(), 100.0 * (corners > 448).mean(),
        100.0 * (surv > 448).mean(), 100.0 * (surv > 640).mean()))


    print("survivors of compass AND even-ring (4 consecutive even positions) / cell: mean %.1f = %.0f %% of the compass survivors  "
          "percentiles %s = %s" % (surv2.mean(), 100.0 * surv2.sum() / max(surv.sum(), 1), q, np.percentile(surv2, q).astype(int).tolist()))
    print("contrast passes of 128 after the even-ring filter: mean %.2f (+ %.2f filter passes over the compass survivors)" % (
        np.ceil(surv2 / 128.0).mean(), np.ceil(surv / 128.0).mean()))
    h_ = np.histogram(surv2, bins=[0, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 448, 10000])[0]
    print("histogram of survivors after the filter (bins 0,32,..,256,320,384,448+): %s" % h_.tolist())
    h0 = np.histogram(surv, bins=[0, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 448, 10000])[0]
    print("histogram of compass survivors          (same bins):                     %s" % h0.tolist())


if __name__ == "__main__":
    main()
