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


def main():
    streams = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    w, h = 1280, 720
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    ex.debug_score_map(True)
    corners, surv = [], []
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
            for i in range(nr):
                for j in range(nc):
                    y0, x0 = 19 + i * hc, 19 + j * wc
                    y1, x1 = min(y0 + hc, H - 19), min(x0 + wc, W - 19)
                    if y1 <= y0 or x1 <= x0:
                        continue
                    corners.append(int((sc[y0:y1, x0:x1] > 0).sum()))
                    surv.append(int(cm[y0:y1, x0:x1].sum()))
    corners, surv = np.array(corners), np.array(surv)
    q = [10, 25, 50, 75, 90, 95, 99, 100]
    print("cells", len(corners))
    print("pre-NMS corners / cell: mean %.1f  percentiles %s = %s" % (corners.mean(), q, np.percentile(corners, q).astype(int).tolist()))
    print("compass survivors / cell: mean %.1f  percentiles %s = %s" % (surv.mean(), q, np.percentile(surv, q).astype(int).tolist()))
    print("contrast passes of 128: mean %.2f; cells with > 256 corners %.1f %%, > 448 %.2f %%; survivors > 448: %.1f %%, > 640: %.1f %%" % (
        np.ceil(surv / 128.0).mean(), 100.0 * (corners > 256).mean(), 100.0 * (corners > 448).mean(),
        100.0 * (surv > 448).mean(), 100.0 * (surv > 640).mean()))


if __name__ == "__main__":
    main()
