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


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    extract_case("extract_160x120_L3.npz", 160, 120, 300, 3, 101, (0, 0))
    extract_case("extract_384x288_L8.npz", 384, 288, 500, 8, 102, (0, 0))
    extract_case("extract_384x288_L8_lap.npz", 384, 288, 500, 8, 102, (100, 250))
    stereo_case("stereo_400x300.npz", 400, 300, 600, 103)
