"""Randomised GPU-vs-oracle parity sweep (extraction + stereo association) over image sizes, extractor parameters
and image statistics.  Marked `slow`: run explicitly with  pytest tests/test_fuzz_parity.py -m "gpu and slow"
(ORBX_FUZZ_CASES sets the number of cases, default 60).  Every case is seeded, so a failure is reproducible."""
import os

import numpy as np
import pytest

import orb_slam3_fast_amd as orbx
from orb_slam3_fast_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def _kb(k):
    return np.ascontiguousarray(k).view(np.uint8).reshape(len(k), 28)


def make_image(rng, w, h, stream):
    L, R = synth.stereo_pair(w, h, stream)
    mode = rng.integers(0, 6)
    out = []
    for im in (L, R):
        im = im.copy()
        if mode == 1:      # low contrast everywhere: many min-threshold cells
            im = (im // 6 + 100).astype(np.uint8)
        elif mode == 2:    # half white noise
            nz = rng.integers(0, 256, im.shape, dtype=np.uint8)
            im[:, : w // 2] = nz[:, : w // 2]
        elif mode == 3:    # large flat regions
            im[h // 4: 3 * h // 4, w // 5: w // 2] = 140
        elif mode == 4:    # strong gradient + texture
            g = np.linspace(0, 120, w, dtype=np.float32)[None, :]
            im = np.clip(im.astype(np.float32) * 0.5 + g, 0, 255).astype(np.uint8)
        out.append(im)
    return out


def test_fuzz_extract_and_stereo(oracle):
    assert orbx.device_count() > 0
    ncases = int(os.environ.get("ORBX_FUZZ_CASES", "60"))
    seed0 = int(os.environ.get("ORBX_FUZZ_SEED", "12345"))
    fails = []
    for case in range(ncases):
        rng = np.random.default_rng(seed0 + case)
        nl = int(rng.integers(2, 9))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5]))
        smallest = 32 + 35 + 6
        while int(np.ceil(smallest * sf ** (nl - 1))) + 8 > 600:
            nl -= 1
        minw = int(np.ceil(smallest * sf ** (nl - 1))) + 8
        w = int(rng.integers(max(minw, 200), 1000))
        h = int(rng.integers(max(minw, 200), 760))
        if max(w, h) / min(w, h) > 2.4:
            h = max(h, int(w / 2.4) + 1)
            if h - 32 < 35 * sf ** (nl - 1):
                continue
        nf = int(rng.integers(150, 3000))
        ini = int(rng.integers(10, 40))
        mn = int(rng.integers(3, ini + 1))
        lap = (0, 0) if rng.random() < 0.6 else (int(rng.integers(0, w // 2)), int(rng.integers(w // 2, w + 50)))
        L, R = make_image(rng, w, h, 500 + case)
        try:
            exL = orbx.ORBextractor(nf, sf, nl, ini, mn, max_width=w, max_height=h)
            exR = orbx.ORBextractor(nf, sf, nl, ini, mn, max_width=w, max_height=h)
        except orbx.OrbxError as e:
            if e.code == orbx.E_UNSUPPORTED:
                continue
            raise
        oL, oR = oracle.OracleExtractor(nf, sf, nl, ini, mn), oracle.OracleExtractor(nf, sf, nl, ini, mn)
        mL, kL, dL = exL(L, lap)
        mR, kR, dR = exR(R, (0, 0))
        omL, okL, odL = oL.extract(L, lap)
        omR, okR, odR = oR.extract(R, (0, 0))
        ok = (mL == omL and mR == omR and np.array_equal(_kb(kL), _kb(okL)) and np.array_equal(dL, odL)
              and np.array_equal(_kb(kR), _kb(okR)) and np.array_equal(dR, odR))
        if ok and lap == (0, 0) and len(kL) and len(kR):
            bf, b = 0.12 * 500.0, 0.12
            u, dep = orbx.ComputeStereoMatches(exL, exR, bf, b)
            ou, od = oracle.stereo_match(oL, oR, okL, odL, okR, odR, bf, b)
            n = len(kL)
            ok = u[0, :n].tobytes() == ou.tobytes() and dep[0, :n].tobytes() == od.tobytes()
        if not ok:
            fails.append((seed0 + case, w, h, nf, sf, nl, ini, mn, lap))
        exL.close()
        exR.close()
    assert not fails, "mismatching cases (seed, w, h, nfeatures, scale, levels, ini, min, lap): %r" % fails
