"""A second, literal restatement of Frame::ComputeStereoMatches (src/Frame.cc:921-1084) in Python / numpy float32,
transcribed statement by statement, checked bit for bit against the C++ oracle on seeded synthetic stereo pairs."""
import math

import numpy as np
import pytest

from orb_slam3_fast_amd import synth

f32 = np.float32
TH_HIGH, TH_LOW = 100, 50


def c_round(v):  # C round(): halves away from zero, on a float32 value
    v = float(v)
    return f32(math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5))


def hamming(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def compute_stereo_matches_py(pyrL, pyrR, kL, dL, kR, dR, scale, inv_scale, mbf, mb):
    N = len(kL)
    uRight, depth = np.full(N, -1.0, f32), np.full(N, -1.0, f32)
    thOrbDist = (TH_HIGH + TH_LOW) // 2
    nRows = pyrL[0].shape[0]
    rows = [[] for _ in range(nRows)]
    for iR in range(len(kR)):
        kpY = f32(kR["y"][iR])
        if kpY == 0.0 and f32(kR["x"][iR]) == 0.0:
            continue
        r = f32(2.0) * scale[kR["octave"][iR]]
        maxr, minr = int(math.ceil(float(kpY + r))), int(math.floor(float(kpY - r)))
        for yi in range(minr, maxr + 1):
            rows[yi].append(iR)
    minZ, minD = f32(mb), f32(0)
    maxD = f32(mbf) / minZ
    vDistIdx = []
    for iL in range(N):
        levelL, vL, uL = int(kL["octave"][iL]), f32(kL["y"][iL]), f32(kL["x"][iL])
        cands = rows[int(vL)]
        if not cands:
            continue
        minU, maxU = uL - maxD, uL - minD
        if maxU < 0:
            continue
        bestDist, bestIdxR = TH_HIGH, 0
        for iR in cands:
            if kR["octave"][iR] < levelL - 1 or kR["octave"][iR] > levelL + 1:
                continue
            uR = f32(kR["x"][iR])
            if minU <= uR <= maxU:
                dist = hamming(dL[iL], dR[iR])
                if dist < bestDist:
                    bestDist, bestIdxR = dist, iR
        if bestDist < thOrbDist:
            uR0 = f32(kR["x"][bestIdxR])
            sfac = inv_scale[levelL]
            scaleduL, scaledvL, scaleduR0 = c_round(uL * sfac), c_round(vL * sfac), c_round(uR0 * sfac)
            w, L = 5, 5
            imL, imR = pyrL[levelL], pyrR[levelL]
            IL = imL[int(scaledvL) - w:int(scaledvL) + w + 1, int(scaleduL) - w:int(scaleduL) + w + 1].astype(np.int64)
            best, bestinc = 2 ** 31 - 1, 0
            vDists = [f32(0)] * (2 * L + 1)
            iniu, endu = scaleduR0 + f32(L) - f32(w), scaleduR0 + f32(L) + f32(w) + f32(1)
            if iniu < 0 or endu >= imR.shape[1]:
                continue
            for inc in range(-L, L + 1):
                c0 = int(scaleduR0) + inc - w
                IR = imR[int(scaledvL) - w:int(scaledvL) + w + 1, c0:c0 + 2 * w + 1].astype(np.int64)
                dist = f32(np.abs(IL - IR).sum())
                if dist < f32(best):
                    best, bestinc = int(dist), inc
                vDists[L + inc] = dist
            if bestinc == -L or bestinc == L:
                continue
            d1, d2, d3 = vDists[L + bestinc - 1], vDists[L + bestinc], vDists[L + bestinc + 1]
            with np.errstate(divide="ignore", invalid="ignore"):
                deltaR = (d1 - d3) / (f32(2.0) * (d1 + d3 - f32(2.0) * d2))
            if deltaR < -1 or deltaR > 1:  # (0/0 = NaN fails both comparisons, as in C, and is rejected by the disparity test)
                continue
            bestuR = scale[levelL] * (scaleduR0 + f32(bestinc) + deltaR)
            disparity = uL - bestuR
            if disparity >= minD and disparity < maxD:
                if disparity <= 0:
                    disparity = f32(0.01)
                    bestuR = f32(float(uL) - 0.01)
                depth[iL] = f32(mbf) / disparity
                uRight[iL] = bestuR
                vDistIdx.append((best, iL))
    if vDistIdx:
        vDistIdx.sort()
        median = f32(vDistIdx[len(vDistIdx) // 2][0])
        thDist = f32(1.5) * f32(1.4) * median
        for d, i in reversed(vDistIdx):
            if f32(d) < thDist:
                break
            uRight[i] = -1
            depth[i] = -1
    return uRight, depth


@pytest.mark.parametrize("w,h,nf,stream", [(400, 300, 500, 201), (512, 384, 700, 202), (640, 480, 400, 203)])
def test_python_restatement_of_compute_stereo_matches_matches_oracle(oracle, w, h, nf, stream):
    L, R = synth.stereo_pair(w, h, stream)
    eL, eR = oracle.OracleExtractor(nf), oracle.OracleExtractor(nf)
    _, kL, dL = eL.extract(L)
    _, kR, dR = eR.extract(R)
    t = eL.tables()
    bf, b = f32(0.12) * f32(532.03), f32(0.12)
    pyrL, pyrR = [eL.level(l) for l in range(8)], [eR.level(l) for l in range(8)]
    eu, ed = compute_stereo_matches_py(pyrL, pyrR, kL, dL, kR, dR, t["scale"], t["inv_scale"], bf, b)
    ou, od = oracle.stereo_match(eL, eR, kL, dL, kR, dR, bf, b)
    assert (ou >= 0).sum() > 50
    assert eu.tobytes() == ou.tobytes() and ed.tobytes() == od.tobytes()
