"""A second, literal restatement of ORBextractor::DistributeOctTree (src/ORBextractor.cc:490-757) in pure Python --
std::list emulated by a Python list with push_front / erase, DivideNode and the loops transcribed statement by
statement -- checked against the C++ oracle on random candidate sets.  The only shared piece is libstdc++'s std::sort
(called through the oracle library) because the tie order of an unstable sort cannot be restated independently."""
import ctypes as C
import math

import numpy as np
import pytest

f32 = np.float32


class Node:
    __slots__ = ("UL", "UR", "BL", "BR", "keys", "no_more")

    def __init__(self):
        self.UL = self.UR = self.BL = self.BR = (0, 0)
        self.keys = []
        self.no_more = False

    def divide(self):
        halfX = int(math.ceil(f32(self.UR[0] - self.UL[0]) / f32(2)))
        halfY = int(math.ceil(f32(self.BR[1] - self.UL[1]) / f32(2)))
        n1, n2, n3, n4 = Node(), Node(), Node(), Node()
        n1.UL = self.UL
        n1.UR = (self.UL[0] + halfX, self.UL[1])
        n1.BL = (self.UL[0], self.UL[1] + halfY)
        n1.BR = (self.UL[0] + halfX, self.UL[1] + halfY)
        n2.UL, n2.UR, n2.BL, n2.BR = n1.UR, self.UR, n1.BR, (self.UR[0], self.UL[1] + halfY)
        n3.UL, n3.UR, n3.BL, n3.BR = n1.BL, n1.BR, self.BL, (n1.BR[0], self.BL[1])
        n4.UL, n4.UR, n4.BL, n4.BR = n3.UR, n2.BR, n3.BR, self.BR
        for kp in self.keys:
            if kp[0] < n1.UR[0]:
                (n1 if kp[1] < n1.BR[1] else n3).keys.append(kp)
            elif kp[1] < n1.BR[1]:
                n2.keys.append(kp)
            else:
                n4.keys.append(kp)
        for n in (n1, n2, n3, n4):
            if len(n.keys) == 1:
                n.no_more = True
        return n1, n2, n3, n4


def std_sort_nodes(oracle, pairs):
    """std::sort(v.begin(), v.end(), compareNodes) on (size, node) pairs via libstdc++ (through the oracle library)."""
    v = np.array([(np.uint64(p[0]) << np.uint64(28)) | (np.uint64(p[1].UL[0]) << np.uint64(16)) | np.uint64(i)
                  for i, p in enumerate(pairs)], np.uint64)
    oracle.lib().oro_std_sort_keys(v.ctypes.data_as(C.c_void_p), len(v))
    return [pairs[int(x) & 0xFFFF] for x in v]


def distribute_octtree_py(oracle, keys, minX, maxX, minY, maxY, N):
    """keys: list of (x, y, response) with float coordinates relative to the window (integral values)."""
    nIni = int(math.floor(float(f32(maxX - minX) / f32(maxY - minY)) + 0.5))  # C round(): halves away from zero
    hX = f32(maxX - minX) / f32(nIni)
    lNodes = []
    ini = []
    for i in range(nIni):
        ni = Node()
        ni.UL = (int(hX * f32(i)), 0)
        ni.UR = (int(hX * f32(i + 1)), 0)
        ni.BL = (ni.UL[0], maxY - minY)
        ni.BR = (ni.UR[0], maxY - minY)
        lNodes.append(ni)
        ini.append(ni)
    for kp in keys:
        ini[int(f32(kp[0]) / hX)].keys.append(kp)
    i = 0
    while i < len(lNodes):
        if len(lNodes[i].keys) == 1:
            lNodes[i].no_more = True
            i += 1
        elif not lNodes[i].keys:
            del lNodes[i]
        else:
            i += 1
    finish = False
    size_and_node = []
    while not finish:
        prev_size = len(lNodes)
        i = 0
        n_to_expand = 0
        size_and_node = []
        while i < len(lNodes):
            lit = lNodes[i]
            if lit.no_more:
                i += 1
                continue
            for n in lit.divide():
                if len(n.keys) > 0:
                    lNodes.insert(0, n)  # push_front
                    i += 1
                    if len(n.keys) > 1:
                        n_to_expand += 1
                        size_and_node.append((len(n.keys), n))
            del lNodes[i]  # lit = lNodes.erase(lit)
        if len(lNodes) >= N or len(lNodes) == prev_size:
            finish = True
        elif len(lNodes) + n_to_expand * 3 > N:
            while not finish:
                prev_size = len(lNodes)
                prev = std_sort_nodes(oracle, size_and_node)
                size_and_node = []
                for j in range(len(prev) - 1, -1, -1):
                    node = prev[j][1]
                    for n in node.divide():
                        if len(n.keys) > 0:
                            lNodes.insert(0, n)
                            if len(n.keys) > 1:
                                size_and_node.append((len(n.keys), n))
                    lNodes.remove(node)  # erase(...->lit); Node has identity semantics
                    if len(lNodes) >= N:
                        break
                if len(lNodes) >= N or len(lNodes) == prev_size:
                    finish = True
    out = []
    for node in lNodes:
        best = node.keys[0]
        for kp in node.keys[1:]:
            if kp[2] > best[2]:
                best = kp
        out.append(best)
    return out


def _cands(rng, W, H, n, clustered):
    if clustered:
        cx, cy = rng.uniform(0, W, 6), rng.uniform(0, H, 6)
        which = rng.integers(0, 6, n)
        x = np.clip(cx[which] + rng.normal(0, W / 15, n), 0, W - 1)
        y = np.clip(cy[which] + rng.normal(0, H / 15, n), 0, H - 1)
    else:
        x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
    pts = np.unique(np.stack([np.floor(x), np.floor(y)], 1).astype(np.int64), axis=0)  # FAST never yields a pixel twice
    rng.shuffle(pts)
    resp = rng.integers(7, 60, len(pts))  # few distinct responses: "first maximum wins" matters
    return pts, resp


@pytest.mark.parametrize("W,H,N,n,clustered,seed", [
    (1248, 688, 326, 6000, False, 1), (1248, 688, 326, 6000, True, 2), (608, 448, 217, 2500, False, 3),
    (720, 448, 217, 900, True, 4), (325, 169, 91, 400, False, 5), (325, 169, 91, 1500, True, 6),
    (1008, 568, 271, 150, False, 7), (480, 480, 100, 3000, True, 8), (1248, 688, 30, 2000, False, 9),
    (300, 120, 60, 700, True, 10),
])
def test_python_restatement_of_distribute_octtree_matches_oracle(oracle, W, H, N, n, clustered, seed):
    rng = np.random.default_rng(seed)
    pts, resp = _cands(rng, W, H, n, clustered)
    keys = [(float(p[0]), float(p[1]), float(r)) for p, r in zip(pts, resp)]
    exp = distribute_octtree_py(oracle, keys, 16, 16 + W, 16, 16 + H, N)
    cand = np.zeros(len(pts), oracle.KP_DTYPE)
    cand["x"], cand["y"], cand["response"], cand["size"], cand["angle"], cand["class_id"] = pts[:, 0], pts[:, 1], resp, 7, -1, -1
    ex = oracle.OracleExtractor(1000)
    got = ex.distribute(cand, 16, 16 + W, 16, 16 + H, N)
    assert len(got) == len(exp) and len(got) >= min(N, 1)
    assert [(float(k["x"]), float(k["y"]), float(k["response"])) for k in got] == exp
