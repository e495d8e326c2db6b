"""Oracle tables vs the values SURVEY.md Appendix C derives from the reference
(src/ORBextractor.cc:408-469) and the pinned rBRIEF pattern."""
import ctypes
import hashlib
import os
import struct

import numpy as np
import pytest

SHA = "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
HERE = os.path.dirname(os.path.abspath(__file__))


def test_pattern_sha_and_skimage_copy(oracle):
    ptr = oracle.lib().oro_pattern()
    vals = np.frombuffer((ctypes.c_int8 * 1024).from_address(ptr), np.int8).astype(np.int32)
    assert hashlib.sha256(struct.pack("<1024i", *vals.tolist())).hexdigest() == SHA
    assert vals.min() == -13 and vals.max() == 12
    golden = np.loadtxt(os.path.join(HERE, "golden", "orb_descriptor_positions_skimage.txt")).astype(np.int32)
    assert np.array_equal(golden.reshape(-1), vals)


def test_umax_and_scale_tables(oracle):
    t = oracle.OracleExtractor(1000, 1.2, 8, 20, 7).tables()
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    s = np.float32(1.0)
    exp = [s]
    for _ in range(7):
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))
        exp.append(s)
    assert np.array_equal(t["scale"], np.array(exp, np.float32))
    assert np.array_equal(t["inv_scale"], (np.float32(1.0) / t["scale"]).astype(np.float32))
    assert np.array_equal(t["sigma2"], t["scale"] * t["scale"])


@pytest.mark.parametrize("n,exp", [
    (1000, [217, 181, 151, 126, 105, 87, 73, 60]),
    (1200, [261, 217, 181, 151, 126, 105, 87, 72]),
    (1500, [326, 271, 226, 189, 157, 131, 109, 91]),
    (5000, [1086, 905, 754, 628, 524, 436, 364, 303]),
])
def test_quotas(oracle, n, exp):
    assert oracle.OracleExtractor(n, 1.2, 8, 20, 7).tables()["nfeat"].tolist() == exp


@pytest.mark.parametrize("wh,levels", [
    ((640, 480), [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    ((752, 480), [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]),
    ((1280, 720), [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)]),
    ((512, 512), [(512, 512), (427, 427), (356, 356), (296, 296), (247, 247), (206, 206), (171, 171), (143, 143)]),
])
def test_pyramid_sizes(oracle, wh, levels):
    ex = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    ex.compute_pyramid(np.zeros((wh[1], wh[0]), np.uint8))
    got = [ex.level(l).shape[::-1] for l in range(8)]
    assert got == levels
