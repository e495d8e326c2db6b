"""Parity on NATURAL images (tests/golden/natural_images.npz, tools/gen_natural_fixture.py): scikit-image's `camera`
(512x512 = BASELINE C4's frame size), `astronaut`, and the rectified Middlebury `motorcycle` stereo pair (741x500).
Real texture has a different corner density (2-3 k FAST candidates at level 0 against ~15 k for synth.py), different tie
and min-threshold-cell statistics than the synthetic streams every other test uses.  The reference's own inputs are
EuRoC / TUM-VI / ZED2 frames (Examples/Monocular/EuRoC.yaml:33-63), which are not on disk.
CPU: the oracle's summary on these frames is pinned (a regression guard for the fixture and the oracle).
GPU: every stage and the end-to-end results of the HIP path equal the oracle bit for bit, stereo association included."""
import hashlib
import os

import numpy as np
import pytest

import orb_slam3_fast_amd as orbx

FIX = os.path.join(os.path.dirname(__file__), "golden", "natural_images.npz")
BF, B = 0.12 * 532.03, 0.12
MONO = ["camera", "astronaut", "moto_left", "moto_right"]


@pytest.fixture(scope="module")
def nat():
    return np.load(FIX)


def _kp_bytes(k):
    return np.ascontiguousarray(k).view(np.uint8).reshape(len(k), 28)


def test_fixture_and_oracle_summary(oracle, nat):
    assert nat["camera"].shape == (512, 512) and nat["astronaut"].shape == (512, 512)
    assert nat["moto_left"].shape == (500, 741) and nat["moto_right"].shape == (500, 741)
    assert hashlib.sha256(nat["camera"].tobytes()).hexdigest().startswith("5cb24482a53416f9")
    assert hashlib.sha256(nat["moto_left"].tobytes()).hexdigest().startswith("ba1aedab5d51d5b9")
    want = {"camera": (1503, 169), "astronaut": (1508, 55)}
    for name, (n, nmin) in want.items():
        oe = oracle.OracleExtractor(1500)
        _, k, d = oe.extract(nat[name])
        assert (len(k), int((k["response"] < 20).sum())) == (n, nmin)   # some cells needed the minThFAST fallback
        assert len(oe.detect_candidates(0)["x"]) < 3000                  # natural density, not synth.py's ~15 k
    oL, oR = oracle.OracleExtractor(1500), oracle.OracleExtractor(1500)
    _, kL, dL = oL.extract(nat["moto_left"])
    _, kR, dR = oR.extract(nat["moto_right"])
    u, dep = oracle.stereo_match(oL, oR, kL, dL, kR, dR, BF, B)
    assert (len(kL), len(kR), int((u >= 0).sum())) == (1504, 1508, 595)
    disp = (kL["x"] - u)[u >= 0]
    assert 30 < np.median(disp) < 60 and (disp >= 0).all()             # a real scene: tens of pixels of disparity


@pytest.fixture(scope="module")
def gpu():
    if orbx.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the MI355X box")
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("name", MONO)
def test_stages_and_end_to_end_on_natural_images(gpu, oracle, nat, name):
    img = nat[name]
    h, w = img.shape
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    oe = oracle.OracleExtractor(1500, 1.2, 8, 20, 7)
    mono, k, d = ex(img, (0, 0))
    omono, ok_, od = oe.extract(img, (0, 0))
    for l in range(8):
        assert np.array_equal(ex.image_pyramid(l), oe.level(l)), "pyramid level %d" % l
        assert np.array_equal(ex.image_pyramid(l, blurred=True), oracle.blur(oe.level(l))), "blur level %d" % l
        c = oe.detect_candidates(l)
        want = np.stack([c["x"], c["y"], c["response"]], 1).astype(np.int32)
        got = ex.debug_candidates(l)
        assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])]), "candidates level %d" % l
    assert mono == omono and len(k) == len(ok_) >= 1500
    assert np.array_equal(_kp_bytes(k), _kp_bytes(ok_))
    assert np.array_equal(d, od)


@pytest.mark.gpu
def test_natural_stereo_pair(gpu, oracle, nat):
    """ORBextractor on both eyes + Frame::ComputeStereoMatches (src/Frame.cc:921-1084) on the motorcycle pair: uRight / depth
    raw float bits equal the oracle's; also through orbx_extract_stereo (one call, one synchronisation)."""
    L, R = nat["moto_left"], nat["moto_right"]
    h, w = L.shape
    exL = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    exR = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h)
    oL, oR = oracle.OracleExtractor(1500), oracle.OracleExtractor(1500)
    _, kL, dL = exL(L)
    _, kR, dR = exR(R)
    _, okL, odL = oL.extract(L)
    _, okR, odR = oR.extract(R)
    assert np.array_equal(_kp_bytes(kL), _kp_bytes(okL)) and np.array_equal(_kp_bytes(kR), _kp_bytes(okR))
    assert np.array_equal(dL, odL) and np.array_equal(dR, odR)
    u, dep = orbx.ComputeStereoMatches(exL, exR, BF, B)
    ou, od = oracle.stereo_match(oL, oR, okL, odL, okR, odR, BF, B)
    n = len(kL)
    assert (ou >= 0).sum() == 595
    assert np.array_equal(u[0, :n].view(np.uint32), ou.view(np.uint32))
    assert np.array_equal(dep[0, :n].view(np.uint32), od.view(np.uint32))
    ex2 = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=2)
    (_, kl2, dl2), (_, kr2, dr2), (u2, dep2) = ex2.extract_stereo(L, R, bf=BF, b=B)
    assert np.array_equal(dl2, odL) and np.array_equal(dr2, odR)
    assert np.array_equal(_kp_bytes(kl2), _kp_bytes(okL)) and np.array_equal(_kp_bytes(kr2), _kp_bytes(okR))
    assert np.array_equal(np.asarray(u2)[:n].view(np.uint32), ou.view(np.uint32))
    assert np.array_equal(np.asarray(dep2)[:n].view(np.uint32), od.view(np.uint32))


@pytest.mark.gpu
def test_natural_images_in_the_batched_device_mode(gpu, oracle, nat):
    """The many-camera entry (orbx_extract_batch_device) on natural frames: camera + astronaut in one batch."""
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    d = DeviceBuffer.from_numpy(np.stack([nat["camera"], nat["astronaut"]]))
    ex = orbx.ORBextractor(1500, 1.2, 8, 20, 7, max_width=512, max_height=512, max_batch=2)
    ex.extract_batch_device(d.ptr.value, 2, 512, 512, 512, 512 * 512)
    ex.sync()
    for i, name in enumerate(("camera", "astronaut")):
        oe = oracle.OracleExtractor(1500)
        om, ok_, od = oe.extract(nat[name])
        m, k, dd = ex.download(i)
        assert m == om and np.array_equal(_kp_bytes(k), _kp_bytes(ok_)) and np.array_equal(dd, od)
