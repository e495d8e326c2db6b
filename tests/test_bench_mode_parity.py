"""Parity of EXACTLY the mode bench.py times: bench.Workload with its default arguments -- 32 distinct
1280x720 stereo streams per batch, a ring of 3 frames, four extractor handles rotating, 7 steps enqueued back to back with
no synchronisation in between, stereo association queued behind each batch -- and then EVERY image of ALL handles' last
batches (keypoints, descriptors, uRight, depth) against the CPU oracle, bit for bit.  Cross-handle ordering bugs (the
k_detect token, the side stream's events, buffers reused while the other batch is in flight) would show here.
The C5 flavour (8 streams x 4 consecutive frames, RCCL all-gather queued on the handles' streams, 1-rank RCCL) checks the
gathered {n, desc[cap][32]} blocks against orbx_batch_download of the same images."""
import os
import sys

import numpy as np
import pytest

try:  # at COLLECTION time, i.e. before any test has loaded liborbx.so: liborbx then binds to the HIP runtime torch brings
    import torch  # noqa: F401  (whatever the order the tests run in; loaded the other way round torch finds no GPU)
except ImportError:  # pragma: no cover
    torch = None

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _kb(k):
    return np.ascontiguousarray(k).view(np.uint8).reshape(len(k), 28)


def _check_handles(wl, bench, steps):
    import orb_slam3_fast_amd as orbx
    from oracle import cpu_bench
    a = wl.a
    B = a.pairs
    for _ in range(steps):
        wl.step()                      # no synchronisation between the steps: exactly the timed loop of bench.py
    wl.sync()
    checked = 0
    for h, ex in enumerate(wl.exs):
        slot = wl.last_slot[h]
        assert slot is not None
        ref = cpu_bench.oracle_pairs(wl.host_left[slot], wl.host_right[slot], a.nfeatures, bench.BF, bench.BASE)
        u, dep = np.zeros((B, ex.capacity), np.float32), np.zeros((B, ex.capacity), np.float32)
        for p in range(B):
            orbx._check(orbx.lib().orbx_stereo_download(ex._h, p, orbx._p(u[p]), orbx._p(dep[p]), ex.capacity))
        for p in range(B):
            okL, odL, okR, odR, ou, od = ref[p]
            _, kL, dL = ex.download(p)
            _, kR, dR = ex.download(B + p)
            assert np.array_equal(_kb(kL), _kb(okL)) and np.array_equal(dL, odL), (h, p, "left")
            assert np.array_equal(_kb(kR), _kb(okR)) and np.array_equal(dR, odR), (h, p, "right")
            n = len(kL)
            assert np.array_equal(u[p, :n].view(np.uint32), ou.view(np.uint32)), (h, p, "uRight")
            assert np.array_equal(dep[p, :n].view(np.uint32), od.view(np.uint32)), (h, p, "depth")
            checked += 1
    return checked


def test_default_bench_workload_matches_the_oracle():
    import bench
    a = bench.parse([])                              # bench.py's defaults: that IS the point
    assert (a.pairs, a.distinct, a.handles, a.width, a.height, a.nfeatures, a.ring) == (32, 32, 4, 1280, 720, 1500, 3)
    wl = bench.Workload(a)
    assert len({s for s in wl.streams}) == 32
    # distinct inputs: no two pairs of a batch are the same image
    assert len({wl.host_left[0, p].tobytes()[:4096 * 64] for p in range(a.pairs)}) == a.pairs
    assert _check_handles(wl, bench, 7) == 128       # every handle's last batch: steps 3, 4, 5, 6 (ring slots 0, 1, 2, 0)


def test_c5_allgather_blocks_equal_the_downloads():
    """bench.py --config C5 on one GPU with a 1-rank RCCL group: 8 streams x 4 consecutive frames per step, the all-gather
    of {n, desc[cap][32]} through the C ABI (orbx_allgather_descriptors: one grouped RCCL call straight from the handle's result
    arrays, queued on the handle's stream behind the extraction; sharding.DescriptorExchange is the thin caller)."""
    import torch
    import torch.distributed as dist
    import bench
    a = bench.parse(["--config", "C5", "--ring", "2"])
    assert a.pairs == 32 and a.distinct == 8 and a.allgather
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        wl = bench.Workload(a, 0, 0, dist)
        # consecutive frames of a stream sit side by side in the batch
        assert not np.array_equal(wl.host_left[0, 0], wl.host_left[0, 1]) and np.array_equal(wl.host_left[0, 1], wl.host_left[1, 0])
        assert _check_handles(wl, bench, 5) == 32 * len(wl.exs)
        torch.cuda.synchronize()
        for h, ex in enumerate(wl.exs):
            cnt, desc = wl.gathered[h]
            cnt, desc = cnt.cpu().numpy(), desc.cpu().numpy()
            assert cnt.shape == (2 * a.pairs,) and desc.shape == (2 * a.pairs, ex.capacity, 32)
            mine_c, mine_d = np.zeros(2 * a.pairs, np.int32), np.zeros((2 * a.pairs, ex.capacity, 32), np.uint8)
            for i in range(2 * a.pairs):
                _, k, d = ex.download(i)
                assert cnt[i] == len(k) and np.array_equal(desc[i, :len(k)], d), (h, i)
                mine_c[i] = len(k)
                mine_d[i, :len(k)] = d
            if h == 0:
                # the torch twins the gloo CPU test runs at world size 2 (tests/test_sharding_gloo.py), on the SAME inputs through
                # RCCL: packed blocks and the two-member form give what orbx_allgather_descriptors wrote (valid rows compared;
                # rows past n are whatever the device arrays hold)
                from orb_slam3_fast_amd import sharding
                tc, td = torch.from_numpy(mine_c).cuda(), torch.from_numpy(mine_d).cuda()
                for fn in (lambda: sharding.allgather_descriptor_blocks(tc, td, ex.capacity), lambda: sharding.allgather_members(tc, td)):
                    gc, gd = fn()
                    gc, gd = gc.cpu().numpy(), gd.cpu().numpy()
                    assert np.array_equal(gc, cnt)
                    for i in range(2 * a.pairs):
                        assert np.array_equal(gd[i, :cnt[i]], desc[i, :cnt[i]]), i
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.gpu
def test_h2d_inclusive_mode_matches_oracle(oracle):
    """The mode bench.py's `h2d_inclusive_value` times: frames in HOST memory every step (orbx_extract_batch: upload on the handle's
    stream), extraction + ComputeStereoMatches, ALL results back through orbx_batch_download_async into host arrays, handles
    alternating without a synchronisation in between -- every pair of the last two steps against the oracle; dense frames (the
    tall 2-D upload) and a strided source (one 2-D copy per image)."""
    import bench
    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd import synth
    from oracle import cpu_bench
    BF, BASE = bench.BF, bench.BASE
    assert orbx.device_count() > 0
    W, H, B, NF = 640, 480, 4, 1000
    pairs = [synth.stereo_pair(W, H, 400 + i) for i in range(2 * B)]
    exs = [orbx.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * B) for _ in range(2)]
    cap = exs[0].capacity
    outs = []
    for step in range(2):
        ex = exs[step]
        sel = pairs[step * B:(step + 1) * B]
        if step == 0:   # dense [2B, H, W]
            host = np.ascontiguousarray(np.stack([p[0] for p in sel] + [p[1] for p in sel]))
            row_pitch, img_pitch = W, W * H
        else:           # strided rows and a gap between images
            host = np.zeros((2 * B, H + 3, W + 24), np.uint8)
            for i, im in enumerate([p[0] for p in sel] + [p[1] for p in sel]):
                host[i, :H, :W] = im
            row_pitch, img_pitch = W + 24, (H + 3) * (W + 24)
        r = dict(cnt=np.zeros(2 * B, np.int32), mono=np.zeros(2 * B, np.int32), kps=np.zeros((2 * B, cap), orbx.KP_DTYPE),
                 desc=np.zeros((2 * B, cap, 32), np.uint8), ur=np.zeros((B, cap), np.float32), dp=np.zeros((B, cap), np.float32), host=host)
        ex.extract_batch_host(host.ctypes.data, 2 * B, W, H, row_pitch, img_pitch)
        orbx.stereo_match_async(ex, ex, BF, BASE, first_left=0, first_right=B, n_pairs=B)
        ex.download_async(r["cnt"].ctypes.data, r["mono"].ctypes.data, r["kps"].ctypes.data, r["desc"].ctypes.data,
                          r["ur"].ctypes.data, r["dp"].ctypes.data, B)
        outs.append(r)                       # no synchronisation between the two steps
    for ex in exs:
        ex.sync()
    for step in range(2):
        r = outs[step]
        sel = pairs[step * B:(step + 1) * B]
        ref = cpu_bench.oracle_pairs([p[0] for p in sel], [p[1] for p in sel], NF, BF, BASE)
        for p, (kL, dL, kR, dR, u, dep) in enumerate(ref):
            nL, nR = int(r["cnt"][p]), int(r["cnt"][B + p])
            assert (nL, nR) == (len(kL), len(kR)) and nL > 900
            assert r["kps"][p, :nL].tobytes() == kL.tobytes() and np.array_equal(r["desc"][p, :nL], dL)
            assert r["kps"][B + p, :nR].tobytes() == kR.tobytes() and np.array_equal(r["desc"][B + p, :nR], dR)
            assert r["ur"][p, :nL].tobytes() == u.tobytes() and r["dp"][p, :nL].tobytes() == dep.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,w,h,nf", [("mono", 640, 480, 1000), ("fisheye", 512, 512, 1500), ("stereo", 640, 480, 1000)])
def test_other_configs_workloads_match_the_oracle(oracle, mode, w, h, nf):
    """The workloads behind bench.py's `other_configs` legs (C2 640x480 mono, 640x480 stereo, C4 512x512 fisheye stereo with
    lapping areas + the batched 2-NN / KB8 association): bench.Workload with exactly the arguments that leg passes, steps enqueued
    without a synchronisation, then every image of every handle's last batch against the oracle; the fisheye association's integer
    outputs against the oracle on the same keypoints (its float tail has its own tolerance tests, tests/test_fisheye.py)."""
    import bench
    import orb_slam3_fast_amd as orbx
    a = bench.parse(["--mode", mode, "--width", str(w), "--height", str(h), "--nfeatures", str(nf), "--pairs", "8", "--distinct", "8"])
    wl = bench.Workload(a)
    for _ in range(5):
        wl.step()
    wl.sync()
    B = a.pairs
    checked = 0
    for hnd, ex in enumerate(wl.exs):
        slot = wl.last_slot[hnd]
        if slot is None:
            continue
        imgs = list(wl.host_left[slot]) + list(wl.host_right[slot])
        got = [ex.download(i) for i in range(2 * B)]
        for i, img in enumerate(imgs):
            lap = (0, 0) if wl.lap is None else tuple(int(v) for v in wl.lap[i])
            oe = oracle.OracleExtractor(nf)
            om, ok_, od = oe.extract(np.ascontiguousarray(img), lap)
            m, k, d = got[i]
            assert m == om and np.array_equal(_kb(k), _kb(ok_)) and np.array_equal(d, od), (mode, hnd, i)
            checked += 1
        if mode == "fisheye":
            sig2 = oracle.OracleExtractor(nf).tables()["sigma2"]
            for p in range(B):
                n, nd, l2r, r2l, dep, pts = orbx.fisheye_download(ex, ex, p)
                (mL, kL, dL), (mR, kR, dR) = got[p], got[B + p]
                on, ond, ol2r, or2l, odep, opts, gates = oracle.fisheye_stereo_match(kL, dL, mL, kR, dR, mR, wl.rig, sig2)
                assert nd == ond                      # ratio-test survivors: integer work, exact
                same = l2r[:len(kL)] == ol2r
                assert same.mean() > 0.995 and abs(n - on) <= 2   # accept / reject may flip only at a gate (tests/test_fisheye.py)
        elif mode == "stereo":
            ref = __import__("oracle.cpu_bench", fromlist=["x"]).oracle_pairs(wl.host_left[slot], wl.host_right[slot], nf, bench.BF, bench.BASE)
            for p in range(B):
                u, dep = np.zeros(ex.capacity, np.float32), np.zeros(ex.capacity, np.float32)
                orbx._check(orbx.lib().orbx_stereo_download(ex._h, p, orbx._p(u), orbx._p(dep), ex.capacity))
                n = len(ref[p][0])
                assert u[:n].tobytes() == ref[p][4].tobytes() and dep[:n].tobytes() == ref[p][5].tobytes()
    assert checked == 2 * B * len(wl.exs)
