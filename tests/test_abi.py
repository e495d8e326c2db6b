"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/orbx.h declares, the host
tables match the oracle, and compute entry points fail loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import orb_slam3_fast_amd as orbx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "orbx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(orbx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = orbx.lib()
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), "liborbx.so does not export " + s
    assert lib.orbx_abi_version() == 1


def test_opencv_compat_rejects_a_null_handle():
    lib = orbx.lib()
    assert lib.orbx_set_opencv_compat(None, 440) == orbx.E_BADARG
    assert b"null handle" in lib.orbx_last_error()


def test_keypoint_layout_is_cv_keypoint():
    assert orbx.KP_DTYPE.itemsize == 28
    assert [orbx.KP_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == \
        [0, 4, 8, 12, 16, 20, 24]


def test_hamming_host_helper(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        a, b = rng.integers(0, 256, (2, 32), dtype=np.uint8)
        assert orbx.ORBmatcher.DescriptorDistance(a, b) == oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())


def test_no_cpu_fallback_without_device():
    if orbx.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(orbx.OrbxError) as e:
        orbx.ORBextractor(1000, 1.2, 8, 20, 7)
    assert e.value.code in (orbx.E_NODEVICE, orbx.E_HIP)
    with pytest.raises(orbx.OrbxError):
        orbx.bf_knn2(np.zeros((4, 32), np.uint8), np.zeros((4, 32), np.uint8))
    with pytest.raises(orbx.OrbxError):
        k = np.zeros(4, orbx.KP_DTYPE)
        orbx.ORBmatcher(0.9).SearchForInitialization(k, np.zeros((4, 32), np.uint8), k, np.zeros((4, 32), np.uint8),
                                                      (0, 0, 640, 480), np.zeros((4, 2), np.float32), 100)


def test_comm_entry_points_fail_loudly_without_device():
    """orbx_comm_* (RCCL behind the C ABI): argument errors are reported, and without a GPU the communicator is refused
    (one rank per GPU; there is no CPU collective behind this entry)."""
    lib = orbx.lib()
    h = C.c_void_p()
    assert lib.orbx_comm_create(None, 1, 0, 0, C.byref(h)) == orbx.E_BADARG
    buf = (C.c_uint8 * orbx.COMM_ID_BYTES)()
    assert lib.orbx_comm_create(buf, 2, 5, 0, C.byref(h)) == orbx.E_BADARG
    assert lib.orbx_allgather_descriptors(None, None, 1, None, None) == orbx.E_BADARG
    if orbx.device_count() == 0:
        assert lib.orbx_comm_create(buf, 1, 0, 0, C.byref(h)) == orbx.E_NODEVICE
        assert b"GPU" in lib.orbx_last_error()


@pytest.mark.gpu
def test_allgather_descriptors_through_the_c_abi_one_rank():
    """A 1-rank RCCL communicator created through the C ABI alone (no torch.distributed): the gathered blocks equal the
    handle's own results, and the call is ordered behind the extraction on the handle's stream."""
    from orb_slam3_fast_amd import synth
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    w, h, n = 320, 240, 4
    d = DeviceBuffer.from_numpy(np.stack([synth.stereo_pair(w, h, 40 + i)[0] for i in range(n)]))
    ex = orbx.ORBextractor(500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n)
    comm = orbx.Comm(orbx.comm_unique_id(), 1, 0, 0)
    nr, rk = C.c_int(), C.c_int()
    assert orbx.lib().orbx_comm_size(comm._h, C.byref(nr), C.byref(rk)) == 0 and (nr.value, rk.value) == (1, 0)
    all_desc = DeviceBuffer.from_numpy(np.full((n, ex.capacity, 32), 255, np.uint8))
    all_cnt = DeviceBuffer.from_numpy(np.full((n,), -1, np.int32))
    ex.extract_batch_device(d.ptr.value, n, w, h, w, w * h)
    ex.allgather_descriptors(comm, n, all_desc.ptr.value, all_cnt.ptr.value)   # no sync in between
    ex.sync()
    cnt, desc = all_cnt.to_numpy(np.int32, (n,)), all_desc.to_numpy(np.uint8, (n, ex.capacity, 32))
    for i in range(n):
        _, k, dd = ex.download(i)
        assert cnt[i] == len(k) > 50 and np.array_equal(desc[i, :len(k)], dd)
    with pytest.raises(orbx.OrbxError):
        ex.allgather_descriptors(comm, n + 1, all_desc.ptr.value, all_cnt.ptr.value)
    comm.close()


@pytest.mark.gpu
def test_one_communicator_serves_three_handles_back_to_back():
    """VERDICT (round 3), the shape of bench.py's C5 step: ONE communicator per rank shared by three handles on three streams,
    six gathers enqueued back to back with NO host synchronisation in between (the C ABI chains a communicator's collectives on
    the device: include/orbx.h, ordering rule), then the bounded orbx_comm_wait, then EVERY block of EVERY handle against the
    handle's own downloads.  A second round overwrites the first: the blocks must be those of the LAST batch of each handle."""
    from orb_slam3_fast_amd import synth
    from orb_slam3_fast_amd.hipmem import DeviceBuffer
    w, h, n, H = 320, 240, 4, 3
    frames = np.stack([synth.stereo_pair(w, h, 60 + i)[0] for i in range(2 * H * n)])          # [2H*n] distinct images
    dev = DeviceBuffer.from_numpy(frames)
    exs = [orbx.ORBextractor(500, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=n) for _ in range(H)]
    comm = orbx.Comm(orbx.comm_unique_id(), 1, 0, 0)
    assert comm.wait(10) == 0                                                                  # nothing enqueued yet
    outs = [(DeviceBuffer.from_numpy(np.full((n, e.capacity, 32), 255, np.uint8)), DeviceBuffer.from_numpy(np.full((n,), -1, np.int32)))
            for e in exs]
    for rnd in range(2):
        for k, e in enumerate(exs):
            e.extract_batch_device(dev.ptr.value + (rnd * H + k) * n * w * h, n, w, h, w, w * h)
            e.allgather_descriptors(comm, n, outs[k][0].ptr.value, outs[k][1].ptr.value)       # no sync anywhere in the loop
    assert comm.wait(30000) == 2 * H
    for k, e in enumerate(exs):
        e.sync()
        cnt, desc = outs[k][1].to_numpy(np.int32, (n,)), outs[k][0].to_numpy(np.uint8, (n, e.capacity, 32))
        for i in range(n):
            _, kp, dd = e.download(i)
            assert cnt[i] == len(kp) > 50 and np.array_equal(desc[i, :len(kp)], dd), (k, i)
    # the three handles saw different images: their blocks differ
    assert not np.array_equal(outs[0][0].to_numpy(np.uint8, (n, exs[0].capacity, 32)), outs[1][0].to_numpy(np.uint8, (n, exs[1].capacity, 32)))
    comm.close()


def test_copy_probe_rejects_bad_arguments_and_has_no_cpu_path():
    import ctypes
    lib = orbx.lib()
    gb = ctypes.c_double(0.0)
    assert lib.orbx_copy_probe(0, 1 << 20, 0, ctypes.byref(gb)) == orbx.E_BADARG          # iters
    assert lib.orbx_copy_probe(0, (1 << 20) + 8, 3, ctypes.byref(gb)) == orbx.E_BADARG    # not a multiple of 16
    assert lib.orbx_copy_probe(0, 1 << 20, 3, None) == orbx.E_BADARG
    if orbx.device_count() == 0:
        assert lib.orbx_copy_probe(0, 1 << 20, 3, ctypes.byref(gb)) == orbx.E_NODEVICE


@pytest.mark.gpu
def test_copy_probe_reads_a_plausible_device_copy_rate():
    """bench.py's `measured_copy_GBps`: a dwordx4 device-to-device copy of 256 MiB must land between 2 and 8 TB/s on an MI355X
    (the hardware guide measures 6.29 TB/s; the nominal HBM3E peak is 8)."""
    import ctypes
    gb = ctypes.c_double(0.0)
    orbx._check(orbx.lib().orbx_copy_probe(0, 256 << 20, 5, ctypes.byref(gb)))
    assert 2000.0 < gb.value < 8000.0, gb.value


def test_comm_wait_rejects_bad_arguments():
    lib = orbx.lib()
    assert lib.orbx_comm_wait(None, 10, None) == orbx.E_BADARG


def test_bad_parameters_rejected():
    lib = orbx.lib()
    p = orbx._Params(0, 1.2, 8, 20, 7)
    h = C.c_void_p()
    assert lib.orbx_extractor_create(C.byref(p), 640, 480, 1, 0, C.byref(h)) == orbx.E_BADARG
    p = orbx._Params(1000, 1.0, 8, 20, 7)
    assert lib.orbx_extractor_create(C.byref(p), 640, 480, 1, 0, C.byref(h)) == orbx.E_BADARG
    p = orbx._Params(1000, 1.2, 99, 20, 7)
    assert lib.orbx_extractor_create(C.byref(p), 640, 480, 1, 0, C.byref(h)) == orbx.E_BADARG
    assert b"invalid" in lib.orbx_last_error()


def test_introsort_replica_matches_libstdcxx(oracle):
    """The quadtree's tie order is std::sort's (src/ORBextractor.cc:686): replica vs the oracle's std::sort."""
    rng = np.random.default_rng(1)
    osort = oracle.lib().oro_std_sort_keys
    for trial in range(300):
        n = int(rng.integers(1, 700))
        cnt = rng.integers(2, 2 + int(rng.integers(1, 6)), n).astype(np.uint64)
        ulx = rng.integers(0, int(rng.integers(1, 9)), n).astype(np.uint64)
        v = (cnt << np.uint64(28)) | (ulx << np.uint64(16)) | np.arange(n, dtype=np.uint64)
        a, b = v.copy(), v.copy()
        orbx.lib().orbx_debug_introsort(a.ctypes.data_as(C.c_void_p), n)
        osort(b.ctypes.data_as(C.c_void_p), n)
        assert np.array_equal(a, b)


def test_header_is_plain_c99_and_struct_sizes(tmp_path):
    """include/orbx.h is the boundary a C, cgo or JNI binding would include: it must compile as C99 (no C++ in it), and the POD views
    keep the sizes the kernels and the oracle assume."""
    import subprocess
    src = tmp_path / "hc.c"
    src.write_text('#include "orbx.h"\n'
                   'int main(void) { return sizeof(orbx_keypoint) == 28 && sizeof(orbx_map_point_view) == 60 && '
                   'sizeof(orbx_projected_point) == 64 && sizeof(orbx_fuse_point) == 56 && sizeof(orbx_tri_rig) == 324 && '
                   'sizeof(orbx_kb8_rig) == 116 ? 0 : 1; }\n')
    exe = tmp_path / "hc"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + inc, str(src), "-o", str(exe)])
    assert subprocess.run([str(exe)]).returncode == 0
