"""Minimal device-memory plumbing over the HIP runtime liborbx.so is linked against (ctypes).

Used by tests and by bench.py when torch is not the allocator.  Loading by SONAME returns the runtime
already mapped into the process (torch's bundled one if torch was imported first, /opt/rocm's otherwise),
so there is never a second HIP runtime in the process.
"""
import ctypes as C

import numpy as np

from . import lib as _orbx_lib

_hip = None


def hip():
    global _hip
    if _hip is None:
        _orbx_lib()
        try:
            _hip = C.CDLL("libamdhip64.so.7")
        except OSError:
            _hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        _hip.hipHostFree.argtypes = [C.c_void_p]
        _hip.hipGetErrorString.restype = C.c_char_p
        _hip.hipGetErrorString.argtypes = [C.c_int]
    return _hip


def _ck(rc):
    if rc != 0:
        raise RuntimeError("HIP error %d: %s" % (rc, hip().hipGetErrorString(rc).decode()))


class DeviceBuffer:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        _ck(hip().hipMalloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        _ck(hip().hipMemcpy(b.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1))
        return b

    def to_numpy(self, dtype, shape):
        out = np.zeros(shape, dtype)
        _ck(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2))
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pinned_like(a):
    """A page-locked host copy of `a` (hipHostMalloc): uploads from it are plain DMA, not staged through the driver's
    bounce buffer -- what a capture pipeline gets by registering its frame buffers once.  Tools / tests only: the block
    stays allocated for the life of the process."""
    a = np.ascontiguousarray(a)
    ptr = C.c_void_p()
    _ck(hip().hipHostMalloc(C.byref(ptr), max(a.nbytes, 1), 0))
    buf = (C.c_uint8 * max(a.nbytes, 1)).from_address(ptr.value)
    out = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
    out[...] = a
    _PINNED_BLOCKS.append(ptr)
    return out


_PINNED_BLOCKS = []
