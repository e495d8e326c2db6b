"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (orb_slam3_fast_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_LIB = os.path.join(_HERE, "liborb_oracle.so")
# ORB_ORACLE_LIB: another build of the same sources (bench.py's cpu_baseline leg times the -O3 -march=native build, `make fast`)
_LIB = os.environ.get("ORB_ORACLE_LIB") or _DEFAULT_LIB

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.cpp", "orb_oracle_c.cpp", "orb_oracle.h")]
    if _LIB != _DEFAULT_LIB:
        return _LIB
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.oro_create.restype = C.c_void_p
        _lib.oro_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _lib.oro_destroy.argtypes = [C.c_void_p]
        _lib.oro_level_ptr.restype = C.c_void_p
        _lib.oro_blurred_ptr.restype = C.c_void_p
        _lib.oro_pattern.restype = C.c_void_p
        _lib.oro_fast_atan2.restype = C.c_float
        _lib.oro_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.oro_sincosf.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        _lib.oro_sincos_model.argtypes = [C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        _lib.oro_sincos_check.restype = C.c_longlong
        _lib.oro_sincos_check.argtypes = [C.c_uint64, C.c_longlong, C.c_int, C.c_void_p]
        _lib.oro_libm_sincos_array.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p]
        _lib.oro_reachable_angles.argtypes = [C.c_uint64, C.c_longlong, C.c_void_p]
        _lib.oro_cv_round_f.argtypes = [C.c_float]
        _lib.oro_ic_angle.restype = C.c_float
        _lib.oro_kb8_triangulate.restype = C.c_float
        _lib.oro_kb8_triangulate.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_void_p]
        _lib.oro_kb8_unproject.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


class OracleExtractor:
    """Mirror of ORB_SLAM3::ORBextractor on the CPU oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = C.c_void_p(lib().oro_create(nfeatures, scale_factor, nlevels, ini_th, min_th))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oro_destroy(self.h)
            self.h = None

    def set_blur_taps(self, variant):
        """GaussianBlur taps of OpenCV 4.0 .. 4.5.0 (440: scalar model; 44016 / 44032: with the flooring 16- / 32-lane vector body
        of those releases' vertical pass) or >= 4.5.1 (451, the default): orb_oracle.cpp kBlurTaps440 / 451, gaussian_blur7."""
        lib().oro_set_blur_taps(self.h, int(variant))

    def tables(self):
        L = self.nlevels
        f = [np.zeros(L, np.float32) for _ in range(4)]
        nf = np.zeros(L, np.int32)
        um = np.zeros(16, np.int32)
        lib().oro_tables(self.h, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(nf), _p(um))
        return dict(scale=f[0], inv_scale=f[1], sigma2=f[2], inv_sigma2=f[3], nfeat=nf, umax=um)

    def extract(self, img, lap=(0, 0)):
        img = _u8(img)
        h, w = img.shape
        cap = self.nfeatures + 3 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        mono = lib().oro_extract(self.h, _p(img), w, h, C.c_long(img.strides[0]), lap[0], lap[1], _p(kps),
                                 _p(desc), cap, C.byref(n))
        if mono < 0:
            return mono, kps[:0], desc[:0]
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def compute_pyramid(self, img):
        img = _u8(img)
        h, w = img.shape
        lib().oro_compute_pyramid(self.h, _p(img), w, h, C.c_long(img.strides[0]))

    def level(self, l, blurred=False):
        w, h = C.c_int(), C.c_int()
        lib().oro_level_size(self.h, l, C.byref(w), C.byref(h))
        ptr = lib().oro_blurred_ptr(self.h, l) if blurred else lib().oro_level_ptr(self.h, l)
        buf = (C.c_uint8 * (w.value * h.value)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(h.value, w.value).copy()

    def detect_candidates(self, l):
        cap = 1 << 18
        out = np.zeros(cap, KP_DTYPE)
        n = lib().oro_detect_candidates(self.h, l, _p(out), cap)
        assert n >= 0
        return out[:n].copy()

    def distribute(self, cand, minX, maxX, minY, maxY, N):
        cand = np.ascontiguousarray(cand)
        out = np.zeros(N + 64, KP_DTYPE)
        n = lib().oro_distribute(self.h, _p(cand), len(cand), minX, maxX, minY, maxY, N, _p(out), len(out))
        assert n >= 0
        return out[:n].copy()


def resize(src, dw, dh):
    src = _u8(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oro_resize(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
    return dst


def fast(img, threshold, nms=True):
    img = _u8(img)
    cap = img.size
    out = np.zeros((cap, 3), np.int32)
    n = lib().oro_fast(_p(img), img.strides[0], img.shape[1], img.shape[0], threshold, int(nms), _p(out), cap)
    return out[:n].copy()


def fast_score_map(img, threshold):
    """cornerScore of every pixel passing the FAST-9-16 segment test at `threshold`, 0 elsewhere (pre-NMS)."""
    img = _u8(img)
    out = np.zeros(img.shape, np.uint8)
    lib().oro_fast_score_map(_p(img), img.strides[0], img.shape[1], img.shape[0], int(threshold), _p(out))
    return out


def blur(img, variant=451):
    img = _u8(img)
    dst = np.zeros_like(img)
    lib().oro_blur(_p(img), img.shape[1], img.shape[0], _p(dst), variant)
    return dst


def fast_atan2(y, x):
    return float(lib().oro_fast_atan2(float(y), float(x)))


def sincosf(a):
    s, c = C.c_float(), C.c_float()
    lib().oro_sincosf(C.c_float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def set_sincos_mode(mode):
    """0 = host libm sinf/cosf (default: the reference's own dependency), 1 / 2 = model of glibc's FMA / SSE2 variant."""
    lib().oro_set_sincos_mode(int(mode))


def host_libm_variant():
    return int(lib().oro_host_libm_variant())


def sincos_model(a, fused):
    s, c = C.c_float(), C.c_float()
    lib().oro_sincos_model(C.c_float(a), int(fused), C.byref(s), C.byref(c))
    return s.value, c.value


def sincos_check(seed, n, fused):
    """Mismatches of the glibc model against the host libm over n fastAtan2-reachable angles (+ the first bad angle)."""
    fb = C.c_float(0)
    bad = lib().oro_sincos_check(int(seed), int(n), int(fused), C.byref(fb))
    return int(bad), fb.value


def reachable_angles(seed, n):
    out = np.zeros(n, np.float32)
    lib().oro_reachable_angles(int(seed), int(n), _p(out))
    return out


def libm_sincos(angles):
    a = np.ascontiguousarray(angles, np.float32)
    s, c = np.zeros_like(a), np.zeros_like(a)
    lib().oro_libm_sincos_array(_p(a), a.size, _p(s), _p(c))
    return s, c


def ic_angle(img, cx, cy):
    img = _u8(img)
    return float(lib().oro_ic_angle(_p(img), img.shape[1], img.shape[0], cx, cy))


def descriptor(blurred, px, py, angle):
    blurred = _u8(blurred)
    out = np.zeros(32, np.uint8)
    lib().oro_descriptor(_p(blurred), blurred.shape[1], blurred.shape[0], C.c_float(px), C.c_float(py),
                         C.c_float(angle), _p(out))
    return out


def hamming(a, b):
    return int(lib().oro_hamming(_p(_u8(a)), _p(_u8(b))))


def stereo_match(exL, exR, kL, dL, kR, dR, bf, b):
    kL, kR = np.ascontiguousarray(kL), np.ascontiguousarray(kR)
    dL, dR = _u8(dL), _u8(dR)
    u = np.zeros(len(kL), np.float32)
    d = np.zeros(len(kL), np.float32)
    lib().oro_stereo_match(exL.h, exR.h, _p(kL), _p(dL), len(kL), _p(kR), _p(dR), len(kR), C.c_float(bf),
                           C.c_float(b), _p(u), _p(d))
    return u, d


def stereo_frame_mt(exL, exR, L, R, bf, b):
    """One stereo frame with the reference's thread structure (2 eye threads x per-level tasks) and REGISTER_TIMES timer
    placement: (kL, dL, kR, dR, uRight, depth, extract_ms, stereo_ms).  Results identical to the serial oracle."""
    L, R = _u8(L), _u8(R)
    h, w = L.shape
    cap = exL.nfeatures + 3 * exL.nlevels + 64
    kL, kR = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
    dL, dR = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    u, dep = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    nL, nR = C.c_int(0), C.c_int(0)
    ms = np.zeros(2, np.float64)
    rc = lib().oro_stereo_frame_mt(exL.h, exR.h, _p(L), _p(R), w, h, C.c_long(L.strides[0]), C.c_float(bf), C.c_float(b),
                                   _p(kL), _p(dL), C.byref(nL), _p(kR), _p(dR), C.byref(nR), cap, _p(u), _p(dep), _p(ms))
    assert rc == 0
    return (kL[:nL.value].copy(), dL[:nL.value].copy(), kR[:nR.value].copy(), dR[:nR.value].copy(), u[:nL.value].copy(),
            dep[:nL.value].copy(), float(ms[0]), float(ms[1]))


def bf_knn2(dQ, dT):
    dQ, dT = _u8(dQ), _u8(dT)
    nQ, nT = len(dQ), len(dT)
    idx = np.zeros((nQ, 2), np.int32)
    dist = np.zeros((nQ, 2), np.int32)
    ok = np.zeros(nQ, np.uint8)
    lib().oro_bf_knn2(_p(dQ), nQ, _p(dT), nT, _p(idx), _p(dist), _p(ok))
    return idx, dist, ok


def search_init(k1, d1, k2, d2, bounds, prev, window=100, nnratio=0.9, check_ori=True):
    k1, k2 = np.ascontiguousarray(k1), np.ascontiguousarray(k2)
    d1, d2 = _u8(d1), _u8(d2)
    prev = np.ascontiguousarray(prev, np.float32).copy()
    m = np.zeros(len(k1), np.int32)
    n = lib().oro_search_init(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), C.c_float(bounds[0]),
                              C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]), _p(prev),
                              _p(m), window, C.c_float(nnratio), int(check_ori))
    return n, m, prev


MP_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("track_depth", "<f4"),
                     ("predicted_level", "<i4"), ("in_view", "u1"), ("bad", "u1"), ("has_observations", "u1"),
                     ("pad_", "u1"), ("desc", "u1", (32,))])
assert MP_DTYPE.itemsize == 60


def search_by_projection(k, desc, uright, bounds, scale_factors, mps, th, far, th_far, nnratio, occupied):
    k = np.ascontiguousarray(k)
    desc = _u8(desc)
    mps = np.ascontiguousarray(mps, MP_DTYPE)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    occ = np.ascontiguousarray(occupied, np.uint8).copy()
    match = np.zeros(len(k), np.int32)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    n = lib().oro_search_by_projection(_p(k), _p(desc), None if ur is None else _p(ur), len(k), C.c_float(bounds[0]),
                                       C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]), _p(sf), len(sf),
                                       _p(mps), len(mps), C.c_float(th), int(far), C.c_float(th_far), C.c_float(nnratio),
                                       _p(occ), _p(match))
    return n, match, occ


def is_in_frustum(pose, pos, normal, min_dist, max_dist, bounds, view_cos_limit, log_scale_factor, nlevels, flags, desc):
    """Frame::isInFrustum + MapPoint::PredictScale for n points and ONE pose (20 floats: Rcw row-major, tcw, Ow, fx fy cx cy, bf).
    Returns (views MP_DTYPE [n], margins [n][2])."""
    pose = np.ascontiguousarray(pose, np.float32).reshape(20)
    pos, normal = np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(normal, np.float32)
    mn, mx = np.ascontiguousarray(min_dist, np.float32), np.ascontiguousarray(max_dist, np.float32)
    n = len(mn)
    views = np.zeros(n, MP_DTYPE)
    fl = np.ascontiguousarray(flags, np.uint8)
    views["bad"], views["has_observations"], views["desc"] = fl & 1, (fl >> 1) & 1, _u8(desc)
    margins = np.zeros((n, 2), np.float64)
    lib().oro_is_in_frustum(_p(pose), n, _p(pos), _p(normal), _p(mn), _p(mx), C.c_float(bounds[0]), C.c_float(bounds[1]),
                            C.c_float(bounds[2]), C.c_float(bounds[3]), C.c_float(view_cos_limit), C.c_float(log_scale_factor),
                            int(nlevels), _p(views), _p(margins))
    return views, margins


def is_in_frustum_kb8(pose, pos, normal, min_dist, max_dist, bounds, view_cos_limit, log_scale_factor, nlevels, flags, desc):
    """Frame::isInFrustumChecks for n points and ONE camera of a stereo-fisheye frame (23 floats: R row-major, t, twc, KB8
    parameters).  Returns (views MP_DTYPE [n] of that camera, margins [n][2])."""
    pose = np.ascontiguousarray(pose, np.float32).reshape(23)
    pos, normal = np.ascontiguousarray(pos, np.float32), np.ascontiguousarray(normal, np.float32)
    mn, mx = np.ascontiguousarray(min_dist, np.float32), np.ascontiguousarray(max_dist, np.float32)
    n = len(mn)
    views = np.zeros(n, MP_DTYPE)
    fl = np.ascontiguousarray(flags, np.uint8)
    views["bad"], views["has_observations"], views["desc"] = fl & 1, (fl >> 1) & 1, _u8(desc)
    margins = np.zeros((n, 2), np.float64)
    lib().oro_is_in_frustum_kb8(_p(pose), n, _p(pos), _p(normal), _p(mn), _p(mx), C.c_float(bounds[0]), C.c_float(bounds[1]),
                                C.c_float(bounds[2]), C.c_float(bounds[3]), C.c_float(view_cos_limit), C.c_float(log_scale_factor),
                                int(nlevels), _p(views), _p(margins))
    return views, margins


PP_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("angle", "<f4"), ("min_level", "<i4"),
                     ("max_level", "<i4"), ("valid", "u1"), ("has_observations", "u1"), ("pad_", "u1", (2,)),
                     ("desc", "u1", (32,))])
assert PP_DTYPE.itemsize == 64


def project_last_frame(pose, direction, pos, octave, angle, flags, desc, th, scale_factors, bounds):
    """The projection block of SearchByProjection(CurrentFrame, LastFrame) (src/ORBmatcher.cc:1606-1669) for n LastFrame points
    and ONE pose (12 floats: Sophus quaternion x y z w, translation, fx fy cx cy, bf).  Returns (views PP_DTYPE [n], margins [n])."""
    pose = np.ascontiguousarray(pose, np.float32).reshape(12)
    pos = np.ascontiguousarray(pos, np.float32)
    octave, angle = np.ascontiguousarray(octave, np.int32), np.ascontiguousarray(angle, np.float32)
    fl, sf = np.ascontiguousarray(flags, np.uint8), np.ascontiguousarray(scale_factors, np.float32)
    n = len(octave)
    views = np.zeros(n, PP_DTYPE)
    views["desc"] = _u8(desc)
    margins = np.zeros(n, np.float64)
    lib().oro_project_last_frame(_p(pose), int(direction), n, _p(pos), _p(octave), _p(angle), _p(fl), C.c_float(th), _p(sf), len(sf),
                                 C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]), _p(views),
                                 _p(margins))
    return views, margins


def search_by_projection_frame(k, desc, uright, bounds, pts, check_ori, occupied):
    k = np.ascontiguousarray(k)
    desc = _u8(desc)
    pts = np.ascontiguousarray(pts, PP_DTYPE)
    occ = np.ascontiguousarray(occupied, np.uint8).copy()
    match = np.zeros(len(k), np.int32)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    n = lib().oro_search_by_projection_frame(_p(k), _p(desc), None if ur is None else _p(ur), len(k),
                                             C.c_float(bounds[0]), C.c_float(bounds[1]), C.c_float(bounds[2]),
                                             C.c_float(bounds[3]), _p(pts), len(pts), int(check_ori), _p(occ), _p(match))
    return n, match, occ


def features_in_area(k, bounds, x, y, r, min_level, max_level):
    k = np.ascontiguousarray(k)
    out = np.zeros(len(k) + 1, np.int32)
    n = lib().oro_features_in_area(_p(k), len(k), C.c_float(bounds[0]), C.c_float(bounds[1]),
                                   C.c_float(bounds[2]), C.c_float(bounds[3]), C.c_float(x), C.c_float(y),
                                   C.c_float(r), min_level, max_level, _p(out), len(out))
    return out[:n].copy()


# ---- KannalaBrandt8 / ComputeStereoFishEyeMatches (float part of the path: tolerance parity) -------------------------
def kb8_rig(cam1, cam2, R12, t12, precision=1e-6):
    """The 29 floats of orbx_kb8_rig: cam1[8] cam2[8] precision R12[9] (row-major) t12[3]."""
    return np.concatenate([np.asarray(cam1, np.float32).ravel(), np.asarray(cam2, np.float32).ravel(),
                           np.array([precision], np.float32), np.asarray(R12, np.float32).ravel(),
                           np.asarray(t12, np.float32).ravel()]).astype(np.float32)


def kb8_project(cam, X):
    cam, X = np.ascontiguousarray(cam, np.float32), np.ascontiguousarray(X, np.float32)
    uv = np.zeros(2, np.float32)
    lib().oro_kb8_project(_p(cam), _p(X), _p(uv))
    return uv


def kb8_unproject(cam, u, v, precision=1e-6):
    cam = np.ascontiguousarray(cam, np.float32)
    ray = np.zeros(3, np.float32)
    lib().oro_kb8_unproject(_p(cam), precision, float(u), float(v), _p(ray))
    return ray


def null_vector4(A):
    A = np.ascontiguousarray(A, np.float32)
    v = np.zeros(4, np.float32)
    lib().oro_null_vector4(_p(A), _p(v))
    return v


def kb8_triangulate(rig, uv1, uv2, sigma1=1.0, sigma2=1.0):
    rig = np.ascontiguousarray(rig, np.float32)
    p = np.zeros(3, np.float32)
    gate = np.zeros(5, np.float32)
    d = lib().oro_kb8_triangulate(_p(rig), float(uv1[0]), float(uv1[1]), float(uv2[0]), float(uv2[1]), sigma1, sigma2,
                                  _p(p), _p(gate))
    return float(d), p, gate


def kb8_triangulate_ex(rig, uv1, uv2, sigma1=1.0, sigma2=1.0, xh=None):
    """kb8_triangulate with the 4x4 system returned and, optionally, the null vector supplied (study hook)."""
    rig = np.ascontiguousarray(rig, np.float32)
    p, gate, A = np.zeros(3, np.float32), np.zeros(5, np.float32), np.zeros(16, np.float32)
    x = None if xh is None else np.ascontiguousarray(xh, np.float32)
    lib().oro_kb8_triangulate_ex.restype = C.c_float
    lib().oro_kb8_triangulate_ex.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p] * 4
    d = lib().oro_kb8_triangulate_ex(_p(rig), float(uv1[0]), float(uv1[1]), float(uv2[0]), float(uv2[1]), sigma1, sigma2,
                                     None if x is None else _p(x), _p(A), _p(p), _p(gate))
    return float(d), p, gate, A.reshape(4, 4)


def fisheye_stereo_match(kL, dL, monoL, kR, dR, monoR, rig, level_sigma2):
    """-> nMatches, descMatches, leftToRight, rightToLeft, depth, p3D [nL,3], gates [nL,6]."""
    kL, kR = np.ascontiguousarray(kL), np.ascontiguousarray(kR)
    dL, dR = _u8(dL), _u8(dR)
    rig = np.ascontiguousarray(rig, np.float32)
    s2 = np.ascontiguousarray(level_sigma2, np.float32)
    nL, nR = len(kL), len(kR)
    l2r, r2l = np.zeros(nL, np.int32), np.zeros(nR, np.int32)
    dep, pts, gates = np.zeros(nL, np.float32), np.zeros((nL, 3), np.float32), np.zeros((nL, 6), np.float32)
    nd = C.c_int(0)
    nm = lib().oro_fisheye_stereo_match(_p(kL), _p(dL), nL, monoL, _p(kR), _p(dR), nR, monoR, _p(rig), _p(s2), len(s2),
                                        _p(l2r), _p(r2l), _p(dep), _p(pts), C.byref(nd), _p(gates))
    return nm, nd.value, l2r, r2l, dep, pts, gates


# ---- Frame::UndistortKeyPoints / ComputeImageBounds ---------------------------------------------------------------------
def undistort_keypoints(kps, K, dist):
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    K = np.ascontiguousarray(K, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(len(kps), KP_DTYPE)
    lib().oro_undistort_keypoints(_p(kps), len(kps), _p(K), _p(dist), len(dist), _p(out))
    return out


def image_bounds(cols, rows, K, dist):
    K = np.ascontiguousarray(K, np.float32)
    dist = np.ascontiguousarray(dist, np.float32)
    b = np.zeros(4, np.float32)
    lib().oro_image_bounds(int(cols), int(rows), _p(K), _p(dist), len(dist), _p(b))
    return b


# ---- stereo-fisheye branches of the SearchByProjection matchers -----------------------------------------------------------
MPR_DTYPE = np.dtype([("proj_yr", "<f4"), ("view_cos_r", "<f4"), ("predicted_level_r", "<i4"), ("in_view_r", "u1"),
                      ("pad_", "u1", (3,))])
assert MPR_DTYPE.itemsize == 16


def search_by_projection_fisheye(k, desc, n_left, bounds, scale_factors, mps, mps_r, th, far, th_far, nnratio, l2r, r2l, occupied):
    k = np.ascontiguousarray(k)
    desc = _u8(desc)
    mps, mps_r = np.ascontiguousarray(mps, MP_DTYPE), np.ascontiguousarray(mps_r, MPR_DTYPE)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    l2r, r2l = np.ascontiguousarray(l2r, np.int32), np.ascontiguousarray(r2l, np.int32)
    occ = np.ascontiguousarray(occupied, np.uint8).copy()
    match = np.zeros(len(k), np.int32)
    n = lib().oro_search_by_projection_fisheye(_p(k), _p(desc), n_left, len(k) - n_left, C.c_float(bounds[0]), C.c_float(bounds[1]),
                                               C.c_float(bounds[2]), C.c_float(bounds[3]), _p(sf), len(sf), _p(mps), _p(mps_r),
                                               len(mps), C.c_float(th), int(far), C.c_float(th_far), C.c_float(nnratio), _p(l2r),
                                               _p(r2l), _p(occ), _p(match))
    return n, match, occ


def search_by_projection_frame_fisheye(k, desc, n_left, bounds, pts, uv_right, check_ori, occupied):
    k = np.ascontiguousarray(k)
    desc = _u8(desc)
    pts = np.ascontiguousarray(pts, PP_DTYPE)
    uv = np.ascontiguousarray(uv_right, np.float32).reshape(-1, 2)
    occ = np.ascontiguousarray(occupied, np.uint8).copy()
    match = np.zeros(len(k), np.int32)
    n = lib().oro_search_by_projection_frame_fisheye(_p(k), _p(desc), n_left, len(k) - n_left, C.c_float(bounds[0]),
                                                     C.c_float(bounds[1]), C.c_float(bounds[2]), C.c_float(bounds[3]), _p(pts),
                                                     _p(uv), len(pts), int(check_ori), _p(occ), _p(match))
    return n, match, occ


# ---- image pre-processing (gray conversion, input resize) -----------------------------------------------------------------
def cvt_gray(img, rgb=True, variant=15):
    img = np.ascontiguousarray(img, np.uint8)
    h, w, cn = img.shape
    dst = np.zeros((h, w), np.uint8)
    lib().oro_cvt_gray(_p(img), w, h, C.c_long(img.strides[0]), cn, int(rgb), _p(dst), C.c_long(dst.strides[0]), variant)
    return dst


def resize_c(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    dst = np.zeros((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().oro_resize_c(_p(img), w, h, C.c_long(img.strides[0]), cn, _p(dst), dw, dh, C.c_long(dst.strides[0]))
    return dst


def remap(img, mapx, mapy):
    img = np.ascontiguousarray(img, np.uint8)
    mapx = np.ascontiguousarray(mapx, np.float32)
    mapy = np.ascontiguousarray(mapy, np.float32)
    h, w = img.shape
    dh, dw = mapx.shape
    dst = np.zeros((dh, dw), np.uint8)
    lib().oro_remap(_p(img), w, h, C.c_long(img.strides[0]), _p(mapx), _p(mapy), C.c_long(dw), _p(dst), dw, dh,
                    C.c_long(dst.strides[0]))
    return dst


def clahe(img, clip=3.0, tiles=(8, 8)):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    dst = np.zeros((h, w), np.uint8)
    lib().oro_clahe(_p(img), w, h, C.c_long(img.strides[0]), C.c_double(clip), tiles[0], tiles[1], _p(dst),
                    C.c_long(dst.strides[0]))
    return dst


# ---- f4: bag of words (DBoW2 vocabulary tree, transform, SearchByBoW) -------------------------------------------------------------
class Vocabulary:
    """Oracle twin of DBoW2::TemplatedVocabulary<FORB> (transform only).  Columns per node as in the ORBvoc.txt file."""

    def __init__(self, k=None, L=None, parent=None, is_leaf=None, desc=None, weight=None, scoring=0, weighting=0, path=None):
        L_ = lib()
        L_.oro_voc_create.restype = C.c_void_p
        L_.oro_voc_load.restype = C.c_void_p
        if path is not None:
            self._h = L_.oro_voc_load(path.encode())
            if not self._h:
                raise IOError("cannot load vocabulary " + path)
        else:
            parent = np.ascontiguousarray(parent, np.int32)
            is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            desc = _u8(desc)
            weight = np.ascontiguousarray(weight, np.float64)
            self._h = L_.oro_voc_create(k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf), _p(desc), _p(weight))
        info = np.zeros(6, np.int32)
        L_.oro_voc_info(C.c_void_p(self._h), _p(info))
        self.k, self.L, self.n_nodes, self.n_words, self.scoring, self.weighting = (int(x) for x in info)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oro_voc_destroy(C.c_void_p(self._h))
            self._h = None

    def save(self, path):
        assert lib().oro_voc_save(C.c_void_p(self._h), path.encode()) == 0

    def export(self):
        n = self.n_nodes
        parent, leaf, desc, weight = np.zeros(n, np.int32), np.zeros(n, np.uint8), np.zeros((n, 32), np.uint8), np.zeros(n, np.float64)
        lib().oro_voc_export(C.c_void_p(self._h), _p(parent), _p(leaf), _p(desc), _p(weight))
        return parent, leaf, desc, weight

    def transform_one(self, desc, levelsup=4):
        desc = _u8(desc)
        n = len(desc)
        w, wt, nd = np.zeros(n, np.int32), np.zeros(n, np.float64), np.zeros(n, np.int32)
        lib().oro_bow_transform_one(C.c_void_p(self._h), _p(desc), n, levelsup, _p(w), _p(wt), _p(nd))
        return w, wt, nd

    def transform(self, desc, levelsup=4):
        """-> (word_ids, values), (node_ids, node_start, feature_idx): BowVector and FeatureVector (CSR)."""
        desc = _u8(desc)
        n = len(desc)
        words, values = np.zeros(n, np.uint32), np.zeros(n, np.float64)
        nodes, start, feats = np.zeros(n, np.uint32), np.zeros(n + 1, np.int32), np.zeros(n, np.uint32)
        cnt = np.zeros(3, np.int32)
        lib().oro_bow_transform(C.c_void_p(self._h), _p(desc), n, levelsup, _p(words), _p(values), _p(nodes), _p(start), _p(feats),
                                _p(cnt))
        return (words[:cnt[0]].copy(), values[:cnt[0]].copy()), (nodes[:cnt[1]].copy(), start[:cnt[1] + 1].copy(), feats[:cnt[2]].copy())


def search_by_bow(kf_fv, kf_desc, kf_angle, kf_valid, f_fv, f_desc, f_angle, n_left_f=-1, nnratio=0.7, check_ori=True):
    kn, ks, kf = (np.ascontiguousarray(kf_fv[0], np.uint32), np.ascontiguousarray(kf_fv[1], np.int32), np.ascontiguousarray(kf_fv[2], np.uint32))
    fn, fs, ff = (np.ascontiguousarray(f_fv[0], np.uint32), np.ascontiguousarray(f_fv[1], np.int32), np.ascontiguousarray(f_fv[2], np.uint32))
    kd, fd = _u8(kf_desc), _u8(f_desc)
    ka, fa = np.ascontiguousarray(kf_angle, np.float32), np.ascontiguousarray(f_angle, np.float32)
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    match = np.zeros(len(fd), np.int32)
    n = lib().oro_search_by_bow(_p(kn), len(kn), _p(ks), _p(kf), _p(kd), _p(ka), _p(kv), _p(fn), len(fn), _p(fs), _p(ff), _p(fd), _p(fa),
                                len(fd), int(n_left_f), C.c_float(nnratio), int(check_ori), _p(match))
    return n, match


def search_by_projection_keyframe(k, desc, bounds, pts, orb_dist, check_ori, occupied):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1808-1918."""
    k = np.ascontiguousarray(k)
    desc = _u8(desc)
    pts = np.ascontiguousarray(pts, PP_DTYPE)
    occ = np.ascontiguousarray(occupied, np.uint8).copy()
    match = np.zeros(len(k), np.int32)
    n = lib().oro_search_by_projection_keyframe(_p(k), _p(desc), len(k), C.c_float(bounds[0]), C.c_float(bounds[1]),
                                                C.c_float(bounds[2]), C.c_float(bounds[3]), _p(pts), len(pts), int(orb_dist),
                                                int(check_ori), _p(occ), _p(match))
    return n, match, occ


def search_for_triangulation(fv1, k1, d1, has_mp1, uright1, fv2, k2, d2, has_mp2, uright2, scale_factors2, level_sigma2_2, ep, F12,
                             only_stereo=False, coarse=False, check_ori=True):
    """ORBmatcher::SearchForTriangulation, src/ORBmatcher.cc:886-1106 (single-camera key frames).  fv = (nodes, start, features)."""
    n1n, s1, f1 = (np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32))
    n2n, s2, f2 = (np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32))
    k1, k2 = np.ascontiguousarray(k1), np.ascontiguousarray(k2)
    d1, d2 = _u8(d1), _u8(d2)
    h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
    u1 = None if uright1 is None else np.ascontiguousarray(uright1, np.float32)
    u2 = None if uright2 is None else np.ascontiguousarray(uright2, np.float32)
    sf, sg = np.ascontiguousarray(scale_factors2, np.float32), np.ascontiguousarray(level_sigma2_2, np.float32)
    epa, Fa = np.ascontiguousarray(ep, np.float32), np.ascontiguousarray(F12, np.float32).reshape(9)
    m = np.zeros(len(k1), np.int32)
    n = lib().oro_search_for_triangulation(_p(n1n), len(n1n), _p(s1), _p(f1), _p(k1), _p(d1), _p(h1), None if u1 is None else _p(u1),
                                           len(k1), _p(n2n), len(n2n), _p(s2), _p(f2), _p(k2), _p(d2), _p(h2),
                                           None if u2 is None else _p(u2), len(k2), _p(sf), _p(sg), len(sf), _p(epa), _p(Fa),
                                           int(only_stereo), int(coarse), int(check_ori), _p(m))
    return n, m


FP_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("predicted_level", "<i4"), ("valid", "u1"),
                     ("pad_", "u1", (3,)), ("desc", "u1", (32,))])
assert FP_DTYPE.itemsize == 56


def fuse_search(k, desc, uright, bounds, inv_level_sigma2, pts, max_dist=50):
    """Search part of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight), src/ORBmatcher.cc:1195-1256 -> (nFused, bestIdx, bestDist)."""
    k = np.ascontiguousarray(k)
    desc = _u8(desc)
    pts = np.ascontiguousarray(pts, FP_DTYPE)
    ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
    isg = np.ascontiguousarray(inv_level_sigma2, np.float32)
    bi, bd = np.zeros(len(pts), np.int32), np.zeros(len(pts), np.int32)
    n = lib().oro_fuse_search(_p(k), _p(desc), None if ur is None else _p(ur), len(k), C.c_float(bounds[0]), C.c_float(bounds[1]),
                              C.c_float(bounds[2]), C.c_float(bounds[3]), _p(isg), len(isg), _p(pts), len(pts), int(max_dist), _p(bi), _p(bd))
    return n, bi, bd


def search_by_bow_keyframes(fv1, d1, angle1, valid1, fv2, d2, angle2, valid2, nnratio=0.75, check_ori=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12), src/ORBmatcher.cc:766-884 -> (nmatches, matches12[n1])."""
    n1n, s1, f1 = (np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32))
    n2n, s2, f2 = (np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32))
    d1, d2 = _u8(d1), _u8(d2)
    a1, a2 = np.ascontiguousarray(angle1, np.float32), np.ascontiguousarray(angle2, np.float32)
    v1, v2 = np.ascontiguousarray(valid1, np.uint8), np.ascontiguousarray(valid2, np.uint8)
    m = np.zeros(len(d1), np.int32)
    n = lib().oro_search_by_bow_keyframes(_p(n1n), len(n1n), _p(s1), _p(f1), _p(d1), _p(a1), _p(v1), len(d1), _p(n2n), len(n2n), _p(s2),
                                          _p(f2), _p(d2), _p(a2), _p(v2), len(d2), C.c_float(nnratio), int(check_ori), _p(m))
    return n, m


TRI_RIG_DTYPE = np.dtype([("cam", "<f4", (4, 8)), ("precision", "<f4"), ("R", "<f4", (4, 9)), ("t", "<f4", (4, 3))])
assert TRI_RIG_DTYPE.itemsize == 4 * (32 + 1 + 36 + 12)


def search_for_triangulation_rig(fv1, k1, d1, has_mp1, n_left1, fv2, k2, d2, has_mp2, n_left2, sigma1, sigma2, rig,
                                 only_stereo=False, coarse=False, check_ori=True):
    """Two-camera-rig branch of ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:906-923,1007-1064) ->
    (nmatches, matches12[n1], borderline[n1])."""
    n1n, s1, f1 = (np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32))
    n2n, s2, f2 = (np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32))
    k1, k2 = np.ascontiguousarray(k1), np.ascontiguousarray(k2)
    d1, d2 = _u8(d1), _u8(d2)
    h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
    sg1, sg2 = np.ascontiguousarray(sigma1, np.float32), np.ascontiguousarray(sigma2, np.float32)
    rig = np.ascontiguousarray(rig, TRI_RIG_DTYPE)
    m, bl = np.zeros(len(k1), np.int32), np.zeros(len(k1), np.uint8)
    n = lib().oro_search_for_triangulation_rig(_p(n1n), len(n1n), _p(s1), _p(f1), _p(k1), _p(d1), _p(h1), int(n_left1), len(k1), _p(n2n),
                                               len(n2n), _p(s2), _p(f2), _p(k2), _p(d2), _p(h2), int(n_left2), len(k2), _p(sg1), _p(sg2),
                                               len(sg1), _p(rig), int(only_stereo), int(coarse), int(check_ori), _p(m), _p(bl))
    return n, m, bl
