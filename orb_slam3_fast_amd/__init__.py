"""orb_slam3_fast_amd — MI355X-native ORB front-end (extract + match) behind the reference's API surface.

Host-side mirror (Python, over the C ABI of include/orbx.h) of the reference's
`ORB_SLAM3::ORBextractor` (include/ORBextractor.h:49-118, src/ORBextractor.cc) and the matching routines
of `ORB_SLAM3::ORBmatcher` / `Frame` that are on the hot path (src/ORBmatcher.cc:618-764,1920-1973,
src/Frame.cc:921-1084,1273-1304).  Same names, argument meaning and error behaviour; all compute runs in
the hand-written HIP kernels of liborbx.so.  There is NO CPU fallback: importing works anywhere, but every
compute call raises if the library or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("ORBX_LIB_NAME", "liborbx.so"))   # ORBX_LIB_NAME: A/B runs of experiment builds (tools)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28  # cv::KeyPoint

OK, E_EMPTY, E_BADARG, E_CAPACITY, E_HIP, E_NODEVICE, E_UNSUPPORTED, E_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # src/ORBmatcher.cc:35-37
NUM_STAGES = 8
MP_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("view_cos", "<f4"), ("track_depth", "<f4"),
                     ("predicted_level", "<i4"), ("in_view", "u1"), ("bad", "u1"), ("has_observations", "u1"),
                     ("pad_", "u1"), ("desc", "u1", (32,))])   # orbx_map_point_view
assert MP_DTYPE.itemsize == 60
PP_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("angle", "<f4"), ("min_level", "<i4"),
                     ("max_level", "<i4"), ("valid", "u1"), ("has_observations", "u1"), ("pad_", "u1", (2,)),
                     ("desc", "u1", (32,))])   # orbx_projected_point
assert PP_DTYPE.itemsize == 64
FP_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("predicted_level", "<i4"), ("valid", "u1"),
                     ("pad_", "u1", (3,)), ("desc", "u1", (32,))])   # orbx_fuse_point
assert FP_DTYPE.itemsize == 56
TRI_RIG_DTYPE = np.dtype([("cam", "<f4", (4, 8)), ("precision", "<f4"), ("R", "<f4", (4, 9)), ("t", "<f4", (4, 3))])   # orbx_tri_rig
assert TRI_RIG_DTYPE.itemsize == 324
MPR_DTYPE = np.dtype([("proj_yr", "<f4"), ("view_cos_r", "<f4"), ("predicted_level_r", "<i4"), ("in_view_r", "u1"),
                      ("pad_", "u1", (3,))])   # orbx_map_point_right
assert MPR_DTYPE.itemsize == 16


class OrbxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbx error %d: %s" % (code, msg))
        self.code = code


class _Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class _PreprocParams(C.Structure):
    _fields_ = [("src_w", C.c_int32), ("src_h", C.c_int32), ("channels", C.c_int32), ("rgb_order", C.c_int32),
                ("out_w", C.c_int32), ("out_h", C.c_int32), ("map_x", C.c_void_p), ("map_y", C.c_void_p),
                ("map_stride", C.c_ssize_t), ("n_maps", C.c_int32), ("clahe_clip_limit", C.c_double),
                ("clahe_tiles_x", C.c_int32), ("clahe_tiles_y", C.c_int32)]


_lib = None


def lib():
    """Load liborbx.so (built by __graft_entry__.build() / csrc/Makefile).  Fails loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("liborbx.so not built (%s): run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` — there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.orbx_last_error.restype = C.c_char_p
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.orbx_extractor_create.argtypes = [C.POINTER(_Params), i, i, i, i, C.POINTER(vp)]
        L.orbx_extractor_destroy.argtypes = [vp]
        L.orbx_get_tables.argtypes = [vp] + [vp] * 6
        L.orbx_extract.argtypes = [vp, vp, i, i, C.c_ssize_t, i, i, vp, vp, i, C.POINTER(i)]
        L.orbx_extract_batch_device.argtypes = [vp, vp, i, i, i, C.c_ssize_t, C.c_ssize_t, vp]
        L.orbx_sync.argtypes = [vp]
        L.orbx_stream_handle.argtypes = [vp, C.POINTER(vp)]
        L.orbx_extract_batch.argtypes = [vp, vp, i, i, i, C.c_ssize_t, C.c_ssize_t, vp]
        L.orbx_batch_download_async.argtypes = [vp, vp, vp, vp, vp, vp, vp, i]
        L.orbx_extract_stereo.argtypes = [vp, vp, vp, i, i, C.c_ssize_t, C.c_ssize_t, vp, vp, vp, vp, i, vp, vp, vp, vp, i, vp, vp, f, f, vp, vp]
        L.orbx_batch_results_device.argtypes = [vp] + [C.POINTER(vp)] * 4 + [C.POINTER(i)]
        L.orbx_batch_download.argtypes = [vp, i, vp, vp, i, C.POINTER(i)]
        L.orbx_pyramid_level.argtypes = [vp, i, i, i, vp, C.c_ssize_t, C.POINTER(i), C.POINTER(i)]
        L.orbx_pyramid_download.argtypes = [vp, i, i, vp, vp]
        L.orbx_host_results.argtypes = [vp, i, C.POINTER(vp), C.POINTER(vp), C.POINTER(i), C.POINTER(i), C.POINTER(vp), C.POINTER(vp)]
        L.orbx_set_host_pyramid.argtypes = [vp, i]
        L.orbx_host_pyramid_level.argtypes = [vp, i, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i), C.POINTER(C.c_ssize_t)]
        L.orbx_debug_candidates.argtypes = [vp, i, i, vp, i]
        L.orbx_hamming256.argtypes = [vp, vp]
        L.orbx_stereo_match_batch.argtypes = [vp, i, vp, i, i, f, f]
        L.orbx_stereo_results_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.orbx_stereo_download.argtypes = [vp, i, vp, vp, i]
        L.orbx_bf_knn2.argtypes = [i, vp, i, vp, i, vp, vp, vp]
        L.orbx_fisheye_stereo_match.argtypes = [i, vp, vp, i, i, vp, vp, i, i, vp, vp, i, vp, vp, vp, vp, vp]
        L.orbx_fisheye_stereo_match_batch.argtypes = [vp, i, vp, i, i, vp]
        L.orbx_undistort_keypoints.argtypes = [i, vp, i, vp, vp, i, vp]
        L.orbx_cvt_gray.argtypes = [i, vp, i, i, C.c_ssize_t, i, i, vp, C.c_ssize_t]
        L.orbx_resize_linear.argtypes = [i, vp, i, i, C.c_ssize_t, i, vp, i, i, C.c_ssize_t]
        L.orbx_remap_linear.argtypes = [i, vp, i, i, C.c_ssize_t, i, vp, vp, C.c_ssize_t, vp, i, i, C.c_ssize_t]
        L.orbx_clahe.argtypes = [i, vp, i, i, C.c_ssize_t, C.c_double, i, i, vp, C.c_ssize_t]
        L.orbx_preproc_create.argtypes = [C.POINTER(_PreprocParams), i, i, C.POINTER(vp)]
        L.orbx_preproc_destroy.argtypes = [vp]
        L.orbx_preproc_destroy.restype = None
        L.orbx_preproc_output_size.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
        L.orbx_preproc_run.argtypes = [vp, vp, C.c_ssize_t, i, vp, C.c_ssize_t]
        L.orbx_preproc_run_device.argtypes = [vp, vp, i, C.c_ssize_t, C.c_ssize_t, C.POINTER(vp), C.POINTER(i), C.POINTER(i),
                                              C.POINTER(C.c_ssize_t), C.POINTER(C.c_ssize_t)]
        L.orbx_extract_batch_raw_device.argtypes = [vp, vp, vp, i, C.c_ssize_t, C.c_ssize_t, vp]
        L.orbx_vocabulary_create.argtypes = [i, i, i, i, i, i, vp, vp, vp, vp, C.POINTER(vp)]
        L.orbx_vocabulary_load_text.argtypes = [i, C.c_char_p, C.POINTER(vp)]
        L.orbx_vocabulary_destroy.argtypes = [vp]
        L.orbx_vocabulary_destroy.restype = None
        L.orbx_vocabulary_info.argtypes = [vp, vp]
        L.orbx_bow_transform.argtypes = [vp, vp, i, i, vp, vp, C.POINTER(i), vp, vp, vp, C.POINTER(i)]
        L.orbx_bow_transform_batch.argtypes = [vp, vp, i]
        L.orbx_bow_results_device.argtypes = [vp] + [C.POINTER(vp)] * 6 + [C.POINTER(i)]
        L.orbx_bow_download.argtypes = [vp, i, vp, vp, C.POINTER(i), vp, vp, vp, C.POINTER(i), i]
        L.orbx_search_by_bow.argtypes = [i, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, i, vp, vp, i, i, f, i, vp]
        L.orbx_search_by_projection_fisheye_batch.argtypes = [vp, i, i, i, f, f, f, f, vp, vp, vp, i, f, i, f, f, vp, vp, vp, vp, vp, vp]
        L.orbx_search_by_projection_frame_fisheye_batch.argtypes = [vp, i, i, i, f, f, f, f, vp, vp, vp, i, i, vp, vp, vp, vp]
        L.orbx_search_by_bow_batch.argtypes = [vp, i, i, vp, vp, vp, i, vp, vp, vp, vp, vp, i, i, f, i, vp, vp]
        L.orbx_search_by_projection_fisheye.argtypes = [i, vp, vp, i, i, f, f, f, f, vp, i, vp, vp, i, f, i, f, f, vp, vp, vp, vp]
        L.orbx_search_by_projection_frame_fisheye.argtypes = [i, vp, vp, i, i, f, f, f, f, vp, vp, i, i, vp, vp]
        L.orbx_compute_image_bounds.argtypes = [i, i, i, vp, vp, i, vp]
        L.orbx_fisheye_results_device.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orbx_fisheye_download.argtypes = [vp, i, vp, vp, vp, vp, i, i, vp]
        L.orbx_search_for_initialization.argtypes = [i, vp, vp, i, vp, vp, i, f, f, f, f, vp, vp, i, f, i]
        L.orbx_search_for_initialization_batch.argtypes = [vp, i, i, vp, vp, vp, i, f, f, f, f, vp, vp, i, f, i, vp]
        L.orbx_search_by_projection.argtypes = [i, vp, vp, vp, i, f, f, f, f, vp, i, vp, i, f, i, f, f, vp, vp]
        L.orbx_search_by_projection_frame.argtypes = [i, vp, vp, vp, i, f, f, f, f, vp, i, i, vp, vp]
        L.orbx_search_by_projection_frame_batch.argtypes = [vp, i, i, f, f, f, f, vp, vp, i, i, i, vp, vp, vp, vp]
        L.orbx_search_by_projection_batch.argtypes = [vp, i, i, f, f, f, f, vp, vp, i, f, i, f, f, i, vp, vp, vp, vp]
        L.orbx_search_by_projection_keyframe.argtypes = [i, vp, vp, i, f, f, f, f, vp, i, i, i, vp, vp]
        L.orbx_map_upload.argtypes = [vp, i, vp, vp, vp, vp, vp, vp]
        L.orbx_project_map_points_batch.argtypes = [vp, i, vp, f, f, f, f, f, vp, vp]
        L.orbx_last_frames_upload.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp]
        L.orbx_project_map_points_fisheye_batch.argtypes = [vp, i, vp, vp, f, f, f, f, f, vp, vp, vp]
        L.orbx_project_last_frames_batch.argtypes = [vp, i, vp, f, f, f, f, f, vp]
        L.orbx_search_for_triangulation.argtypes = [i, vp, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, vp, i, vp, vp, i,
                                                    vp, vp, i, i, i, vp]
        L.orbx_search_for_triangulation_rig.argtypes = [i, vp, vp, vp, i, vp, vp, vp, i, i, vp, vp, vp, i, vp, vp, vp, i, i, vp, vp, i, vp,
                                                        i, i, i, vp]
        L.orbx_search_by_bow_keyframes.argtypes = [i, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, i, f, i, vp]
        L.orbx_fuse_search.argtypes = [i, vp, vp, vp, i, f, f, f, f, vp, i, vp, i, i, vp, vp]
        L.orbx_features_in_area.argtypes = [i, vp, i, f, f, f, f, vp, i, vp, vp, i, vp, vp]
        L.orbx_comm_unique_id.argtypes = [vp]
        L.orbx_comm_create.argtypes = [vp, i, i, i, C.POINTER(vp)]
        L.orbx_comm_adopt.argtypes = [vp, i, C.POINTER(vp)]
        L.orbx_comm_destroy.argtypes = [vp]
        L.orbx_comm_destroy.restype = None
        L.orbx_comm_size.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
        L.orbx_allgather_descriptors.argtypes = [vp, vp, i, vp, vp]
        L.orbx_comm_wait.argtypes = [vp, i, C.POINTER(C.c_ulonglong)]
        L.orbx_clock_probe_start.argtypes = [i, i, C.POINTER(vp)]
        L.orbx_clock_probe_finish.argtypes = [vp, C.POINTER(C.c_double)]
        L.orbx_copy_probe.argtypes = [i, C.c_size_t, i, C.POINTER(C.c_double)]
        L.orbx_profile_enable.argtypes = [vp, i]
        L.orbx_profile_collect.argtypes = [vp, vp, vp]
        L.orbx_stage_name.restype = C.c_char_p
        L.orbx_stage_name.argtypes = [i]
        L.orbx_level_stats.argtypes = [vp, i, vp, vp, vp, vp]
        L.orbx_debug_introsort.argtypes = [vp, i]
        L.orbx_debug_introsort.restype = None
        L.orbx_debug_introsort_device.argtypes = [i, vp, i]
        L.orbx_debug_set_detect_list_cap.argtypes = [i]
        L.orbx_debug_set_detect_list_cap.restype = None
        L.orbx_debug_set_octree_global.argtypes = [i]
        L.orbx_debug_set_octree_global.restype = None
        L.orbx_debug_set_stereo_direct.argtypes = [i]
        L.orbx_debug_set_stereo_direct.restype = None
        L.orbx_debug_set_clahe_cell_kernel.argtypes = [i]
        L.orbx_debug_set_clahe_cell_kernel.restype = None
        L.orbx_debug_upload_results.argtypes = [vp, i, vp, vp, i, i]
        L.orbx_debug_set_remap_lds.argtypes = [i]
        L.orbx_debug_set_remap_lds.restype = None
        L.orbx_debug_set_resize_tail.argtypes = [i, i, i]
        L.orbx_debug_set_resize_tail.restype = None
        L.orbx_debug_resize_plan.argtypes = [vp, vp, vp, vp, i]
        L.orbx_debug_sincos.argtypes = [i, vp, i, i, vp, vp]
        L.orbx_debug_score_map.argtypes = [vp, i]
        L.orbx_set_opencv_compat.argtypes = [vp, i]
        L.orbx_debug_score_level.argtypes = [vp, i, i, vp, C.c_ssize_t]
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise OrbxError(rc, lib().orbx_last_error().decode())
    return rc


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().orbx_device_count()


class ORBextractor:
    """Mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:49-118).

    `ex(image, lap)` is operator() (src/ORBextractor.cc:1015-1106): returns (monoIndex, keypoints,
    descriptors) with keypoints as a structured array laid out like cv::KeyPoint.  An empty image returns
    (-1, [], []) like the reference.  The batched entry points (`extract_batch_device`) are the
    many-camera mode: independent images of one size through every kernel in one launch each.
    """

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width=1280, max_height=720,
                 max_batch=1, device=0):
        self.nfeatures, self.nlevels = int(nfeatures), int(nlevels)
        self.max_batch = max_batch
        p = _Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        h = C.c_void_p()
        _check(lib().orbx_extractor_create(C.byref(p), max_width, max_height, max_batch, device, C.byref(h)))
        self._h = h
        self.device = device
        L = self.nlevels
        t = [np.zeros(L, np.float32) for _ in range(4)]
        self._nfeat = np.zeros(L, np.int32)
        self._umax = np.zeros(16, np.int32)
        _check(lib().orbx_get_tables(h, _p(t[0]), _p(t[1]), _p(t[2]), _p(t[3]), _p(self._nfeat), _p(self._umax)))
        self._scale, self._inv_scale, self._sigma2, self._inv_sigma2 = t
        cap = C.c_int()
        _check(lib().orbx_batch_results_device(h, None, None, None, None, C.byref(cap)))
        self.capacity = cap.value

    def close(self):
        if getattr(self, "_h", None):
            lib().orbx_extractor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters, include/ORBextractor.h:65-83
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return float(self._scale[1]) if self.nlevels > 1 else 1.0

    def GetScaleFactors(self):
        return self._scale.copy()

    def GetInverseScaleFactors(self):
        return self._inv_scale.copy()

    def GetScaleSigmaSquares(self):
        return self._sigma2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self._inv_sigma2.copy()

    def features_per_level(self):
        return self._nfeat.copy()

    def umax(self):
        return self._umax.copy()

    # ---- operator()
    def _host_bufs(self):
        """Views of the handle's page-locked result block (orbx_host_results): the single-frame entries are called with NULL
        output arrays and the valid prefix is copied ONCE, byte-wise (numpy copies structured records field by field; with the
        per-call np.zeros of six arrays the wrapper's share of a 1280x720 stereo frame was 80 us)."""
        b = getattr(self, "_hb", None)
        if b is None:
            b = {"n": [C.c_int() for _ in range(4)], "lap": np.zeros(4, np.int32), "views": None}
            b["nref"] = [C.byref(x) for x in b["n"]]
            b["lapp"] = (b["lap"].ctypes.data, b["lap"].ctypes.data + 8)
            self._hb = b
        return b

    def _result_views(self, hb, nimg, stereo):
        """numpy views over the result block (its address is fixed for the life of the handle: a section is mapped the first
        time a call has filled it)."""
        v = hb["views"]
        if v is None:
            v = hb["views"] = {}
        cap = self.capacity
        for img in range(nimg):
            need_st = stereo and img == 0 and "ur" not in v
            if img in v and not need_st:
                continue
            pk, pd, pu, pz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            _check(lib().orbx_host_results(self._h, img, C.byref(pk), C.byref(pd), None, None, C.byref(pu), C.byref(pz)))
            if img not in v:
                kb = np.frombuffer((C.c_uint8 * (cap * 28)).from_address(pk.value), np.uint8).reshape(cap, 28)
                db = np.frombuffer((C.c_uint8 * (cap * 32)).from_address(pd.value), np.uint8).reshape(cap, 32)
                v[img] = (kb, db)
            if need_st:
                v["ur"] = np.frombuffer((C.c_float * cap).from_address(pu.value), np.float32)
                v["dp"] = np.frombuffer((C.c_float * cap).from_address(pz.value), np.float32)
        return v

    @staticmethod
    def _take(view, n):
        return view[0][:n].copy().view(KP_DTYPE).reshape(n), view[1][:n].copy()

    def __call__(self, image, vLappingArea=(0, 0)):
        if image is None or image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 image expected"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        b = self._host_bufs()
        mono = _check(lib().orbx_extract(self._h, image.ctypes.data, w, h, image.strides[0], int(vLappingArea[0]),
                                         int(vLappingArea[1]), None, None, self.capacity, b["nref"][0]))
        return (mono,) + self._take(self._result_views(b, 1, False)[0], b["n"][0].value)

    # ---- batched many-camera mode
    def extract_stereo(self, left, right, lap_left=(0, 0), lap_right=(0, 0), bf=0.0, b=0.0):
        """Both eyes of one stereo frame in one batched pipeline (orbx_extract_stereo; the handle needs max_batch >= 2).
        Returns ((monoL, kpsL, descL), (monoR, kpsR, descR)) and, when bf > 0, also (mvuRight, mvDepth) of the left eye."""
        L, R = np.asarray(left), np.asarray(right)
        if L.ndim != 2 or L.shape != R.shape:
            raise ValueError("two gray images of the same size")
        if L.dtype != np.uint8 or L.strides[-1] != 1:
            L = np.ascontiguousarray(L, np.uint8)
        if R.dtype != np.uint8 or R.strides[-1] != 1:
            R = np.ascontiguousarray(R, np.uint8)
        h, w = L.shape
        hb = self._host_bufs()
        lap = hb["lap"]
        lap[0], lap[1], lap[2], lap[3] = lap_left[0], lap_left[1], lap_right[0], lap_right[1]
        nl, nr, ml, mr = hb["n"]
        rl, rr, rml, rmr = hb["nref"]
        cap = self.capacity
        _check(lib().orbx_extract_stereo(self._h, L.ctypes.data, R.ctypes.data, w, h, L.strides[0], R.strides[0], hb["lapp"][0],
                                         hb["lapp"][1], None, None, cap, rl, rml, None, None, cap, rr, rmr,
                                         float(bf), float(b), None, None))
        v = self._result_views(hb, 2, bf > 0)
        out = ((ml.value,) + self._take(v[0], nl.value), (mr.value,) + self._take(v[1], nr.value))
        return out + ((v["ur"][:nl.value].copy(), v["dp"][:nl.value].copy()),) if bf > 0 else out

    def map_upload(self, world_pos, normal, min_distance, max_distance, desc, flags):
        """The local map as structure-of-arrays, resident on the device (orbx_map_upload): GetWorldPos / GetNormal [n][3],
        mfMinDistance / mfMaxDistance [n], GetDescriptor [n][32], flags bit 0 isBad, bit 1 Observations() > 0."""
        a = [np.ascontiguousarray(world_pos, np.float32), np.ascontiguousarray(normal, np.float32),
             np.ascontiguousarray(min_distance, np.float32), np.ascontiguousarray(max_distance, np.float32),
             np.ascontiguousarray(desc, np.uint8), np.ascontiguousarray(flags, np.uint8)]
        n = len(a[2])
        _check(lib().orbx_map_upload(self._h, n, *[_p(x) for x in a]))
        self._map_n = n

    def project_map_points(self, poses, bounds, viewing_cos_limit=0.5, skip=None, want_views=False):
        """Frame::isInFrustum of every uploaded map point for every pose ([n_frames][20] floats: Rcw row-major, tcw, Ow, fx fy cx cy,
        bf) on the device (orbx_project_map_points_batch).  Returns the views (MP_DTYPE [n_frames][n]) when want_views."""
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 20)
        F = len(poses)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8).reshape(F, self._map_n)
        views = np.zeros((F, self._map_n), MP_DTYPE) if want_views else None
        _check(lib().orbx_project_map_points_batch(self._h, F, _p(poses), bounds[0], bounds[1], bounds[2], bounds[3],
                                                   float(viewing_cos_limit), None if sk is None else _p(sk),
                                                   None if views is None else _p(views)))
        return views

    def project_map_points_fisheye(self, left_poses, right_poses, bounds, viewing_cos_limit=0.5, skip=None, want_views=False):
        """Frame::isInFrustum for stereo-fisheye frames on the device (orbx_project_map_points_fisheye_batch): one pose per camera
        and frame, 23 floats each (R row-major, t, twc, 8 KB8 parameters) -- the right camera's as the reference derives it
        (src/Frame.cc:1342-1351).  Returns (MP_DTYPE [F][n], MPR_DTYPE [F][n]) when want_views."""
        pl = np.ascontiguousarray(left_poses, np.float32).reshape(-1, 23)
        pr = np.ascontiguousarray(right_poses, np.float32).reshape(-1, 23)
        F = len(pl)
        sk = None if skip is None else np.ascontiguousarray(skip, np.uint8).reshape(F, self._map_n)
        vl = np.zeros((F, self._map_n), MP_DTYPE) if want_views else None
        vr = np.zeros((F, self._map_n), MPR_DTYPE) if want_views else None
        _check(lib().orbx_project_map_points_fisheye_batch(self._h, F, _p(pl), _p(pr), bounds[0], bounds[1], bounds[2], bounds[3],
                                                           float(viewing_cos_limit), None if sk is None else _p(sk),
                                                           None if vl is None else _p(vl), None if vr is None else _p(vr)))
        return (vl, vr) if want_views else None

    def last_frames_upload(self, n_points, world_pos, octave, angle, desc, flags):
        """The LastFrames of the batch's cameras as structure-of-arrays on the device (orbx_last_frames_upload): n_points [F];
        world_pos [F][stride][3], octave / angle / flags [F][stride], desc [F][stride][32]; flags bit 0 = map point present and no
        outlier, bit 1 = Observations() > 0."""
        npts = np.ascontiguousarray(n_points, np.int32)
        F = len(npts)
        pos = np.ascontiguousarray(world_pos, np.float32).reshape(F, -1, 3)
        stride = pos.shape[1]
        a = [pos, np.ascontiguousarray(octave, np.int32).reshape(F, stride), np.ascontiguousarray(angle, np.float32).reshape(F, stride),
             np.ascontiguousarray(desc, np.uint8).reshape(F, stride, 32), np.ascontiguousarray(flags, np.uint8).reshape(F, stride)]
        _check(lib().orbx_last_frames_upload(self._h, F, stride, _p(npts), *[_p(x) for x in a]))
        self._lf_n, self._lf_stride = npts.copy(), stride

    def project_last_frames(self, poses, directions, bounds, th, want_views=False):
        """The projection block of SearchByProjection(CurrentFrame, LastFrame) (src/ORBmatcher.cc:1606-1669) for every uploaded
        LastFrame point on the device (orbx_project_last_frames_batch): poses [F][12] floats (Sophus quaternion x y z w, translation,
        fx fy cx cy, bf), directions [F] (0 neither, 1 bForward, 2 bBackward).  Returns the views (PP_DTYPE [F][stride]) if asked."""
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 12)
        F = len(poses)
        rec = np.zeros(F, np.dtype([("p", "<f4", (12,)), ("direction", "<i4")]))
        rec["p"], rec["direction"] = poses, np.ascontiguousarray(directions, np.int32)
        views = np.zeros((F, self._lf_stride), PP_DTYPE) if want_views else None
        _check(lib().orbx_project_last_frames_batch(self._h, F, _p(rec), bounds[0], bounds[1], bounds[2], bounds[3], float(th),
                                                    None if views is None else _p(views)))
        return views

    def extract_batch_device(self, d_images_ptr, n_images, w, h, row_pitch, image_pitch, lap=None):
        """Enqueue extraction of device-resident images (raw device pointer).  Asynchronous."""
        lap_arr = None if lap is None else np.ascontiguousarray(lap, np.int32).reshape(n_images, 2)
        _check(lib().orbx_extract_batch_device(self._h, C.c_void_p(d_images_ptr), n_images, w, h, row_pitch,
                                               image_pitch, None if lap_arr is None else _p(lap_arr)))

    def stream_handle(self):
        """Address of the handle's hipStream_t (orbx_stream_handle), e.g. for torch.cuda.ExternalStream."""
        st = C.c_void_p()
        _check(lib().orbx_stream_handle(self._h, C.byref(st)))
        return st.value or 0

    def allgather_descriptors(self, comm, n_images, d_all_desc_ptr, d_all_counts_ptr):
        """orbx_allgather_descriptors: one grouped RCCL all-gather of the last batch's descriptor blocks (desc rows + counts)
        of every rank into the caller's DEVICE arrays, enqueued on this handle's stream.  Asynchronous."""
        _check(lib().orbx_allgather_descriptors(self._h, comm._h, n_images, C.c_void_p(d_all_desc_ptr),
                                                C.c_void_p(d_all_counts_ptr)))

    def extract_batch_host(self, images_ptr, n_images, w, h, row_pitch, image_pitch, lap=None):
        """orbx_extract_batch: frames in HOST memory (page-locked for real overlap); asynchronous."""
        lp = None if lap is None else _p(np.ascontiguousarray(lap, np.int32))
        _check(lib().orbx_extract_batch(self._h, C.c_void_p(images_ptr), n_images, w, h, row_pitch, image_pitch, lp))

    def download_async(self, counts_ptr, mono_ptr, kps_ptr, desc_ptr, uright_ptr=None, depth_ptr=None, n_pairs=0):
        """orbx_batch_download_async into host arrays given by address (page-locked for real asynchrony)."""
        _check(lib().orbx_batch_download_async(self._h, counts_ptr, mono_ptr, kps_ptr, desc_ptr, uright_ptr, depth_ptr, n_pairs))

    def extract_batch_raw_device(self, preproc, d_frames_ptr, n_frames, row_pitch, image_pitch, lap=None):
        """Pre-processing chain + extraction of device-resident RAW frames on this handle's stream.  Asynchronous."""
        lap_arr = None if lap is None else np.ascontiguousarray(lap, np.int32).reshape(n_frames, 2)
        _check(lib().orbx_extract_batch_raw_device(self._h, preproc._h, C.c_void_p(d_frames_ptr), n_frames, row_pitch,
                                                   image_pitch, None if lap_arr is None else _p(lap_arr)))

    def sync(self):
        _check(lib().orbx_sync(self._h))

    def download(self, image):
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int()
        mono = _check(lib().orbx_batch_download(self._h, image, _p(kps), _p(desc), self.capacity, C.byref(n)))
        return mono, kps[:n.value].copy(), desc[:n.value].copy()

    def results_device(self):
        """(d_kps, d_desc, d_counts, d_mono, cap) raw device pointers of the last extraction."""
        a, b, c, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        cap = C.c_int()
        _check(lib().orbx_batch_results_device(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d),
                                               C.byref(cap)))
        return a.value, b.value, c.value, d.value, cap.value

    # ---- measurement
    def profile_enable(self, on=True, stage=None):
        """on=True: bracket every kernel launch with HIP events; stage="k_detect" etc.: only that kernel."""
        mode = int(bool(on))
        if on and stage is not None:
            names = [lib().orbx_stage_name(s).decode() for s in range(NUM_STAGES)]
            mode = 2 + names.index(stage)
        _check(lib().orbx_profile_enable(self._h, mode))

    def profile_collect(self):
        """{stage name: (total ms, launches)} since the last collect (HIP events on the launch stream)."""
        ms = np.zeros(NUM_STAGES, np.float64)
        cnt = np.zeros(NUM_STAGES, np.int32)
        _check(lib().orbx_profile_collect(self._h, _p(ms), _p(cnt)))
        return {lib().orbx_stage_name(s).decode(): (float(ms[s]), int(cnt[s])) for s in range(NUM_STAGES)}

    def level_stats(self, image=0):
        L = self.nlevels
        w, h, nc, ns = (np.zeros(L, np.int32) for _ in range(4))
        _check(lib().orbx_level_stats(self._h, image, _p(w), _p(h), _p(nc), _p(ns)))
        return w, h, nc, ns

    # ---- mvImagePyramid (include/ORBextractor.h:86)
    def set_opencv_compat(self, opencv_version):
        """Gaussian taps of the reference build's OpenCV (include/orbx.h: orbx_set_opencv_compat): 440 = OpenCV 4.0 .. 4.5.0
        (README.md:101 "tested with 4.4.0") through the scalar fixed-point path, 44016 / 44032 = the same taps with the flooring 16- /
        32-lane vector body of those releases' vertical pass, 451 = OpenCV >= 4.5.1 (the default)."""
        _check(lib().orbx_set_opencv_compat(self._h, int(opencv_version)))

    def image_pyramid(self, level, image=0, blurred=False):
        w, h = C.c_int(), C.c_int()
        _check(lib().orbx_pyramid_level(self._h, image, level, int(blurred), None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _check(lib().orbx_pyramid_level(self._h, image, level, int(blurred), _p(out), out.strides[0],
                                        C.byref(w), C.byref(h)))
        return out

    def pyramid_download(self, image=0, out=None):
        """orbx_pyramid_download: every level of one image of the last extraction with ONE synchronisation (what the C++
        mirror's mvImagePyramid refresh does).  `out` = list of preallocated [h_l, w_l] uint8 arrays to reuse."""
        nl = self.GetLevels()
        if out is None:
            out = []
            w, h = C.c_int(), C.c_int()
            for l in range(nl):
                _check(lib().orbx_pyramid_level(self._h, image, l, 0, None, 0, C.byref(w), C.byref(h)))
                out.append(np.empty((h.value, w.value), np.uint8))
        ptrs = (C.c_void_p * nl)(*[a.ctypes.data for a in out])
        strides = (C.c_ssize_t * nl)(*[a.strides[0] for a in out])
        _check(lib().orbx_pyramid_download(self._h, image, nl, ptrs, strides))
        return out

    def set_host_pyramid(self, enable=True):
        """orbx_set_host_pyramid: the single-frame entries (`ex(image)`, `extract_stereo`) keep a host copy of the pyramid --
        the reference's public mvImagePyramid (include/ORBextractor.h:86) -- current, copied beside the frame's kernels."""
        _check(lib().orbx_set_host_pyramid(self._h, 1 if enable else 0))
        self._hpv = {}

    def host_pyramid(self, image=0):
        """orbx_host_pyramid_level for every level: numpy VIEWS of the handle's page-locked copy (no copy, no synchronisation;
        valid until the next extraction on this handle, like mvImagePyramid)."""
        key = None
        views = getattr(self, "_hpv", None)
        if views is None:
            views = self._hpv = {}
        p, w, h, st = C.c_void_p(), C.c_int(), C.c_int(), C.c_ssize_t()
        _check(lib().orbx_host_pyramid_level(self._h, image, 0, C.byref(p), C.byref(w), C.byref(h), C.byref(st)))
        key = (image, p.value, w.value, h.value, st.value)
        if key in views:   # the block and the geometry are unchanged: the views of the previous frame are this frame's
            return views[key]
        out = []
        for l in range(self.GetLevels()):
            _check(lib().orbx_host_pyramid_level(self._h, image, l, C.byref(p), C.byref(w), C.byref(h), C.byref(st)))
            buf = (C.c_uint8 * (st.value * (h.value - 1) + w.value)).from_address(p.value)
            a = np.frombuffer(buf, np.uint8)
            out.append(np.lib.stride_tricks.as_strided(a, (h.value, w.value), (st.value, 1), writeable=False))
        views[key] = out
        return out

    def debug_score_map(self, enable=True):
        """Test tap: following extractions also keep k_detect's pre-NMS FAST scores at iniThFAST (orbx_debug_score_map)."""
        _check(lib().orbx_debug_score_map(self._h, 1 if enable else 0))

    def debug_score_level(self, level, image=0):
        w, h = C.c_int(), C.c_int()
        _check(lib().orbx_pyramid_level(self._h, image, level, 0, None, 0, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value), np.uint8)
        _check(lib().orbx_debug_score_level(self._h, image, level, _p(out), w.value))
        return out

    def debug_resize_plan(self):
        """Fused small-level resize launches of the handle's current image size: [(first_level, n_levels, n_bands), ...]."""
        a, b, c = (np.zeros(16, np.int32) for _ in range(3))
        n = _check(lib().orbx_debug_resize_plan(self._h, _p(a), _p(b), _p(c), 16))
        return [(int(a[i]), int(b[i]), int(c[i])) for i in range(n)]

    def debug_candidates(self, level, image=0):
        cap = 1 << 20
        out = np.zeros((cap, 3), np.int32)
        n = _check(lib().orbx_debug_candidates(self._h, image, level, _p(out), cap))
        return out[:n].copy()


def clock_probe_start(device=0, spin_us=2000):
    """orbx_clock_probe_start: one wave spins for spin_us on a private stream; finish() returns the shader clock in GHz."""
    h = C.c_void_p()
    _check(lib().orbx_clock_probe_start(device, int(spin_us), C.byref(h)))
    return h


def clock_probe_finish(probe):
    g = C.c_double()
    _check(lib().orbx_clock_probe_finish(probe, C.byref(g)))
    return g.value


COMM_ID_BYTES = 128


def comm_unique_id():
    """orbx_comm_unique_id: the 128-byte RCCL id rank 0 creates and ships to the other ranks."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().orbx_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """RCCL communicator behind the C ABI (orbx_comm_create: collective over all ranks; one rank per GPU)."""

    def __init__(self, unique_id, n_ranks, rank, device=0):
        assert len(unique_id) == COMM_ID_BYTES
        self._h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(lib().orbx_comm_create(buf, n_ranks, rank, device, C.byref(self._h)))
        self.n_ranks, self.rank = n_ranks, rank

    def wait(self, timeout_ms=60000):
        """Block until the communicator's most recent collective has completed (OrbxError E_TIMEOUT after timeout_ms: a rank
        is missing or out of order).  Returns the number of collectives enqueued so far."""
        n = C.c_ulonglong(0)
        _check(lib().orbx_comm_wait(self._h, int(timeout_ms), C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().orbx_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ComputeStereoMatches(left, right, bf, b, first_left=0, first_right=0, n_pairs=1):
    """Frame::ComputeStereoMatches (src/Frame.cc:921-1084) on the last extractions of two extractors
    (or one extractor holding both eyes).  Returns (mvuRight, mvDepth) arrays [n_pairs][capacity]."""
    _check(lib().orbx_stereo_match_batch(left._h, first_left, right._h, first_right, n_pairs, bf, b))
    out_u = np.zeros((n_pairs, left.capacity), np.float32)
    out_d = np.zeros((n_pairs, left.capacity), np.float32)
    for p in range(n_pairs):
        _check(lib().orbx_stereo_download(left._h, p, _p(out_u[p]), _p(out_d[p]), left.capacity))
    return out_u, out_d


def stereo_match_async(left, right, bf, b, first_left=0, first_right=0, n_pairs=1):
    _check(lib().orbx_stereo_match_batch(left._h, first_left, right._h, first_right, n_pairs, bf, b))


def debug_sincos(angles, fused=True, device=0):
    """Test hook: the device's sinf / cosf (csrc/orbx_sincos.h: glibc's FMA or SSE2 variant) of angles in radians."""
    a = np.ascontiguousarray(angles, np.float32)
    sn, cs = np.zeros_like(a), np.zeros_like(a)
    _check(lib().orbx_debug_sincos(device, _p(a), a.size, 1 if fused else 0, _p(sn), _p(cs)))
    return sn, cs


def bf_knn2(descQ, descT, device=0):
    """cv::BFMatcher(NORM_HAMMING).knnMatch(Q, T, 2) + Lowe ratio 0.7 of ComputeStereoFishEyeMatches
    (src/Frame.cc:1293-1302). Returns (idx[nQ,2], dist[nQ,2], ratio_ok[nQ])."""
    q = np.ascontiguousarray(descQ, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(descT, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    ok = np.zeros(len(q), np.uint8)
    _check(lib().orbx_bf_knn2(device, _p(q), len(q), _p(t), len(t), _p(idx), _p(dist), _p(ok)))
    return idx, dist, ok


def kb8_rig(cam1, cam2, R12, t12, precision=1e-6):
    """orbx_kb8_rig as 29 float32: the two KannalaBrandt8 parameter vectors (fx fy cx cy k0 k1 k2 k3), the Newton stop
    of unproject, and mRlr (row-major) / mtlr (include/Frame.h:208-209)."""
    rig = np.concatenate([np.asarray(cam1, np.float32).ravel(), np.asarray(cam2, np.float32).ravel(),
                          np.array([precision], np.float32), np.asarray(R12, np.float32).ravel(),
                          np.asarray(t12, np.float32).ravel()]).astype(np.float32)
    if rig.size != 29:
        raise ValueError("kb8_rig: need 8 + 8 camera parameters, a 3x3 rotation and a 3-vector")
    return rig


def ComputeStereoFishEyeMatches(kpsL, descL, monoL, kpsR, descR, monoR, rig, level_sigma2, device=0):
    """Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1273-1331) incl. KannalaBrandt8::TriangulateMatches.
    Returns (nMatches, descMatches, mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvStereo3Dpoints[nL, 3])."""
    kl, kr = np.ascontiguousarray(kpsL, KP_DTYPE), np.ascontiguousarray(kpsR, KP_DTYPE)
    dl = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
    dr = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
    if len(dl) != len(kl) or len(dr) != len(kr):
        raise ValueError("one descriptor row per keypoint")
    rig = np.ascontiguousarray(rig, np.float32)
    if rig.size != 29:
        raise ValueError("rig: use kb8_rig()")
    s2 = np.ascontiguousarray(level_sigma2, np.float32)
    l2r, r2l = np.zeros(len(kl), np.int32), np.zeros(len(kr), np.int32)
    depth, pts = np.zeros(len(kl), np.float32), np.zeros((len(kl), 3), np.float32)
    nd = C.c_int(0)
    n = _check(lib().orbx_fisheye_stereo_match(device, _p(kl), _p(dl), len(kl), int(monoL), _p(kr), _p(dr), len(kr),
                                               int(monoR), _p(rig), _p(s2), len(s2), _p(l2r), _p(r2l), _p(depth),
                                               _p(pts), C.byref(nd)))
    return n, nd.value, l2r, r2l, depth, pts


def fisheye_match_async(left, right, rig, first_left=0, first_right=0, n_pairs=1):
    """Enqueue Frame::ComputeStereoFishEyeMatches for n_pairs image pairs of the extractors' last (batch) extraction;
    everything stays on the device (config C4 in batched mode)."""
    rig = np.ascontiguousarray(rig, np.float32)
    if rig.size != 29:
        raise ValueError("rig: use kb8_rig()")
    _check(lib().orbx_fisheye_stereo_match_batch(left._h, first_left, right._h, first_right, n_pairs, _p(rig)))


def fisheye_download(left, right, pair=0):
    """Host copy of one pair of the last fisheye_match_async: (nMatches, descMatches, l2r, r2l, depth, p3d), arrays
    sized by the handles' capacities (entries beyond the keypoint counts are -1 / 0)."""
    l2r, r2l = np.zeros(left.capacity, np.int32), np.zeros(right.capacity, np.int32)
    depth, pts = np.zeros(left.capacity, np.float32), np.zeros((left.capacity, 3), np.float32)
    nd = C.c_int(0)
    n = _check(lib().orbx_fisheye_download(left._h, pair, _p(l2r), _p(r2l), _p(depth), _p(pts), left.capacity,
                                           right.capacity, C.byref(nd)))
    return n, nd.value, l2r, r2l, depth, pts


def cvtColorGray(img, rgb=True, device=0):
    """cv::cvtColor(img, COLOR_RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY) as Tracking::GrabImage* calls it
    (src/Tracking.cc:1394-1412): img is H x W x 3|4 uint8, rgb = the reference's mbRGB."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, cn = img.shape
    dst = np.zeros((h, w), np.uint8)
    _check(lib().orbx_cvt_gray(device, _p(img), w, h, img.strides[0], cn, int(bool(rgb)), _p(dst), dst.strides[0]))
    return dst


def resize(img, dst_w, dst_h, device=0):
    """cv::resize(img, out, Size(dst_w, dst_h)) with INTER_LINEAR (src/System.cc:297-298) for H x W [x 3|4] uint8 images."""
    img = np.ascontiguousarray(img, np.uint8)
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    dst = np.zeros((dst_h, dst_w) if img.ndim == 2 else (dst_h, dst_w, cn), np.uint8)
    _check(lib().orbx_resize_linear(device, _p(img), w, h, img.strides[0], cn, _p(dst), dst_w, dst_h, dst.strides[0]))
    return dst


def remap(img, map_x, map_y, device=0):
    """cv::remap(img, out, map_x, map_y, cv::INTER_LINEAR) with CV_32FC1 maps (src/System.cc:294-295) for H x W [x 3|4] uint8."""
    img = np.ascontiguousarray(img, np.uint8)
    mx = np.ascontiguousarray(map_x, np.float32)
    my = np.ascontiguousarray(map_y, np.float32)
    if mx.shape != my.shape or mx.ndim != 2:
        raise ValueError("map_x and map_y must be two float images of the same size")
    cn = 1 if img.ndim == 2 else img.shape[2]
    h, w = img.shape[:2]
    dh, dw = mx.shape
    dst = np.zeros((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    _check(lib().orbx_remap_linear(device, _p(img), w, h, img.strides[0], cn, _p(mx), _p(my), dw, _p(dst), dw, dh, dst.strides[0]))
    return dst


class CLAHE:
    """cv::createCLAHE(clipLimit, tileGridSize) as the TUM-VI examples use it (Examples/Stereo/stereo_tum_vi.cc:100,142-143)."""

    def __init__(self, clipLimit=3.0, tileGridSize=(8, 8), device=0):
        self.clip, self.tiles, self.device = float(clipLimit), (int(tileGridSize[0]), int(tileGridSize[1])), device

    def apply(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim != 2:
            raise ValueError("CLAHE needs a single-channel image")
        h, w = img.shape
        dst = np.zeros((h, w), np.uint8)
        _check(lib().orbx_clahe(self.device, _p(img), w, h, img.strides[0], self.clip, self.tiles[0], self.tiles[1], _p(dst),
                                dst.strides[0]))
        return dst


class Preproc:
    """Device-resident pre-processing chain [CLAHE] -> [remap | resize] -> [gray] in front of the extractor
    (include/orbx.h orbx_preproc_*).  maps = (map_x, map_y) float arrays of shape (n_maps, out_h, out_w) or (out_h, out_w)."""

    def __init__(self, src_w, src_h, channels=1, rgb=True, maps=None, out_size=None, clahe=None, max_batch=2, device=0):
        prm = _PreprocParams()
        prm.src_w, prm.src_h, prm.channels, prm.rgb_order = src_w, src_h, channels, int(bool(rgb))
        keep = []
        if maps is not None:
            mx = np.ascontiguousarray(maps[0], np.float32)
            my = np.ascontiguousarray(maps[1], np.float32)
            if mx.ndim == 2:
                mx, my = mx[None], my[None]
            prm.n_maps, prm.out_h, prm.out_w = mx.shape
            prm.map_x, prm.map_y, prm.map_stride = mx.ctypes.data, my.ctypes.data, mx.shape[2]
            keep = [mx, my]
        elif out_size is not None:
            prm.out_w, prm.out_h = out_size
        if clahe is not None:
            prm.clahe_clip_limit, (prm.clahe_tiles_x, prm.clahe_tiles_y) = float(clahe[0]), clahe[1]
        h = C.c_void_p()
        _check(lib().orbx_preproc_create(C.byref(prm), max_batch, device, C.byref(h)))
        del keep
        self._h = h
        ow, oh = C.c_int(), C.c_int()
        _check(lib().orbx_preproc_output_size(self._h, C.byref(ow), C.byref(oh)))
        self.out_w, self.out_h, self.channels, self.src_w, self.src_h = ow.value, oh.value, channels, src_w, src_h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orbx_preproc_destroy(self._h)
            self._h = None

    def run(self, frame, map_index=0):
        frame = np.ascontiguousarray(frame, np.uint8)
        dst = np.zeros((self.out_h, self.out_w), np.uint8)
        _check(lib().orbx_preproc_run(self._h, _p(frame), frame.strides[0], map_index, _p(dst), dst.strides[0]))
        return dst

    def run_device(self, d_frames_ptr, n_frames, row_pitch, image_pitch):
        """-> (device pointer, w, h, row_pitch, image_pitch) of the pre-processed gray frames (owned by the handle)."""
        out, w, h = C.c_void_p(), C.c_int(), C.c_int()
        rp, ip = C.c_ssize_t(), C.c_ssize_t()
        _check(lib().orbx_preproc_run_device(self._h, C.c_void_p(d_frames_ptr), n_frames, row_pitch, image_pitch, C.byref(out),
                                             C.byref(w), C.byref(h), C.byref(rp), C.byref(ip)))
        return out.value, w.value, h.value, rp.value, ip.value


class ORBVocabulary:
    """ORBVocabulary (DBoW2::TemplatedVocabulary<FORB>) on the device: loadFromTextFile (src/System.cc:131) or the file's
    columns (parent, is_leaf, descriptor, weight per node); transform() = Frame::ComputeBoW (src/Frame.cc:846-851)."""

    def __init__(self, k=None, L=None, parent=None, is_leaf=None, desc=None, weight=None, scoring=0, weighting=0, path=None,
                 device=0):
        h = C.c_void_p()
        if path is not None:
            _check(lib().orbx_vocabulary_load_text(device, path.encode(), C.byref(h)))
        else:
            parent = np.ascontiguousarray(parent, np.int32)
            leaf = np.ascontiguousarray(is_leaf, np.uint8)
            d = np.ascontiguousarray(desc, np.uint8)
            w = np.ascontiguousarray(weight, np.float64)
            _check(lib().orbx_vocabulary_create(device, k, L, scoring, weighting, len(parent), _p(parent), _p(leaf), _p(d), _p(w),
                                                C.byref(h)))
        self._h = h
        info = np.zeros(6, np.int32)
        _check(lib().orbx_vocabulary_info(self._h, _p(info)))
        self.k, self.L, self.n_nodes, self.n_words, self.scoring, self.weighting = (int(x) for x in info)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orbx_vocabulary_destroy(self._h)
            self._h = None

    def transform(self, desc, levelsup=4):
        """-> (word_ids, values), (node_ids, node_start, feature_idx): mBowVec and mFeatVec (CSR)."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        words, values = np.zeros(n, np.uint32), np.zeros(n, np.float64)
        nodes, start, feats = np.zeros(n, np.uint32), np.zeros(n + 1, np.int32), np.zeros(n, np.uint32)
        nw, nn = C.c_int(), C.c_int()
        nf = _check(lib().orbx_bow_transform(self._h, _p(d), n, levelsup, _p(words), _p(values), C.byref(nw), _p(nodes), _p(start),
                                             _p(feats), C.byref(nn)))
        return (words[:nw.value].copy(), values[:nw.value].copy()), (nodes[:nn.value].copy(), start[:nn.value + 1].copy(), feats[:nf].copy())

    def transform_batch(self, extractor, levelsup=4):
        """ComputeBoW for every image of the extractor's last extraction, on its stream (results stay on the device)."""
        _check(lib().orbx_bow_transform_batch(extractor._h, self._h, levelsup))

    @staticmethod
    def download(extractor, image):
        cap = extractor.capacity
        words, values = np.zeros(cap, np.uint32), np.zeros(cap, np.float64)
        nodes, start, feats = np.zeros(cap, np.uint32), np.zeros(cap + 1, np.int32), np.zeros(cap, np.uint32)
        nw, nn = C.c_int(), C.c_int()
        nf = _check(lib().orbx_bow_download(extractor._h, image, _p(words), _p(values), C.byref(nw), _p(nodes), _p(start), _p(feats),
                                            C.byref(nn), cap))
        return (words[:nw.value].copy(), values[:nw.value].copy()), (nodes[:nn.value].copy(), start[:nn.value + 1].copy(), feats[:nf].copy())


def SearchByBoW(kf_fv, kf_kps, kf_desc, kf_valid, f_fv, f_kps, f_desc, n_left_f=-1, nnratio=0.7, check_ori=True, device=0):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:230-404): feature vectors as the CSR triples
    ORBVocabulary.transform returns; -> (nmatches, match[n_f] = keyframe feature index or -1)."""
    kn, ks, kfi = (np.ascontiguousarray(kf_fv[0], np.uint32), np.ascontiguousarray(kf_fv[1], np.int32), np.ascontiguousarray(kf_fv[2], np.uint32))
    fn, fs, ffi = (np.ascontiguousarray(f_fv[0], np.uint32), np.ascontiguousarray(f_fv[1], np.int32), np.ascontiguousarray(f_fv[2], np.uint32))
    kk, fk = np.ascontiguousarray(kf_kps, KP_DTYPE), np.ascontiguousarray(f_kps, KP_DTYPE)
    kd, fd = np.ascontiguousarray(kf_desc, np.uint8), np.ascontiguousarray(f_desc, np.uint8)
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    match = np.zeros(len(fk), np.int32)
    n = _check(lib().orbx_search_by_bow(device, _p(kn), _p(ks), _p(kfi), len(kn), _p(kk), _p(kd), _p(kv), len(kk), _p(fn), _p(fs),
                                        _p(ffi), len(fn), _p(fk), _p(fd), len(fk), int(n_left_f), float(nnratio), int(bool(check_ori)),
                                        _p(match)))
    return n, match


def SearchByBoWBatch(ex, first_image, kf_fvs, kf_kps, kf_descs, kf_valids, n_left_f=-1, nnratio=0.7, check_ori=True):
    """SearchByBoW(KeyFrame, Frame) for the frames of ex's last extraction batch in one call (orbx_search_by_bow_batch): the frames'
    keypoints, descriptors and feature vectors (ORBVocabulary.transform_batch) stay on the device; the key frame of pair f comes
    as kf_fvs[f] = (node_ids, node_start, feature_idx), kf_kps[f], kf_descs[f], kf_valids[f].
    Returns (n_matches [F], list of match arrays [n_f])."""
    F = len(kf_fvs)
    nn = np.array([len(fv[0]) for fv in kf_fvs], np.int32)
    nk = np.array([len(k) for k in kf_kps], np.int32)
    ns, ks = max(int(nn.max()) if F else 0, 1), max(int(nk.max()) if F else 0, 1)
    ids, st = np.zeros((F, ns), np.uint32), np.zeros((F, ns + 1), np.int32)
    fi, K = np.zeros((F, ks), np.uint32), np.zeros((F, ks), KP_DTYPE)
    D, V = np.zeros((F, ks, 32), np.uint8), np.zeros((F, ks), np.uint8)
    for f in range(F):
        ids[f, :nn[f]] = kf_fvs[f][0]
        st[f, :nn[f] + 1] = kf_fvs[f][1]
        fi[f, :len(kf_fvs[f][2])] = kf_fvs[f][2]
        K[f, :nk[f]] = kf_kps[f]
        D[f, :nk[f]] = np.asarray(kf_descs[f], np.uint8).reshape(-1, 32)
        V[f, :nk[f]] = kf_valids[f]
    cap = ex.capacity
    match = np.full((F, cap), -1, np.int32)
    nm = np.zeros(F, np.int32)
    _check(lib().orbx_search_by_bow_batch(ex._h, int(first_image), F, _p(ids), _p(st), _p(nn), ns, _p(fi), _p(K), _p(D), _p(V), _p(nk), ks,
                                          int(n_left_f), float(nnratio), int(bool(check_ori)), _p(match), _p(nm)))
    return nm, match


def SearchByBoWKeyFrames(fv1, kps1, desc1, valid1, fv2, kps2, desc2, valid2, nnratio=0.75, check_ori=True, device=0):
    """ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) (src/ORBmatcher.cc:766-884) ->
    (nmatches, matches12[n1] = feature of pKF2 or -1)."""
    n1n, s1, f1 = (np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32))
    n2n, s2, f2 = (np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32))
    k1, k2 = np.ascontiguousarray(kps1, KP_DTYPE), np.ascontiguousarray(kps2, KP_DTYPE)
    d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
    v1, v2 = np.ascontiguousarray(valid1, np.uint8), np.ascontiguousarray(valid2, np.uint8)
    m = np.full(len(k1), -1, np.int32)
    n = _check(lib().orbx_search_by_bow_keyframes(device, _p(n1n), _p(s1), _p(f1), len(n1n), _p(k1), _p(d1), _p(v1), len(k1), _p(n2n),
                                                  _p(s2), _p(f2), len(n2n), _p(k2), _p(d2), _p(v2), len(k2), float(nnratio),
                                                  int(bool(check_ori)), _p(m)))
    return n, m


def UndistortKeyPoints(kps, K, dist, device=0):
    """Frame::UndistortKeyPoints (src/Frame.cc:853-885): mvKeysUn from mvKeys; K = (fx, fy, cx, cy), dist = mDistCoef."""
    k = np.ascontiguousarray(kps, KP_DTYPE)
    K = np.ascontiguousarray(K, np.float32)
    d = np.ascontiguousarray(dist, np.float32).ravel()
    out = np.zeros(len(k), KP_DTYPE)
    _check(lib().orbx_undistort_keypoints(device, _p(k), len(k), _p(K), _p(d) if len(d) else None, len(d), _p(out)))
    return out


def ComputeImageBounds(cols, rows, K, dist, device=0):
    """Frame::ComputeImageBounds (src/Frame.cc:887-919) -> (mnMinX, mnMinY, mnMaxX, mnMaxY)."""
    K = np.ascontiguousarray(K, np.float32)
    d = np.ascontiguousarray(dist, np.float32).ravel()
    b = np.zeros(4, np.float32)
    _check(lib().orbx_compute_image_bounds(device, int(cols), int(rows), _p(K), _p(d) if len(d) else None, len(d), _p(b)))
    return b


def GetFeaturesInArea(kpsUn, bounds, queries, device=0, return_grid=False):
    """Frame::AssignFeaturesToGrid + GetFeaturesInArea (src/Frame.cc:520-547,765-844) for a batch of queries
    (rows of x, y, r, minLevel, maxLevel).  Returns a list of index arrays (reference order), optionally the
    grid as (cell_start[3073], items[n])."""
    k = np.ascontiguousarray(kpsUn, KP_DTYPE)
    q = np.ascontiguousarray(queries, np.float32).reshape(-1, 5)
    offs = np.zeros(len(q) + 1, np.int32)
    cs = np.zeros(64 * 48 + 1, np.int32)
    items = np.zeros(max(len(k), 1), np.int32)
    cap = max(1, len(k)) * max(1, len(q))
    idx = np.zeros(min(cap, 1 << 26), np.int32)
    tot = _check(lib().orbx_features_in_area(device, _p(k), len(k), bounds[0], bounds[1], bounds[2], bounds[3], _p(q),
                                             len(q), _p(offs), _p(idx), len(idx), _p(cs), _p(items)))
    res = [idx[offs[i]:offs[i + 1]].copy() for i in range(len(q))]
    assert tot == offs[-1]
    return (res, cs, items[:len(k)]) if return_grid else res


class ORBmatcher:
    """Mirror of the hot-path part of ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:36-101)."""
    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.mfNNratio, self.mbCheckOrientation, self.device = float(nnratio), bool(checkOri), device

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(32)
        b = np.ascontiguousarray(b, np.uint8).reshape(32)
        return lib().orbx_hamming256(_p(a), _p(b))

    def SearchByProjection(self, kpsUn, desc, uRight, bounds, scaleFactors, mapPoints, occupied, th=1.0,
                           bFarPoints=False, thFarPoints=50.0):
        """src/ORBmatcher.cc:41-221 (local map points -> frame, pinhole case).  mapPoints: MP_DTYPE records.
        Returns (nmatches, match[n] = map point index or -1, updated occupied[n])."""
        k = np.ascontiguousarray(kpsUn, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        mp = np.ascontiguousarray(mapPoints, MP_DTYPE)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        occ = np.ascontiguousarray(occupied, np.uint8).copy()
        ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        match = np.full(len(k), -1, np.int32)
        n = _check(lib().orbx_search_by_projection(
            self.device, _p(k), _p(d), None if ur is None else _p(ur), len(k), bounds[0], bounds[1], bounds[2],
            bounds[3], _p(sf), len(sf), _p(mp), len(mp), float(th), int(bFarPoints), float(thFarPoints),
            self.mfNNratio, _p(occ), _p(match)))
        return n, match, occ

    def SearchByProjectionFrame(self, kpsUn, desc, uRight, bounds, projectedPoints, occupied):
        """Matching part of SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1594-1806,
        pinhole case); projectedPoints: PP_DTYPE records computed by the caller's pose / camera projection.
        Returns (nmatches, match[n] = LastFrame point index or -1, updated occupied[n])."""
        k = np.ascontiguousarray(kpsUn, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        pp = np.ascontiguousarray(projectedPoints, PP_DTYPE)
        occ = np.ascontiguousarray(occupied, np.uint8).copy()
        ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        match = np.full(len(k), -1, np.int32)
        n = _check(lib().orbx_search_by_projection_frame(
            self.device, _p(k), _p(d), None if ur is None else _p(ur), len(k), bounds[0], bounds[1], bounds[2],
            bounds[3], _p(pp), len(pp), int(self.mbCheckOrientation), _p(occ), _p(match)))
        return n, match, occ

    def SearchForInitializationBatch(self, ex, first_image, kps1, desc1, bounds2, vbPrevMatched, windowSize=10):
        """SearchForInitialization for the frames of ex's last extraction batch in one call (orbx_search_for_initialization_batch):
        pair f matches F1 = (kps1[f], desc1[f]) -- lists of per-pair arrays -- against image first_image + f of the batch.
        Returns (n_matches [F], list of vnMatches12, list of updated vbPrevMatched)."""
        F = len(kps1)
        n1 = np.array([len(k) for k in kps1], np.int32)
        stride = max(int(n1.max()) if F else 0, 1)
        K = np.zeros((F, stride), KP_DTYPE)
        D = np.zeros((F, stride, 32), np.uint8)
        P = np.zeros((F, stride, 2), np.float32)
        for f in range(F):
            K[f, :n1[f]] = kps1[f]
            D[f, :n1[f]] = np.asarray(desc1[f], np.uint8).reshape(-1, 32)
            P[f, :n1[f]] = np.asarray(vbPrevMatched[f], np.float32).reshape(-1, 2)
        M = np.full((F, stride), -1, np.int32)
        nm = np.zeros(F, np.int32)
        _check(lib().orbx_search_for_initialization_batch(
            ex._h, int(first_image), F, _p(K), _p(D), _p(n1), stride, bounds2[0], bounds2[1], bounds2[2], bounds2[3], _p(P), _p(M),
            int(windowSize), self.mfNNratio, int(self.mbCheckOrientation), _p(nm)))
        return nm, [M[f, :n1[f]].copy() for f in range(F)], [P[f, :n1[f]].copy() for f in range(F)]

    def SearchByProjectionFrameBatch(self, ex, first_image, n_frames, bounds, points, n_points, occupied=None, stereo_pair0=-1):
        """SearchByProjectionFrame on the frames of ex's last extraction batch in one call (include/orbx.h:
        orbx_search_by_projection_frame_batch): points [n_frames][stride] PP_DTYPE, n_points [n_frames].
        Returns (n_matches [n_frames], match [n_frames][cap], occupied [n_frames][cap])."""
        pp = np.ascontiguousarray(points, PP_DTYPE).reshape(n_frames, -1)
        npts = np.ascontiguousarray(n_points, np.int32)
        cap = ex.capacity
        occ_in = None if occupied is None else np.ascontiguousarray(occupied, np.uint8).reshape(n_frames, cap)
        occ = np.zeros((n_frames, cap), np.uint8)
        match = np.full((n_frames, cap), -1, np.int32)
        nm = np.zeros(n_frames, np.int32)
        _check(lib().orbx_search_by_projection_frame_batch(
            ex._h, int(first_image), int(n_frames), bounds[0], bounds[1], bounds[2], bounds[3], _p(pp), _p(npts), pp.shape[1],
            int(self.mbCheckOrientation), int(stereo_pair0), None if occ_in is None else _p(occ_in), _p(occ), _p(match), _p(nm)))
        return nm, match, occ

    def SearchByProjectionFrameBatchDevice(self, ex, first_image, n_frames, bounds, occupied=None, stereo_pair0=-1):
        """SearchByProjectionFrame on the frames of ex's last extraction batch with the views made ON THE DEVICE by
        project_last_frames (orbx_search_by_projection_frame_batch with points = NULL)."""
        npts = np.ascontiguousarray(ex._lf_n[:n_frames], np.int32)
        cap = ex.capacity
        occ_in = None if occupied is None else np.ascontiguousarray(occupied, np.uint8).reshape(n_frames, cap)
        occ = np.zeros((n_frames, cap), np.uint8)
        match = np.full((n_frames, cap), -1, np.int32)
        nm = np.zeros(n_frames, np.int32)
        _check(lib().orbx_search_by_projection_frame_batch(
            ex._h, int(first_image), int(n_frames), bounds[0], bounds[1], bounds[2], bounds[3], None, _p(npts), ex._lf_stride,
            int(self.mbCheckOrientation), int(stereo_pair0), None if occ_in is None else _p(occ_in), _p(occ), _p(match), _p(nm)))
        return nm, match, occ

    def SearchByProjectionBatch(self, ex, first_image, n_frames, bounds, mapPoints, n_map_points, occupied=None, th=1.0,
                                bFarPoints=False, thFarPoints=50.0, stereo_pair0=-1):
        """SearchByProjection (local map points) on the frames of ex's last extraction batch in one call
        (orbx_search_by_projection_batch): mapPoints [n_frames][stride] MP_DTYPE.  Returns (n_matches, match, occupied)."""
        mp = np.ascontiguousarray(mapPoints, MP_DTYPE).reshape(n_frames, -1)
        npts = np.ascontiguousarray(n_map_points, np.int32)
        cap = ex.capacity
        occ_in = None if occupied is None else np.ascontiguousarray(occupied, np.uint8).reshape(n_frames, cap)
        occ = np.zeros((n_frames, cap), np.uint8)
        match = np.full((n_frames, cap), -1, np.int32)
        nm = np.zeros(n_frames, np.int32)
        _check(lib().orbx_search_by_projection_batch(
            ex._h, int(first_image), int(n_frames), bounds[0], bounds[1], bounds[2], bounds[3], _p(mp), _p(npts), mp.shape[1],
            float(th), int(bFarPoints), float(thFarPoints), self.mfNNratio, int(stereo_pair0),
            None if occ_in is None else _p(occ_in), _p(occ), _p(match), _p(nm)))
        return nm, match, occ

    def SearchByProjectionBatchDevice(self, ex, first_image, n_frames, bounds, occupied=None, th=1.0, bFarPoints=False,
                                      thFarPoints=50.0, stereo_pair0=-1):
        """SearchByProjection (local map points) on the frames of ex's last extraction batch with the views made ON THE DEVICE by
        project_map_points (orbx_search_by_projection_batch with map_points = NULL)."""
        n = ex._map_n
        npts = np.full(n_frames, n, np.int32)
        cap = ex.capacity
        occ_in = None if occupied is None else np.ascontiguousarray(occupied, np.uint8).reshape(n_frames, cap)
        occ = np.zeros((n_frames, cap), np.uint8)
        match = np.full((n_frames, cap), -1, np.int32)
        nm = np.zeros(n_frames, np.int32)
        _check(lib().orbx_search_by_projection_batch(
            ex._h, int(first_image), int(n_frames), bounds[0], bounds[1], bounds[2], bounds[3], None, _p(npts), n,
            float(th), int(bFarPoints), float(thFarPoints), self.mfNNratio, int(stereo_pair0),
            None if occ_in is None else _p(occ_in), _p(occ), _p(match), _p(nm)))
        return nm, match, occ

    def SearchByProjectionKeyFrame(self, kpsUn, desc, bounds, projectedPoints, occupied, ORBdist=100):
        """Matching part of the relocalisation matcher SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist)
        (src/ORBmatcher.cc:1808-1918); projectedPoints: PP_DTYPE records of pKF's map points after the caller's projection
        and gates; occupied[i2] <=> CurrentFrame.mvpMapPoints[i2] != NULL.
        Returns (nmatches, match[n] = point index or -1, updated occupied[n])."""
        k = np.ascontiguousarray(kpsUn, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        pp = np.ascontiguousarray(projectedPoints, PP_DTYPE)
        occ = np.ascontiguousarray(occupied, np.uint8).copy()
        match = np.full(len(k), -1, np.int32)
        n = _check(lib().orbx_search_by_projection_keyframe(
            self.device, _p(k), _p(d), len(k), bounds[0], bounds[1], bounds[2], bounds[3], _p(pp), len(pp), int(ORBdist),
            int(self.mbCheckOrientation), _p(occ), _p(match)))
        return n, match, occ

    def FuseSearch(self, kps, desc, uRight, bounds, invLevelSigma2, points, maxDist=TH_LOW):
        """The search of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight) (src/ORBmatcher.cc:1195-1256); points: FP_DTYPE records
        of the map points after the caller's projection and gates.  Returns (nFused, bestIdx[n_points] = keypoint index or -1,
        bestDist[n_points])."""
        k = np.ascontiguousarray(kps, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        pp = np.ascontiguousarray(points, FP_DTYPE)
        ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        isg = np.ascontiguousarray(invLevelSigma2, np.float32)
        bi, bd = np.full(len(pp), -1, np.int32), np.full(len(pp), 256, np.int32)
        n = _check(lib().orbx_fuse_search(self.device, _p(k), _p(d), None if ur is None else _p(ur), len(k), bounds[0], bounds[1],
                                          bounds[2], bounds[3], _p(isg), len(isg), _p(pp), len(pp), int(maxDist), _p(bi), _p(bd)))
        return n, bi, bd

    def SearchBySim3(self, kps1, desc1, bounds1, kps2, desc2, bounds2, points1in2, points2in1):
        """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1392-1592): pKF1's map points searched in pKF2 and pKF2's in pKF1 (FP_DTYPE
        records after the caller's Sim3 projections; one record per feature of the key frame they come from), TH_HIGH, no
        chi-square gate, then the agreement check.  Returns (nFound, vnMatch12[N1] = feature of pKF2 or -1)."""
        nl1 = int(np.max(np.ascontiguousarray(kps1, KP_DTYPE)["octave"], initial=0)) + 1
        nl2 = int(np.max(np.ascontiguousarray(kps2, KP_DTYPE)["octave"], initial=0)) + 1
        _, m1, _ = self.FuseSearch(kps2, desc2, None, bounds2, np.zeros(nl2, np.float32), points1in2, TH_HIGH)
        _, m2, _ = self.FuseSearch(kps1, desc1, None, bounds1, np.zeros(nl1, np.float32), points2in1, TH_HIGH)
        i1 = np.arange(len(m1))
        ok = (m1 >= 0) & (m1 < len(m2))
        ok[ok] = m2[m1[ok]] == i1[ok]
        out = np.where(ok, m1, -1).astype(np.int32)
        return int(ok.sum()), out

    def SearchForTriangulationRig(self, fv1, kps1, desc1, hasMapPoint1, nLeft1, fv2, kps2, desc2, hasMapPoint2, nLeft2, levelSigma2_1,
                                  levelSigma2_2, rig, bOnlyStereo=False, bCoarse=False):
        """ORBmatcher::SearchForTriangulation for two-camera key frames (src/ORBmatcher.cc:906-923,1007-1064): kps / desc hold
        mvKeys | mvKeysRight, rig = a TRI_RIG_DTYPE record.  Returns (nmatches, vMatches12[n1])."""
        n1n, s1, f1 = (np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32))
        n2n, s2, f2 = (np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32))
        k1, k2 = np.ascontiguousarray(kps1, KP_DTYPE), np.ascontiguousarray(kps2, KP_DTYPE)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        h1, h2 = np.ascontiguousarray(hasMapPoint1, np.uint8), np.ascontiguousarray(hasMapPoint2, np.uint8)
        g1, g2 = np.ascontiguousarray(levelSigma2_1, np.float32), np.ascontiguousarray(levelSigma2_2, np.float32)
        rg = np.ascontiguousarray(rig, TRI_RIG_DTYPE)
        m = np.full(len(k1), -1, np.int32)
        n = _check(lib().orbx_search_for_triangulation_rig(
            self.device, _p(n1n), _p(s1), _p(f1), len(n1n), _p(k1), _p(d1), _p(h1), int(nLeft1), len(k1), _p(n2n), _p(s2), _p(f2), len(n2n),
            _p(k2), _p(d2), _p(h2), int(nLeft2), len(k2), _p(g1), _p(g2), len(g1), _p(rg), int(bOnlyStereo), int(bCoarse),
            int(self.mbCheckOrientation), _p(m)))
        return n, m

    def SearchForTriangulation(self, fv1, kps1, desc1, hasMapPoint1, uRight1, fv2, kps2, desc2, hasMapPoint2, uRight2,
                               scaleFactors2, levelSigma2_2, ep, F12, bOnlyStereo=False, bCoarse=False):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:886-1106),
        single-camera key frames; fv = (node ids, node start, feature indices) as ORBVocabulary.transform returns them.
        Returns (nmatches, vMatchedPairs as an [nmatches, 2] array in ascending idx1, vMatches12[n1])."""
        n1n, s1, f1 = (np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32))
        n2n, s2, f2 = (np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32))
        k1, k2 = np.ascontiguousarray(kps1, KP_DTYPE), np.ascontiguousarray(kps2, KP_DTYPE)
        d1, d2 = np.ascontiguousarray(desc1, np.uint8), np.ascontiguousarray(desc2, np.uint8)
        h1, h2 = np.ascontiguousarray(hasMapPoint1, np.uint8), np.ascontiguousarray(hasMapPoint2, np.uint8)
        u1 = None if uRight1 is None else np.ascontiguousarray(uRight1, np.float32)
        u2 = None if uRight2 is None else np.ascontiguousarray(uRight2, np.float32)
        sf, sg = np.ascontiguousarray(scaleFactors2, np.float32), np.ascontiguousarray(levelSigma2_2, np.float32)
        epa = np.ascontiguousarray(ep, np.float32)
        Fa = None if F12 is None else np.ascontiguousarray(F12, np.float32).reshape(9)
        m = np.full(len(k1), -1, np.int32)
        n = _check(lib().orbx_search_for_triangulation(
            self.device, _p(n1n), _p(s1), _p(f1), len(n1n), _p(k1), _p(d1), _p(h1), None if u1 is None else _p(u1), len(k1),
            _p(n2n), _p(s2), _p(f2), len(n2n), _p(k2), _p(d2), _p(h2), None if u2 is None else _p(u2), len(k2), _p(sf), _p(sg),
            len(sf), _p(epa), None if Fa is None else _p(Fa), int(bOnlyStereo), int(bCoarse), int(self.mbCheckOrientation), _p(m)))
        idx1 = np.nonzero(m >= 0)[0]
        return n, np.stack([idx1, m[idx1]], axis=1).astype(np.int64), m

    def SearchByProjectionFisheye(self, kps, desc, n_left, bounds, scaleFactors, mapPoints, mapPointsRight, leftToRight,
                                  rightToLeft, occupied, th=1.0, bFarPoints=False, thFarPoints=50.0):
        """src/ORBmatcher.cc:41-221 with F.Nleft != -1: kps / desc = mvKeys then mvKeysRight (N = n_left + n_right rows),
        mapPointsRight = MPR_DTYPE records.  Returns (nmatches, match[N], updated occupied[N])."""
        k = np.ascontiguousarray(kps, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        mp, mr = np.ascontiguousarray(mapPoints, MP_DTYPE), np.ascontiguousarray(mapPointsRight, MPR_DTYPE)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        l2r, r2l = np.ascontiguousarray(leftToRight, np.int32), np.ascontiguousarray(rightToLeft, np.int32)
        occ = np.ascontiguousarray(occupied, np.uint8).copy()
        match = np.full(len(k), -1, np.int32)
        n = _check(lib().orbx_search_by_projection_fisheye(
            self.device, _p(k), _p(d), int(n_left), len(k) - int(n_left), bounds[0], bounds[1], bounds[2], bounds[3], _p(sf),
            len(sf), _p(mp), _p(mr), len(mp), float(th), int(bFarPoints), float(thFarPoints), self.mfNNratio, _p(l2r), _p(r2l),
            _p(occ), _p(match)))
        return n, match, occ

    def SearchByProjectionFrameFisheye(self, kps, desc, n_left, bounds, projectedPoints, uvRight, occupied):
        """src/ORBmatcher.cc:1594-1806 with CurrentFrame.Nleft != -1; uvRight[i] = projection of point i into the right
        camera.  Returns (nmatches, match[N], updated occupied[N])."""
        k = np.ascontiguousarray(kps, KP_DTYPE)
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        pp = np.ascontiguousarray(projectedPoints, PP_DTYPE)
        uv = np.ascontiguousarray(uvRight, np.float32).reshape(-1, 2)
        occ = np.ascontiguousarray(occupied, np.uint8).copy()
        match = np.full(len(k), -1, np.int32)
        n = _check(lib().orbx_search_by_projection_frame_fisheye(
            self.device, _p(k), _p(d), int(n_left), len(k) - int(n_left), bounds[0], bounds[1], bounds[2], bounds[3], _p(pp),
            _p(uv), len(pp), int(self.mbCheckOrientation), _p(occ), _p(match)))
        return n, match, occ

    def SearchByProjectionFisheyeBatch(self, ex, first_left, first_right, n_frames, bounds, mapPoints, mapPointsRight, n_points,
                                       leftToRight, rightToLeft, occupied=None, th=1.0, bFarPoints=False, thFarPoints=50.0):
        """SearchByProjectionFisheye on the two-camera frames of ex's last extraction batch in one call
        (orbx_search_by_projection_fisheye_batch): frame f = images first_left + f / first_right + f; mapPoints / mapPointsRight
        [n_frames][stride], leftToRight / rightToLeft [n_frames][cap], occupied [n_frames][2 cap] (rows [left | right]).
        Returns (n_matches [n_frames], match [n_frames][2 cap], occupied [n_frames][2 cap])."""
        cap = ex.capacity
        mp = np.ascontiguousarray(mapPoints, MP_DTYPE).reshape(n_frames, -1)
        mr = np.ascontiguousarray(mapPointsRight, MPR_DTYPE).reshape(n_frames, -1)
        npts = np.ascontiguousarray(n_points, np.int32)
        l2r = np.ascontiguousarray(leftToRight, np.int32).reshape(n_frames, cap)
        r2l = np.ascontiguousarray(rightToLeft, np.int32).reshape(n_frames, cap)
        occ_in = None if occupied is None else np.ascontiguousarray(occupied, np.uint8).reshape(n_frames, 2 * cap)
        occ, match, nm = np.zeros((n_frames, 2 * cap), np.uint8), np.full((n_frames, 2 * cap), -1, np.int32), np.zeros(n_frames, np.int32)
        _check(lib().orbx_search_by_projection_fisheye_batch(
            ex._h, int(first_left), int(first_right), int(n_frames), bounds[0], bounds[1], bounds[2], bounds[3], _p(mp), _p(mr), _p(npts),
            mp.shape[1], float(th), int(bFarPoints), float(thFarPoints), self.mfNNratio, _p(l2r), _p(r2l),
            None if occ_in is None else _p(occ_in), _p(occ), _p(match), _p(nm)))
        return nm, match, occ

    def SearchByProjectionFisheyeBatchDevice(self, ex, first_left, first_right, n_frames, bounds, leftToRight, rightToLeft, occupied=None,
                                             th=1.0, bFarPoints=False, thFarPoints=50.0):
        """SearchByProjectionFisheye on the two-camera frames of ex's last extraction batch with both cameras' views made ON THE
        DEVICE by project_map_points_fisheye (orbx_search_by_projection_fisheye_batch with map_points = map_points_right = NULL)."""
        cap, n = ex.capacity, ex._map_n
        npts = np.full(n_frames, n, np.int32)
        l2r = np.ascontiguousarray(leftToRight, np.int32).reshape(n_frames, cap)
        r2l = np.ascontiguousarray(rightToLeft, np.int32).reshape(n_frames, cap)
        occ_in = None if occupied is None else np.ascontiguousarray(occupied, np.uint8).reshape(n_frames, 2 * cap)
        occ, match, nm = np.zeros((n_frames, 2 * cap), np.uint8), np.full((n_frames, 2 * cap), -1, np.int32), np.zeros(n_frames, np.int32)
        _check(lib().orbx_search_by_projection_fisheye_batch(
            ex._h, int(first_left), int(first_right), int(n_frames), bounds[0], bounds[1], bounds[2], bounds[3], None, None, _p(npts), n,
            float(th), int(bFarPoints), float(thFarPoints), self.mfNNratio, _p(l2r), _p(r2l),
            None if occ_in is None else _p(occ_in), _p(occ), _p(match), _p(nm)))
        return nm, match, occ

    def SearchByProjectionFrameFisheyeBatch(self, ex, first_left, first_right, n_frames, bounds, points, uvRight, n_points, occupied=None):
        """SearchByProjectionFrameFisheye on the two-camera frames of ex's last extraction batch in one call
        (orbx_search_by_projection_frame_fisheye_batch): points [n_frames][stride] PP_DTYPE, uvRight [n_frames][stride][2].
        Returns (n_matches, match [n_frames][2 cap], occupied [n_frames][2 cap])."""
        cap = ex.capacity
        pp = np.ascontiguousarray(points, PP_DTYPE).reshape(n_frames, -1)
        uv = np.ascontiguousarray(uvRight, np.float32).reshape(n_frames, pp.shape[1], 2)
        npts = np.ascontiguousarray(n_points, np.int32)
        occ_in = None if occupied is None else np.ascontiguousarray(occupied, np.uint8).reshape(n_frames, 2 * cap)
        occ, match, nm = np.zeros((n_frames, 2 * cap), np.uint8), np.full((n_frames, 2 * cap), -1, np.int32), np.zeros(n_frames, np.int32)
        _check(lib().orbx_search_by_projection_frame_fisheye_batch(
            ex._h, int(first_left), int(first_right), int(n_frames), bounds[0], bounds[1], bounds[2], bounds[3], _p(pp), _p(uv), _p(npts),
            pp.shape[1], int(self.mbCheckOrientation), None if occ_in is None else _p(occ_in), _p(occ), _p(match), _p(nm)))
        return nm, match, occ

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, bounds2, vbPrevMatched, windowSize=10):
        """src/ORBmatcher.cc:618-764.  kps = mvKeysUn of F1 / F2, bounds2 = (mnMinX, mnMinY, mnMaxX, mnMaxY)
        of F2.  Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        k1 = np.ascontiguousarray(kps1, KP_DTYPE)
        k2 = np.ascontiguousarray(kps2, KP_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2).copy()
        assert len(prev) == len(k1)
        m12 = np.full(len(k1), -1, np.int32)
        n = _check(lib().orbx_search_for_initialization(
            self.device, _p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), bounds2[0], bounds2[1], bounds2[2],
            bounds2[3], _p(prev), _p(m12), int(windowSize), self.mfNNratio, int(self.mbCheckOrientation)))
        return n, m12, prev
