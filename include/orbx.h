/* orbx.h — C ABI of the MI355X-native ORB front-end (liborbx.so).
 *
 * Drop-in boundary for the reference's ORB hot path (hellovuong/ORB_SLAM3_FAST).  Every entry point names
 * the reference interface it replaces (file:line relative to the reference root).  Plain pointers and
 * sizes only: no OpenCV, no torch types.  The C++ mirror classes ORB_SLAM3::ORBextractor / ORBmatcher in
 * orb_slam3_fast_amd/csrc/ORBextractor.h / ORBmatcher.h sit on top of exactly these calls; INTEGRATION.md
 * shows the binding a maintainer of the reference would add.
 *
 * All compute runs in hand-written HIP kernels for gfx950.  There is no CPU fallback: without a HIP device
 * every compute call returns ORBX_E_NODEVICE / ORBX_E_HIP.
 */
#ifndef ORBX_H_
#define ORBX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_MAX_LEVELS 12

/* Error codes.  ORBX_E_EMPTY keeps ORBextractor::operator()'s "-1 on empty image"
 * (src/ORBextractor.cc:1021). */
#define ORBX_OK 0
#define ORBX_E_EMPTY (-1)
#define ORBX_E_BADARG (-2)
#define ORBX_E_CAPACITY (-3)
#define ORBX_E_HIP (-4)
#define ORBX_E_NODEVICE (-5)
#define ORBX_E_UNSUPPORTED (-6) /* image too small for the pyramid or aspect (Q11) */
#define ORBX_E_TIMEOUT (-7)     /* orbx_comm_wait: a collective did not complete in time (a rank missing / out of order) */

/* Layout-identical to cv::KeyPoint (28 bytes): pt.x pt.y size angle response octave class_id
 * (SURVEY 8a row a13). */
typedef struct orbx_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} orbx_keypoint;

/* Constructor arguments of ORBextractor (include/ORBextractor.h:53-57, src/ORBextractor.cc:408-417). */
typedef struct orbx_params {
  int32_t nfeatures;
  float scale_factor;
  int32_t nlevels;
  int32_t ini_th_fast;
  int32_t min_th_fast;
} orbx_params;

typedef struct orbx_extractor orbx_extractor;

const char* orbx_last_error(void);
int orbx_device_count(void);
int orbx_abi_version(void);

/* ---- ORBextractor ---------------------------------------------------------------------------------- */

/* Replaces ORBextractor::ORBextractor (src/ORBextractor.cc:408-469).  One handle = one extractor instance
 * bound to `device` with its own HIP stream; it owns pyramids and result buffers for up to max_batch images
 * of up to max_width x max_height per call.  Handles are not re-entrant (like the reference instance, which
 * mutates mvImagePyramid); distinct handles may be used concurrently from different threads
 * (src/Frame.cc:200-203). */
int orbx_extractor_create(const orbx_params* p, int max_width, int max_height, int max_batch, int device,
                          orbx_extractor** out);
void orbx_extractor_destroy(orbx_extractor* ex);

/* Replaces GetScaleFactors/GetInverseScaleFactors/GetScaleSigmaSquares/GetInverseScaleSigmaSquares
 * (include/ORBextractor.h:65-83) plus the private mnFeaturesPerLevel / umax tables.  Any pointer may be
 * NULL.  Arrays hold nlevels entries, umax16 holds 16. */
int orbx_get_tables(const orbx_extractor* ex, float* scale, float* inv_scale, float* sigma2,
                    float* inv_sigma2, int32_t* nfeatures_per_level, int32_t* umax16);

/* Which OpenCV the reference build links decides the descriptor bits: cv::GaussianBlur(7x7, sigma 2) of ORBextractor::operator()
 * (src/ORBextractor.cc:1074-1076) runs on 8-bit fixed-point taps that changed between releases -- {18,34,49,55,49,34,18} / 256 in
 * OpenCV 4.0 .. 4.5.0 (the README's "tested with 4.4.0", README.md:101; the taps sum to 257, results saturate at 255) and
 * {18,34,48,56,48,34,18} / 256 from 4.5.1 on (CMakeLists.txt:38-41 asks for "> 4.4"; what distributions ship).
 * opencv_version = 451 (the default), 440, 44016 or 44032; anything else is ORBX_E_BADARG.  Applies to every later extraction of
 * the handle (k_describe's per-keypoint blur and the blurred levels of orbx_pyramid_level).  OpenCV 3.x's float filter is not
 * modelled.
 *   440   : the 257-sum taps through the SCALAR ufixedpoint path, every column rounded once.
 *   44016 / 44032 : the same taps as a build whose vertical pass runs the 16-lane (SSE2 / NEON baseline) or 32-lane (AVX2
 *           dispatch) vector body of smooth.simd.hpp: that body re-biases its 8.8 rows by -32768 and gives the bias back as the
 *           constant 128 << 16, which is 32768 short when the taps sum to 257 -- exactly the rounding half, so columns
 *           [0, (w / lanes) * lanes) of every level FLOOR and only the scalar tail behind them rounds (flat 100 -> 100 in the
 *           body, 101 in the tail; oracle/orb_oracle.cpp gaussian_blur7, tests/test_tables.py).  A reference built against
 *           OpenCV 4.4.0 on x86-64 is expected to behave as 44032.
 * STATUS: all settings follow this repo's restatement of OpenCV's fixed-point path (oracle/orb_oracle.cpp gaussian_blur7), which
 * is not pinned against any real OpenCV build (none exists in this environment); tools/gen_golden_opencv.py with opencv-python
 * 4.4.0 would settle which of 440 / 44016 / 44032 a given build is -- switching is this one call. */
int orbx_set_opencv_compat(orbx_extractor* ex, int opencv_version);

/* Replaces ORBextractor::operator() (src/ORBextractor.cc:1015-1106) for ONE host image (CV_8UC1, `stride`
 * bytes per row).  lap0/lap1 = vLappingArea.  Writes *n_out keypoints (serial-order slots: mono from the
 * front, lapping from the back) and n_out x 32 descriptor bytes.  Returns monoIndex (>= 0), ORBX_E_EMPTY for
 * an empty image, or another negative error.  cap = capacity of kps / desc rows: the handle's own row count
 * (orbx_batch_results_device's *capacity = nfeatures + 36 per level) always suffices; the reference returns at most
 * nfeatures + 3 per level for the usual quotas, but a level whose quota is below 4 x its initial quadtree roots can keep up
 * to 4 * nIni nodes (src/ORBextractor.cc:575-601), so nfeatures + 3 * nlevels is NOT a bound for tiny nfeatures.  ORBX_E_CAPACITY
 * when the result does not fit.  kps / desc may be NULL: the results stay in the handle's result block (orbx_host_results). */
int orbx_extract(orbx_extractor* ex, const uint8_t* img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                 orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Both eyes of ONE stereo frame through one batched pipeline and one synchronisation: replaces the two threaded
 * ExtractORB calls of the stereo Frame constructor (src/Frame.cc:200-203, 549-560) and, when bf > 0, the
 * ComputeStereoMatches that follows them (:921-1084; b = baseline, maxD = bf / b).  The handle needs max_batch >= 2; the
 * left eye becomes image 0 and the right eye image 1 of the extraction (orbx_pyramid_level, orbx_stereo_match_batch with
 * left == right handle, first_left 0, first_right 1).  Outputs as orbx_extract for each eye (n_* keypoints, mono_* =
 * monoIndex); uright / depth (cap_left floats each, -1 = no match) are ignored when bf <= 0.  COMPATIBILITY (since round 5): bf > 0
 * runs the stereo association also when uright / depth are NULL -- the results then wait in the result block
 * (orbx_host_results) --; pass bf = 0 to skip it.  Any of the output ARRAYS
 * (kps_*, desc_*, uright, depth) may be NULL: the results then stay in the handle's page-locked result block, where
 * orbx_host_results hands them out in place (one copy less per frame for a caller that converts them anyway).  Arrays that ARE
 * passed cost nothing at the end of the call: keypoints and descriptors of both eyes reach the block two launches before the frame
 * ends and are copied into the caller's memory while the stereo association still runs (1280x720, 1500 features: 0.183 - 0.20 ms per
 * call either way, profiles/r5c_latency_ab.txt).
 * Returns ORBX_OK, ORBX_E_EMPTY for an empty image, or another negative error. */
int orbx_extract_stereo(orbx_extractor* ex, const uint8_t* img_left, const uint8_t* img_right, int w, int h,
                        ptrdiff_t stride_left, ptrdiff_t stride_right, const int32_t lap_left[2],
                        const int32_t lap_right[2], orbx_keypoint* kps_left, uint8_t* desc_left, int cap_left,
                        int* n_left, int* mono_left, orbx_keypoint* kps_right, uint8_t* desc_right, int cap_right,
                        int* n_right, int* mono_right, float bf, float b, float* uright, float* depth);

/* Results of the last orbx_extract / orbx_extract_stereo call IN PLACE.  Those entries gather everything into one page-locked
 * block owned by the handle (one kernel, one synchronisation) and then copy into the caller's arrays; this accessor exposes
 * the block itself: n keypoints (28-byte cv::KeyPoint records, src/ORBextractor.cc:1053-1104 order) and n x 32 descriptor
 * bytes of image 0 (left eye) or 1 (right eye), monoIndex, and -- image 0 after a call with bf > 0 -- mvuRight / mvDepth
 * (src/Frame.cc:921-1084), n floats each.  Any output pointer may be NULL.  Valid until the next call on the handle. */
int orbx_host_results(const orbx_extractor* ex, int image, const orbx_keypoint** kps, const uint8_t** desc, int* n, int* mono,
                      const float** uright, const float** depth);

/* Batched many-camera mode: n_images device-resident images (image i at d_images + i*image_pitch, rows
 * row_pitch bytes apart; base and pitches 4-byte aligned), all w x h.  d_lap = n_images x 2 int32 lapping
 * areas on the HOST (NULL = all {0,0} as the rectified stereo callers pass, src/Frame.cc:200-201).
 * Enqueues the whole extraction on the handle's stream and returns without synchronising; the images must
 * stay valid until orbx_sync (level 0 of the pyramid aliases them). */
int orbx_extract_batch_device(orbx_extractor* ex, const uint8_t* d_images, int n_images, int w, int h,
                              ptrdiff_t row_pitch, ptrdiff_t image_pitch, const int32_t* lap);
int orbx_sync(orbx_extractor* ex);
/* The HIP stream (hipStream_t, returned as void*) the handle enqueues on: lets a caller queue its own consumers of the
 * device-resident results (a collective over the descriptor blocks, a copy) behind the extraction without a host
 * synchronisation.  The stream belongs to the handle. */
int orbx_stream_handle(const orbx_extractor* ex, void** stream);
/* The same batch from HOST memory (ideally page-locked: then the upload overlaps other handles' kernels): the frames
 * are uploaded into the handle's staging area with asynchronous copies on its stream, the extraction is enqueued behind
 * them, nothing synchronises.  The host frames must stay valid until orbx_sync. */
int orbx_extract_batch(orbx_extractor* ex, const uint8_t* images, int n_images, int w, int h, ptrdiff_t row_pitch,
                       ptrdiff_t image_pitch, const int32_t* lap);
/* All results of the last batch into HOST arrays with asynchronous copies on the handle's stream (page-locked arrays keep
 * them asynchronous): counts[n] / mono[n], kps[n][cap], desc[n][cap][32] (cap = orbx_batch_results_device's cap) and, when
 * n_pairs > 0 and a stereo association ran on this handle, uright / depth [n_pairs][cap].  NULL pointers are skipped.  The
 * arrays are complete after orbx_sync. */
int orbx_batch_download_async(orbx_extractor* ex, int32_t* counts, int32_t* mono, orbx_keypoint* kps, uint8_t* desc,
                              float* uright, float* depth, int n_pairs);

/* ---- Cross-camera descriptor exchange of the batched many-camera mode (BASELINE config C5).
 * The reference has no counterpart: it drives ONE rig per process (Frame's process-global statics,
 * include/Frame.h:230-235,314-319); north_star adds "RCCL over xGMI only for the optional cross-camera descriptor
 * all-gather", one process per GPU.  orbx_comm wraps an RCCL communicator (ncclComm_t):
 *   rank 0: orbx_comm_unique_id(id); the caller ships the 128 bytes to the other ranks (MPI, a file, torch.distributed);
 *   every rank: orbx_comm_create(id, n_ranks, rank, device, &comm)            (collective: all ranks must call it)
 *   or, for a process that already owns an ncclComm_t: orbx_comm_adopt(comm, device, &c) (not destroyed by liborbx).
 * orbx_allgather_descriptors enqueues, on the handle's own stream (behind the extraction, no host synchronisation),
 * ONE grouped RCCL call that gathers the first n_images images' results of the last batch from every rank, straight
 * from the handle's result arrays (no pack step, no staging copy):
 *   d_all_desc   [n_ranks][n_images][cap][32] u8    (cap = orbx_batch_results_device's cap; rows >= count are stale)
 *   d_all_counts [n_ranks][n_images] int32
 * ordered by rank, then by the rank's local image index.  Both destinations are DEVICE arrays owned by the caller and are
 * complete after orbx_sync (or for any work queued later on orbx_stream_handle's stream).  Every rank must call it with the
 * same n_images and a handle of the same capacity.  Errors: ORBX_E_UNSUPPORTED when no RCCL library can be loaded,
 * ORBX_E_NODEVICE without a GPU, ORBX_E_HIP for an RCCL failure (orbx_last_error holds ncclGetErrorString). */
#define ORBX_COMM_ID_BYTES 128
typedef struct orbx_comm orbx_comm;
int orbx_comm_unique_id(uint8_t id[ORBX_COMM_ID_BYTES]);
int orbx_comm_create(const uint8_t id[ORBX_COMM_ID_BYTES], int n_ranks, int rank, int device, orbx_comm** out);
int orbx_comm_adopt(void* nccl_comm, int device, orbx_comm** out);
void orbx_comm_destroy(orbx_comm* c);
int orbx_comm_size(const orbx_comm* c, int* n_ranks, int* rank);
int orbx_allgather_descriptors(orbx_extractor* ex, orbx_comm* c, int n_images, uint8_t* d_all_desc, int32_t* d_all_counts);
/* ORDERING RULE: one communicator per rank may (and should) serve every handle of that rank.  RCCL matches a communicator's
 * collectives by issue order, so every rank must make the same orbx_allgather_descriptors calls in the same order; the calls of
 * one communicator are chained on the device (each waits, stream-side, for the previous one's completion event), so handles
 * on different streams never run two collectives of the communicator concurrently.  orbx_comm_wait blocks the HOST until the
 * communicator's most recent collective has completed, at most timeout_ms: ORBX_OK, or ORBX_E_TIMEOUT with a message that
 * names the collective's sequence number and the likely causes -- a bounded wait in place of a hang in orbx_sync when a
 * rank is missing or out of order.  n_collectives (optional) receives the number of collectives enqueued so far. */
int orbx_comm_wait(orbx_comm* c, int timeout_ms, unsigned long long* n_collectives);

/* Device-resident results of the last (batch) extraction: keypoints [n_images][cap] and descriptors
 * [n_images][cap][32], counts[n_images] (n) and mono[n_images] (monoIndex), all on the device. */
int orbx_batch_results_device(const orbx_extractor* ex, const orbx_keypoint** d_kps, const uint8_t** d_desc,
                              const int32_t** d_counts, const int32_t** d_mono, int* cap);
/* Copy one image's results to the host (synchronises the handle's stream). Returns monoIndex or error. */
int orbx_batch_download(orbx_extractor* ex, int image, orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out);

/* Replaces reads of the public member ORBextractor::mvImagePyramid (include/ORBextractor.h:86;
 * used by src/Frame.cc:927,1011,1024,1029): copies level `level` of image `image` of the last extraction
 * to dst (dst_stride bytes per row; dst may be NULL to query the size only). blurred != 0 returns the
 * 7x7 Gaussian-blurred working copy (src/ORBextractor.cc:1074-1076) instead (computed on demand, once per extraction).
 * Lifetime: level 0 of a batch extracted with orbx_extract_batch_device IS the caller's device buffer (never copied), so
 * reading level 0 -- plain or blurred -- requires that buffer to be alive and unchanged; levels >= 1 and every level of the
 * host entry points (orbx_extract, orbx_extract_stereo, orbx_extract_batch) live in handle-owned memory until the next
 * extraction on the handle. */
int orbx_pyramid_level(orbx_extractor* ex, int image, int level, int blurred, uint8_t* dst,
                       ptrdiff_t dst_stride, int* w, int* h);
/* All (n_levels <= nlevels) levels of one image of the last extraction into caller buffers with ONE synchronisation:
 * dst[l] receives level l (w_l x h_l bytes, rows dst_stride[l] apart; NULL entries are skipped); the copies are queued
 * asynchronously on the handle's stream and the call returns after a single stream synchronisation.  This is what the C++
 * mirror's mvImagePyramid refresh uses (one call per eye instead of 2 x nlevels blocking copies).
 * Level 0 of a batch extracted with orbx_extract_batch_device is the CALLER's device buffer: it must still be alive. */
int orbx_pyramid_download(orbx_extractor* ex, int image, int n_levels, uint8_t* const* dst, const ptrdiff_t* dst_stride);
/* The reference keeps its pyramid in host memory and an UNMODIFIED Frame::ComputeStereoMatches reads it there
 * (`mpORBextractorLeft->mvImagePyramid[l]`, src/Frame.cc:927,1011,1024,1029; include/ORBextractor.h:86).
 * orbx_set_host_pyramid(handle, 1) makes the single-frame host entries (orbx_extract, orbx_extract_stereo) keep such a
 * host copy current: every level of their image(s) is copied into page-locked memory owned by the handle, by the DMA
 * engines on a side stream BESIDE the frame's kernels (orbx_extract_stereo: from the moment both pyramids exist), and the
 * call returns after both.  orbx_host_pyramid_level then hands out the level in place -- pointer, size and row stride,
 * exactly what a `cv::Mat(h, w, CV_8UC1, data, stride)` header needs -- with no further copy and no synchronisation.
 * Lifetime = the reference's: until the next extraction on the handle (src/ORBextractor.cc:1108-1145 overwrites
 * mvImagePyramid on every call).  image = 0 (orbx_extract; the left eye) or 1 (the right eye of orbx_extract_stereo). */
int orbx_set_host_pyramid(orbx_extractor* ex, int enable);
int orbx_host_pyramid_level(const orbx_extractor* ex, int image, int level, const uint8_t** data, int* w, int* h,
                            ptrdiff_t* stride);

/* Stage taps for differential tests: FAST candidates handed to DistributeOctTree for (image, level), in
 * unspecified order (x, y relative to the (16,16) window origin as in src/ORBextractor.cc:965-967;
 * response = FAST score).  Returns the count or a negative error; copies at most cap entries. */
int orbx_debug_candidates(orbx_extractor* ex, int image, int level, int32_t* xys, int cap);

/* ---- ORBmatcher / Frame matching -------------------------------------------------------------------- */

/* Replaces ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1959-1973) — host-side convenience, the
 * device matchers use v_bcnt on the same 8 little-endian words. */
int orbx_hamming256(const void* a, const void* b);

/* Replaces Frame::ComputeStereoMatches (src/Frame.cc:921-1084) for n_pairs rectified pairs whose left
 * images are images [first_left, first_left+n_pairs) of `left`'s last extraction and whose right images are
 * [first_right, ...) of `right`'s (left == right is allowed: one handle holding both eyes).  bf = mbf,
 * b = mb (maxD = bf / b).  Results stay on the device of `left`:
 * uRight / depth [n_pairs][cap_left] floats (-1 = no match).  Enqueued on left's stream after right's work. */
int orbx_stereo_match_batch(orbx_extractor* left, int first_left, orbx_extractor* right, int first_right,
                            int n_pairs, float bf, float b);
int orbx_stereo_results_device(const orbx_extractor* left, const float** d_uright, const float** d_depth);
/* Host copy of pair `pair` (synchronises): n = keypoint count of the left image. */
int orbx_stereo_download(orbx_extractor* left, int pair, float* uright, float* depth, int cap);

/* Replaces cv::BFMatcher(NORM_HAMMING).knnMatch(Q, T, k=2) + Lowe ratio of
 * Frame::ComputeStereoFishEyeMatches (src/Frame.cc:46,1293-1302).  Host descriptor rows in, host results
 * out: idx2 / dist2 [nQ][2] (-1 when the train set has fewer rows), ratio_ok[nQ] = d0 < d1*0.7. */
int orbx_bf_knn2(int device, const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, int32_t* idx2,
                 int32_t* dist2, uint8_t* ratio_ok);

/* The fisheye stereo rig Frame::ComputeStereoFishEyeMatches works on: the two KannalaBrandt8 cameras
 * (GeometricCamera::mvParameters = fx fy cx cy k0 k1 k2 k3, include/CameraModels/KannalaBrandt8.h:42-57), the Newton
 * stop of KannalaBrandt8::unproject (`precision`, :102) and the left-from-right transform mRlr / mtlr
 * (include/Frame.h:208-209, src/Frame.cc:1240-1243), R12 row-major. */
typedef struct orbx_kb8_rig {
  float cam1[8];
  float cam2[8];
  float precision;
  float R12[9];
  float t12[3];
} orbx_kb8_rig;

/* Replaces the whole of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1273-1331): brute-force 2-NN of the
 * lapping-area rows [mono_left, n_left) x [mono_right, n_right) (as orbx_bf_knn2), Lowe ratio 0.7, and for every
 * surviving pair KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:341-432: unproject by Newton
 * iteration, parallax gate 0.9998, linear triangulation = smallest right singular vector of the 4x4 system, cheirality
 * and the two chi-square reprojection gates 5.991 * mvLevelSigma2[octave]).  Outputs (host, caller-allocated):
 * left_to_right[n_left] = mvLeftToRightMatch, right_to_left[n_right] = mvRightToLeftMatch (serial semantics: a right
 * keypoint claimed by several left ones keeps the LAST), depth[n_left] = mvDepth (-1 = none), points3d[n_left][3] =
 * mvStereo3Dpoints (zeros where unmatched), *n_desc_matches = pairs that passed the ratio test (may be NULL).
 * Returns nMatches >= 0 or an ORBX_E_* code.  This is the one floating-point routine of the path: results agree with
 * the reference to float rounding (device libm, no Eigen), not bit for bit -- see DESIGN.md. */
int orbx_fisheye_stereo_match(int device, const orbx_keypoint* kps_left, const uint8_t* desc_left, int n_left,
                              int mono_left, const orbx_keypoint* kps_right, const uint8_t* desc_right, int n_right,
                              int mono_right, const orbx_kb8_rig* rig, const float* level_sigma2, int n_levels,
                              int32_t* left_to_right, int32_t* right_to_left, float* depth, float* points3d,
                              int32_t* n_desc_matches);

/* The same routine on device-resident extraction results (config C4, batched many-camera mode): pair p associates
 * image first_left + p of `left`'s last extraction with image first_right + p of `right`'s (left == right allowed);
 * the lapping rows [monoIndex, n) and mvLevelSigma2 come from the handles.  Enqueued on left's stream after right's
 * work; results stay on the device of `left`: left_to_right / depth [n_pairs][cap_left], points3d [n_pairs][cap_left][3],
 * right_to_left [n_pairs][cap_right], counts [n_pairs][2] = {nMatches, descMatches}. */
int orbx_fisheye_stereo_match_batch(orbx_extractor* left, int first_left, orbx_extractor* right, int first_right,
                                    int n_pairs, const orbx_kb8_rig* rig);
int orbx_fisheye_results_device(const orbx_extractor* left, const int32_t** d_left_to_right,
                                const int32_t** d_right_to_left, const float** d_depth, const float** d_points3d,
                                const int32_t** d_counts);
/* Host copy of pair `pair` (synchronises; any pointer may be NULL).  Returns nMatches or a negative error. */
int orbx_fisheye_download(orbx_extractor* left, int pair, int32_t* left_to_right, int32_t* right_to_left, float* depth,
                          float* points3d, int cap_left, int cap_right, int32_t* n_desc_matches);

/* ---- image pre-processing in front of the extractor ------------------------------------------ */

/* Replaces cv::cvtColor(im, im, cv::COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) of Tracking::GrabImageStereo / RGBD / Monocular
 * (src/Tracking.cc:1394-1412, 1441-1459, 1481-1499): OpenCV's 8-bit fixed-point formula, coefficients 9798 / 19235 /
 * 3735 with (sum + 16384) >> 15.  channels = 3 or 4 (interleaved), rgb_order != 0 when the first channel is red (mbRGB).
 * Host images in and out. */
int orbx_cvt_gray(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, int rgb_order,
                  uint8_t* dst, ptrdiff_t dst_stride);
/* Replaces cv::resize(im, out, newImSize) (INTER_LINEAR) of System::TrackStereo / TrackRGBD / TrackMonocular
 * (src/System.cc:297-298, 369-370, 437-438) for 8UC1 / 8UC3 / 8UC4 images: the fixed-point bilinear arithmetic of
 * ComputePyramid's cv::resize on every interleaved channel. */
int orbx_resize_linear(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, uint8_t* dst,
                       int dst_w, int dst_h, ptrdiff_t dst_stride);

/* Replaces cv::remap(im, imToFeed, M1, M2, cv::INTER_LINEAR) of System::TrackStereo (src/System.cc:294-295) with the
 * CV_32F maps Settings::precomputeRectificationMaps builds (src/Settings.cc:557-572): OpenCV's fixed-point bilinear remap
 * (positions rounded to 1/32 px, 15-bit weights, BORDER_CONSTANT 0) on 8UC1 / 8UC3 / 8UC4.  map_x / map_y hold dst_h rows of
 * dst_w floats, map_stride floats apart.  Host images in and out. */
int orbx_remap_linear(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, const float* map_x,
                      const float* map_y, ptrdiff_t map_stride, uint8_t* dst, int dst_w, int dst_h, ptrdiff_t dst_stride);
/* Replaces cv::createCLAHE(clip_limit, cv::Size(tiles_x, tiles_y))->apply(im, im) of the TUM-VI front ends
 * (Examples/Stereo/stereo_tum_vi.cc:100,142-143; Examples/Stereo-Inertial/stereo_inertial_tum_vi.cc:151,190-191; the
 * examples pass 3.0 and 8 x 8) on 8UC1.  Host images in and out (src == dst is allowed). */
int orbx_clahe(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, double clip_limit, int tiles_x, int tiles_y,
               uint8_t* dst, ptrdiff_t dst_stride);

/* Device-resident pre-processing chain in front of the extractor, so that raw camera frames never return to the host:
 * [CLAHE] -> [remap | resize] -> [gray], the order in which the reference applies them (example main: clahe->apply;
 * System::TrackStereo: remap or resize, src/System.cc:288-302; Tracking::GrabImageStereo: cvtColor, src/Tracking.cc:1394-1412).
 * A stage is enabled by its fields: clahe_tiles_x/y > 0 (single-channel frames only); map_x/map_y != NULL (n_maps maps of
 * out_h x out_w floats each, map m directly after map m-1, rows map_stride floats apart, 0 = out_w; frame i of a batch uses
 * map i % n_maps, i.e. 2 maps = left / right eye interleaved); otherwise out_w x out_h != src size enables cv::resize;
 * channels 3 / 4 enables the gray conversion.  The maps are copied to the device at creation. */
typedef struct orbx_preproc_params {
  int32_t src_w, src_h, channels, rgb_order;
  int32_t out_w, out_h;
  const float* map_x;
  const float* map_y;
  ptrdiff_t map_stride;
  int32_t n_maps;
  double clahe_clip_limit;
  int32_t clahe_tiles_x, clahe_tiles_y;
} orbx_preproc_params;
typedef struct orbx_preproc orbx_preproc;
int orbx_preproc_create(const orbx_preproc_params* p, int max_batch, int device, orbx_preproc** out);
void orbx_preproc_destroy(orbx_preproc* pp);
int orbx_preproc_output_size(const orbx_preproc* pp, int* out_w, int* out_h);
/* One host frame through the chain (map `map_index`), host result out_w x out_h gray. */
int orbx_preproc_run(orbx_preproc* pp, const uint8_t* frame, ptrdiff_t stride, int map_index, uint8_t* dst, ptrdiff_t dst_stride);
/* n_frames device-resident raw frames (frame i at d_frames + i*image_pitch) through the chain; synchronises and returns the
 * device-resident result (owned by the handle, valid until its next run). */
int orbx_preproc_run_device(orbx_preproc* pp, const uint8_t* d_frames, int n_frames, ptrdiff_t row_pitch,
                            ptrdiff_t image_pitch, const uint8_t** d_out, int* out_w, int* out_h, ptrdiff_t* out_row_pitch,
                            ptrdiff_t* out_image_pitch);
/* orbx_extract_batch_device on raw frames: the chain and the extraction are enqueued on the extractor's stream, nothing
 * synchronises.  The pre-processor's buffers back level 0 of the pyramid until orbx_sync: use one orbx_preproc per extractor
 * handle in flight. */
int orbx_extract_batch_raw_device(orbx_extractor* ex, orbx_preproc* pp, const uint8_t* d_frames, int n_frames,
                                  ptrdiff_t row_pitch, ptrdiff_t image_pitch, const int32_t* lap);

/* ---- bag of words --------------------------------------------------------------------------------- */

/* Replaces ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (include/ORBVocabulary.h;
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) as far as Frame::ComputeBoW uses it: the tree on the device.
 * orbx_vocabulary_load_text replaces loadFromTextFile(strVocFile) of System::System (src/System.cc:131,
 * TemplatedVocabulary.h:1338-1421; "k L scoring weighting", then one node per line: parent isLeaf 32 bytes weight).
 * orbx_vocabulary_create takes the same columns: node 0 = root (its columns are ignored), parent[i] < i, the children of a
 * node in file order, words numbered in the file order of the leaves.  info = k, L, n_nodes, n_words, scoring, weighting. */
typedef struct orbx_vocabulary orbx_vocabulary;
int orbx_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                           const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights, orbx_vocabulary** out);
int orbx_vocabulary_load_text(int device, const char* path, orbx_vocabulary** out);
void orbx_vocabulary_destroy(orbx_vocabulary* voc);
int orbx_vocabulary_info(const orbx_vocabulary* voc, int32_t info[6]);

/* Replaces Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:846-851, src/KeyFrame.cc:100-107):
 * mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) (TemplatedVocabulary.h:1125-1250) for n descriptors (n x 32
 * bytes, n <= 8192).  mBowVec: n_words ascending (word id, value) pairs, values bit-identical to the reference's sequential
 * double additions and normalisation; mFeatVec as CSR: n_nodes ascending node ids, features of node j =
 * feature_idx[node_start[j] .. node_start[j + 1]) in ascending order.  Output arrays hold n entries (node_start n + 1).
 * Returns the number of features in the feature vector (stopped words are left out) or a negative error. */
int orbx_bow_transform(const orbx_vocabulary* voc, const uint8_t* desc, int n, int levelsup, uint32_t* word_ids,
                       double* word_values, int* n_words, uint32_t* node_ids, int32_t* node_start, uint32_t* feature_idx,
                       int* n_nodes);
/* The same for every image of the handle's last extraction, enqueued on its stream; results stay on the device: arrays of
 * [n_images][cap] (node_start [n_images][cap + 1]), counts [n_images][3] = n_words, n_nodes, n_features. */
int orbx_bow_transform_batch(orbx_extractor* ex, const orbx_vocabulary* voc, int levelsup);
int orbx_bow_results_device(const orbx_extractor* ex, const uint32_t** d_word_ids, const double** d_word_values,
                            const uint32_t** d_node_ids, const int32_t** d_node_start, const uint32_t** d_feature_idx,
                            const int32_t** d_counts, int* cap);
/* Host copy of image `image` (synchronises); cap = capacity of the arrays.  Returns n_features or a negative error. */
int orbx_bow_download(orbx_extractor* ex, int image, uint32_t* word_ids, double* word_values, int* n_words, uint32_t* node_ids,
                      int32_t* node_start, uint32_t* feature_idx, int* n_nodes, int cap);

/* Replaces ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
 * (src/ORBmatcher.cc:230-404).  kf_* = pKF->mFeatVec (CSR as above), its keypoints (angle is read; left | right
 * concatenated for two-camera rigs), mDescriptors and kf_valid[i] = (vpMapPointsKF[i] && !isBad()); f_* = F.mFeatVec,
 * F's keypoints and descriptors, n_left_f = F.Nleft (-1 for monocular / rectified frames).  matches[i] = index of the
 * keyframe feature whose map point F's feature i received, or -1 (vpMapPointMatches).  nnratio = mfNNratio,
 * check_orientation = mbCheckOrientation.  Returns nmatches or a negative error. */
int orbx_search_by_bow(int device, const uint32_t* kf_node_ids, const int32_t* kf_node_start, const uint32_t* kf_feature_idx,
                       int n_kf_nodes, const orbx_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_valid, int n_kf,
                       const uint32_t* f_node_ids, const int32_t* f_node_start, const uint32_t* f_feature_idx, int n_f_nodes,
                       const orbx_keypoint* f_kps, const uint8_t* f_desc, int n_f, int n_left_f, float nnratio,
                       int check_orientation, int32_t* matches);
/* The same search for the frames of an extraction BATCH: frame f = image first_image + f of `ex`'s last batch, whose keypoints,
 * descriptors and feature vector (orbx_bow_transform_batch must have run on that extraction) stay in HBM; the key frame of pair f
 * comes from the host as strided arrays -- kf_node_ids [n_frames][nodes_stride], kf_node_start [n_frames][nodes_stride + 1],
 * kf_feature_idx / kf_kps / kf_desc / kf_valid [n_frames][kf_stride](x 32) with n_kf_nodes[f] / n_kf[f] valid entries.
 * matches is [n_frames][cap] (cap = orbx_batch_results_device's cap; -1 past a frame's keypoints), n_matches [n_frames].
 * Every kernel runs ONCE for all pairs (blockIdx.y = pair); results are those of n_frames separate calls.
 * Returns the total number of matches or a negative error. */
int orbx_search_by_bow_batch(orbx_extractor* ex, int first_image, int n_frames, const uint32_t* kf_node_ids,
                             const int32_t* kf_node_start, const int32_t* n_kf_nodes, int nodes_stride,
                             const uint32_t* kf_feature_idx, const orbx_keypoint* kf_kps, const uint8_t* kf_desc,
                             const uint8_t* kf_valid, const int32_t* n_kf, int kf_stride, int n_left_f, float nnratio,
                             int check_orientation, int32_t* matches, int32_t* n_matches);

/* Replaces Frame::UndistortKeyPoints (src/Frame.cc:853-885): mvKeysUn from mvKeys through
 * cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK) -- five fixed-point iterations of the inverse distortion
 * in double, then x' = fx x + cx.  K = fx fy cx cy (Pinhole::toK()); dist = the n_dist (4, 5, 8, 12 or 14) OpenCV
 * coefficients of mDistCoef, tilt terms unsupported (must be 0).  dist[0] == 0 copies the keypoints, as the reference
 * does.  Host arrays in and out; out may alias kps. */
int orbx_undistort_keypoints(int device, const orbx_keypoint* kps, int n, const float K[4], const float* dist, int n_dist,
                             orbx_keypoint* out);
/* Replaces Frame::ComputeImageBounds (src/Frame.cc:887-919): bounds = mnMinX, mnMinY, mnMaxX, mnMaxY of a cols x rows
 * image (the undistorted corners, or 0, 0, cols, rows when dist[0] == 0). */
int orbx_compute_image_bounds(int device, int cols, int rows, const float K[4], const float* dist, int n_dist,
                              float bounds[4]);

/* Replaces ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:618-764) incl. Frame::GetFeaturesInArea /
 * AssignFeaturesToGrid / PosInGrid on F2 (src/Frame.cc:520-547,765-844) and ComputeThreeMaxima
 * (src/ORBmatcher.cc:1920-1955).  kps are the undistorted keypoints (mvKeysUn); bounds = mnMinX, mnMinY,
 * mnMaxX, mnMaxY of F2; prev_matched = vbPrevMatched (2*n1 floats, in/out); matches12 = vnMatches12 (n1).
 * Returns nmatches or a negative error. */
int orbx_search_for_initialization(int device, const orbx_keypoint* kps1, const uint8_t* desc1, int n1,
                                   const orbx_keypoint* kps2, const uint8_t* desc2, int n2, float min_x,
                                   float min_y, float max_x, float max_y, float* prev_matched,
                                   int32_t* matches12, int window_size, float nnratio, int check_orientation);
/* The same search for the frames of an extraction BATCH (many-camera mode: every camera still initialising its map matches its
 * initial frame against its current one, src/Tracking.cc:2438-2440, in one call).  Pair f: F2 = image first_image + f of `ex`'s
 * last batch (keypoints taken as mvKeysUn and descriptors stay in HBM), F1 = kps1 / desc1 [f * stride .. f * stride + n1[f]) on
 * the host; prev_matched is [n_frames][stride][2] (in/out), matches12 [n_frames][stride], n_matches [n_frames].  Every kernel
 * of the chain runs ONCE for all pairs (blockIdx.y = pair) with a fixed number of fixed-point rounds enqueued without reading
 * a convergence flag; a pair that needs more rounds or larger candidate lists is redone through the one-shot call: results are
 * those of n_frames separate orbx_search_for_initialization calls.  Returns the total number of matches or a negative error. */
int orbx_search_for_initialization_batch(orbx_extractor* ex, int first_image, int n_frames, const orbx_keypoint* kps1,
                                         const uint8_t* desc1, const int32_t* n1, int stride, float min_x, float min_y,
                                         float max_x, float max_y, float* prev_matched, int32_t* matches12, int window_size,
                                         float nnratio, int check_orientation, int32_t* n_matches);

/* Replaces Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:520-547,833-844: 64 x 48 grid, round-to-cell) and
 * Frame::GetFeaturesInArea (src/Frame.cc:765-831) for a batch of queries.  kps = mvKeysUn (n), bounds = mnMinX/Y,
 * mnMaxX/Y; queries = n_queries x {x, y, r, minLevel, maxLevel} floats.  Output is CSR: offsets[n_queries + 1] and
 * indices (keypoint indices in the reference's order: ix outer, iy inner, in-cell ascending).  Optionally returns
 * the grid itself: grid_cell_start[64*48 + 1] (cell ix*48 + iy) and grid_items[n] (= mGrid[ix][iy] concatenated).
 * Returns the total number of indices or a negative error (ORBX_E_CAPACITY if indices_cap is too small; offsets are
 * still valid then). */
int orbx_features_in_area(int device, const orbx_keypoint* kps, int n, float min_x, float min_y, float max_x,
                          float max_y, const float* queries, int n_queries, int32_t* offsets, int32_t* indices,
                          int indices_cap, int32_t* grid_cell_start, int32_t* grid_items);

/* ---- widening row f1: projection-guided matching ------------------------------------------ */

/* The MapPoint members ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, ...) reads
 * (src/ORBmatcher.cc:41-221): mTrackProjX/Y/XR, mTrackViewCos, mTrackDepth, mnTrackScaleLevel, mbTrackInView,
 * isBad(), Observations() > 0, GetDescriptor().  60 bytes. */
typedef struct orbx_map_point_view {
  float proj_x, proj_y, proj_xr, view_cos, track_depth;
  int32_t predicted_level;
  uint8_t in_view, bad, has_observations, pad_;
  uint8_t desc[32];
} orbx_map_point_view;

/* Replaces ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints,
 * thFarPoints) (src/ORBmatcher.cc:41-221) for the pinhole case (F.Nleft == -1), serial iMP semantics.
 * F is given by mvKeysUn (n), mDescriptors, mvuRight (may be NULL), the grid bounds mnMinX/Y, mnMaxX/Y and
 * mvScaleFactors (nlevels).  occupied[i] != 0 <=> F.mvpMapPoints[i] already holds a point with Observations() > 0
 * (in/out).  match[i] receives the index of the map point newly assigned to keypoint i, or -1.
 * Returns nmatches or a negative error. */
int orbx_search_by_projection(int device, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right,
                              int n, float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                              int nlevels, const orbx_map_point_view* map_points, int n_map_points, float th,
                              int far_points, float th_far_points, float nnratio, uint8_t* occupied, int32_t* match);

/* One LastFrame point after the reference's pose / camera projection (src/ORBmatcher.cc:1606-1648): uv, the
 * right-image coordinate ur = u - mbf * invz, radius = th * mvScaleFactors[nLastOctave], the level window picked by
 * bForward / bBackward ((nLastOctave, -1), (0, nLastOctave) or (nLastOctave-1, nLastOctave+1)), the last-frame
 * keypoint angle, the MapPoint's descriptor and Observations() > 0.  valid = 0 for points the reference skips
 * (no MapPoint, outlier, invz < 0, projection outside the image).  64 bytes. */
typedef struct orbx_projected_point {
  float u, v, ur, radius, angle;
  int32_t min_level, max_level;
  uint8_t valid, has_observations, pad_[2];
  uint8_t desc[32];
} orbx_projected_point;

/* Replaces the matching part of ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th,
 * bMono) (src/ORBmatcher.cc:1594-1806, pinhole case): window search on CurrentFrame's grid, occupancy gate,
 * stereo-consistency gate, best Hamming <= TH_HIGH, assignment in serial order, rotation-histogram cull
 * (check_orientation).  match[i2] = LastFrame point index assigned to keypoint i2 or -1.  Returns nmatches. */
int orbx_search_by_projection_frame(int device, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right,
                                    int n, float min_x, float min_y, float max_x, float max_y,
                                    const orbx_projected_point* points, int n_points, int check_orientation,
                                    uint8_t* occupied, int32_t* match);

/* The two pinhole SearchByProjection flavours above on the frames of an extraction BATCH (the many-camera mode: one tracking step
 * of n_frames cameras).  Frame f = image first_image + f of `ex`'s last batch: its keypoints (taken as mvKeysUn: pinhole without
 * distortion / rectified input, like the batch itself) and descriptors stay in HBM; points / map points of frame f are
 * points[f * points_stride .. + n_points[f]); bounds = mnMinX .. mnMaxY; scale factors = the handle's; stereo_pair0 >= 0 takes
 * mvuRight of frame f from pair stereo_pair0 + f of the handle's last orbx_stereo_match_batch (-1: monocular, no consistency
 * check); occupied_in (may be NULL = all free) / occupied / match are [n_frames][cap] with cap = orbx_batch_results_device's cap,
 * n_matches [n_frames].  Every kernel of the chain runs ONCE for all frames (blockIdx.y = frame) with a fixed number of
 * fixed-point rounds enqueued blindly; a frame that needs more (or larger candidate lists) is redone through the one-shot path:
 * results are those of n_frames separate calls.  map_points == NULL (round 6): the views come from the handle's last
 * orbx_project_map_points_batch and never touch the host (points_stride and every n_map_points[f] must equal its n); points ==
 * NULL in the frame flavour: from the last orbx_project_last_frames_batch (points_stride / n_points[f] = the upload's).
 * Returns the total number of matches or a negative error. */
int orbx_search_by_projection_batch(orbx_extractor* ex, int first_image, int n_frames, float min_x, float min_y, float max_x,
                                    float max_y, const orbx_map_point_view* map_points, const int32_t* n_map_points,
                                    int points_stride, float th, int far_points, float th_far_points, float nnratio,
                                    int stereo_pair0, const uint8_t* occupied_in, uint8_t* occupied, int32_t* match,
                                    int32_t* n_matches);
int orbx_search_by_projection_frame_batch(orbx_extractor* ex, int first_image, int n_frames, float min_x, float min_y, float max_x,
                                          float max_y, const orbx_projected_point* points, const int32_t* n_points,
                                          int points_stride, int check_orientation, int stereo_pair0, const uint8_t* occupied_in,
                                          uint8_t* occupied, int32_t* match, int32_t* n_matches);

/* ---- device-side projection for the batched local-map matcher (round 6) -----------------------------------------------------
 * Tracking::SearchLocalPoints (src/Tracking.cc:3303-3328) calls Frame::isInFrustum (src/Frame.cc:632-690) for every local map
 * point, which stores mTrackProjX / Y / XR, mTrackDepth, mnTrackScaleLevel (MapPoint::PredictScale, src/MapPoint.cc:559-573),
 * mTrackViewCos and mbTrackInView in the MapPoint; SearchByProjection then reads them back (src/ORBmatcher.cc:62-76).  Here the
 * local map is uploaded ONCE as structure-of-arrays, every frame of the batch contributes its pose, and the projection, the frustum
 * / distance / viewing-angle gates and the level prediction run on the device: the views never exist on the host.
 * Pose of a pinhole frame = the members isInFrustum reads: mRcw (row-major), mtcw, mOw, the Pinhole parameters fx fy cx cy, mbf. */
typedef struct orbx_frame_pose {
  float Rcw[9], tcw[3], Ow[3], fx, fy, cx, cy, bf;
} orbx_frame_pose;
/* n local map points: GetWorldPos(), GetNormal() (n x 3 each), mfMinDistance / mfMaxDistance (the 0.8 / 1.2 factors of
 * Get{Min,Max}DistanceInvariance are applied by the library), GetDescriptor() (n x 32), flags bit 0 = isBad(), bit 1 =
 * Observations() > 0.  Copied into the handle; stays valid until the next upload. */
int orbx_map_upload(orbx_extractor* ex, int n, const float* world_pos, const float* normal, const float* min_distance,
                    const float* max_distance, const uint8_t* desc, const uint8_t* flags);
/* isInFrustum(pMP, viewing_cos_limit) of all n uploaded points for n_frames frames (bounds = mnMinX .. mnMaxY); skip (may be NULL)
 * is [n_frames][n], != 0 = the point is not offered to this frame (already matched: mnLastFrameSeen == frame id,
 * src/Tracking.cc:3310-3316).  The views [n_frames][n] stay on the device, attached to the handle, for
 * orbx_search_by_projection_batch(map_points = NULL, points_stride = n, n_map_points[f] = n); views_out (may be NULL) receives
 * a host copy.  Float arithmetic in the reference's expression order; PredictScale's logarithm is the device's logf: a level may
 * differ from a glibc build by one where log(ratio) / logScaleFactor is within rounding of an integer. */
int orbx_project_map_points_batch(orbx_extractor* ex, int n_frames, const orbx_frame_pose* poses, float min_x, float min_y,
                                  float max_x, float max_y, float viewing_cos_limit, const uint8_t* skip,
                                  orbx_map_point_view* views_out);

/* ---- device-side projection for the batched frame-to-frame matcher (round 6) -------------------------------------------------
 * ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) projects LastFrame's map points itself
 * (src/ORBmatcher.cc:1606-1648): x3Dc = Tcw * x3Dw with Tcw a Sophus::SE3f -- the unit-quaternion sandwich of
 * Thirdparty/Sophus/sophus/so3.hpp:358-366 plus the translation --, invzc = 1.0 / z, Pinhole::project, the bounds skips, radius =
 * th * mvScaleFactors[nLastOctave], the level window by bForward / bBackward, ur = u - mbf * invzc.  Here the LastFrames of the
 * batch's cameras are uploaded once per tracking step as structure-of-arrays, every camera contributes its pose, and the
 * orbx_projected_point views are made on the device and stay there for orbx_search_by_projection_frame_batch(points = NULL).
 * Pose = Tcw as Sophus stores it (quaternion x y z w, translation), the Pinhole parameters, mbf and the caller's forward /
 * backward decision (:1611-1612, from tlc(2) and mb): direction 0 = neither, 1 = bForward, 2 = bBackward. */
typedef struct orbx_frame_pose_q {
  float q[4], t[3], fx, fy, cx, cy, bf;
  int32_t direction;
} orbx_frame_pose_q;
/* LastFrame f of camera f (n_frames cameras, points_stride entries each, n_points[f] used): per keypoint i of that LastFrame the
 * world position of mvpMapPoints[i] (n_frames x stride x 3), mvKeys[i].octave, mvKeysUn[i].angle, pMP->GetDescriptor()
 * (x 32), flags bit 0 = the point exists and is no outlier (:1615-1617), bit 1 = Observations() > 0.  Copied into the handle. */
int orbx_last_frames_upload(orbx_extractor* ex, int n_frames, int points_stride, const int32_t* n_points, const float* world_pos,
                            const int32_t* octave, const float* angle, const uint8_t* desc, const uint8_t* flags);
/* The projection block for every uploaded point of n_frames frames (scale factors = the handle's; bounds = mnMinX .. mnMaxY).
 * views_out (may be NULL) receives a host copy [n_frames][points_stride].  Float arithmetic in the reference's expression order
 * (tolerance parity at the gates, like orbx_project_map_points_batch; 0 / 0 projections are marked invalid). */
int orbx_project_last_frames_batch(orbx_extractor* ex, int n_frames, const orbx_frame_pose_q* poses, float min_x, float min_y,
                                   float max_x, float max_y, float th, orbx_projected_point* views_out);

/* Replaces the matching part of the relocalisation matcher ORBmatcher::SearchByProjection(Frame& CurrentFrame,
 * KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist) (src/ORBmatcher.cc:1808-1918; callers
 * src/Tracking.cc:3631-3632,3645-3646 with (th, ORBdist) = (10, 100) and (3, 64)).  points[i] = the key frame's i-th
 * map point after the caller's pose / camera projection and gates (:1826-1851): valid = 0 for a missing / bad /
 * already-found point, a projection outside the image bounds or a distance outside the scale-invariance range; (u, v);
 * radius = th * mvScaleFactors[nPredictedLevel]; (min_level, max_level) = (nPredictedLevel - 1, nPredictedLevel + 1);
 * angle = pKF->mvKeysUn[i].angle; desc = pMP->GetDescriptor().  ur and has_observations are not read: this flavour has
 * no stereo gate, and its occupancy gate is a plain non-null test (:1871), so EVERY assignment occupies its keypoint.
 * occupied[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2] != NULL (in/out: on return also set for the new matches and
 * clear again for the ones the rotation-consistency cull removed, :1910-1913).  Best Hamming < 256 by strict first
 * minimum over the free candidates, accepted if <= orb_dist (0..255).  match[i2] = index i of the assigned point or -1.
 * Returns nmatches after the cull, or a negative error.
 * The loop-closing searches SearchByProjection(KeyFrame* pKF, Sim3f& Scw, vpPoints, vpMatched, th, ratioHamming) and its
 * vpMatchedKF twin (src/ORBmatcher.cc:406-503, :505-612) run the same loop -- occupancy = vpMatched[idx] != NULL, level window
 * [nPredictedLevel - 1, nPredictedLevel], `bestDist <= TH_LOW * ratioHamming`, no orientation check -- and are served by this
 * entry with (min_level, max_level) = (nPredictedLevel - 1, nPredictedLevel), check_orientation = 0 and
 * orb_dist = floor(TH_LOW * ratioHamming) (tests/test_reloc_triangulation.py checks a literal transcription of that loop). */
int orbx_search_by_projection_keyframe(int device, const orbx_keypoint* kps_un, const uint8_t* desc, int n, float min_x,
                                       float min_y, float max_x, float max_y, const orbx_projected_point* points,
                                       int n_points, int orb_dist, int check_orientation, uint8_t* occupied,
                                       int32_t* match);

/* Replaces ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<pair<size_t, size_t>>& vMatchedPairs,
 * bOnlyStereo, bCoarse) (src/ORBmatcher.cc:886-1106; caller LocalMapping::CreateNewMapPoints) for single-camera key frames
 * (mpCamera2 == NULL in both; NLeft == -1).  *_1 / *_2 = pKF1 / pKF2: mFeatVec as CSR (ascending node ids, as orbx_bow_transform
 * returns it), mvKeysUn, mDescriptors, has_map_point[i] = (GetMapPoint(i) != NULL), u_right = mvuRight (NULL: no stereo
 * observations, i.e. monocular).  scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2 (nlevels2 entries);
 * ep = pKF2->mpCamera->project(T2w * pKF1->GetCameraCenter()) (:897-901); F12 = the fundamental matrix
 * Pinhole::epipolarConstrain forms on every call, K1^-T [t12]x R12 K2^-1 (src/CameraModels/Pinhole.cpp:130-133), row-major,
 * computed once by the caller (not read when coarse != 0).  matches12[idx1] = idx2 or -1; vMatchedPairs = its non-negative
 * entries in ascending idx1 (:1095-1103).  Like the reference (vbMatched2 is never set, :933,976) one feature of pKF2 may be
 * paired with several of pKF1.  Returns nmatches after the rotation-consistency cull, or a negative error. */
int orbx_search_for_triangulation(int device, const uint32_t* node_ids1, const int32_t* node_start1, const uint32_t* feature_idx1,
                                  int n_nodes1, const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1,
                                  const float* u_right1, int n1, const uint32_t* node_ids2, const int32_t* node_start2,
                                  const uint32_t* feature_idx2, int n_nodes2, const orbx_keypoint* kps2, const uint8_t* desc2,
                                  const uint8_t* has_map_point2, const float* u_right2, int n2, const float* scale_factors2,
                                  const float* level_sigma2_2, int nlevels2, const float ep[2], const float F12[9], int only_stereo,
                                  int coarse, int check_orientation, int32_t* matches12);

/* The two-camera members ORBmatcher::SearchForTriangulation reads when pKF1->mpCamera2 && pKF2->mpCamera2
 * (src/ORBmatcher.cc:906-923, 1007-1042): cam[0..3] = the KannalaBrandt8 mvParameters (fx fy cx cy k0..k3) of pKF1->mpCamera,
 * pKF1->mpCamera2, pKF2->mpCamera, pKF2->mpCamera2; precision = KannalaBrandt8::precision; R[c] / t[c] = rotation (row-major) and
 * translation of Tll = T1w * Tw2, Tlr = T1w * Twr2, Trl = Tr1w * Tw2, Trr = Tr1w * Twr2.  81 floats. */
typedef struct orbx_tri_rig {
  float cam[4][8];
  float precision;
  float R[4][9], t[4][3];
} orbx_tri_rig;

/* ORBmatcher::SearchForTriangulation for two-camera (stereo-fisheye) key frames: kps / desc / has_map_point hold the N = NLeft +
 * NRight features of each key frame (mvKeys then mvKeysRight, n_left = KeyFrame::NLeft); the epipolar test is
 * KannalaBrandt8::epipolarConstrain = TriangulateMatches(...) > 0.0001f (src/CameraModels/KannalaBrandt8.cpp:240-250, :341-417)
 * with (R12, t12, cameras) chosen by the eyes the two features sit in (:1007-1042), sigmaLevel = level_sigma2_1[kp1.octave], unc =
 * level_sigma2_2[kp2.octave]; there is no epipole gate and bStereo is false for every feature, so only_stereo != 0 matches
 * nothing (:957-959).  Float arithmetic in the reference's expression order, the 4x4 null vector by a double one-sided Jacobi
 * instead of Eigen::JacobiSVD<Matrix4f>: accept / reject decisions equal the oracle's unless a gated quantity lies within
 * rounding noise of its threshold (tolerance parity, like orbx_fisheye_stereo_match).  Returns nmatches or a negative error. */
int orbx_search_for_triangulation_rig(int device, const uint32_t* node_ids1, const int32_t* node_start1, const uint32_t* feature_idx1,
                                      int n_nodes1, const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1,
                                      int n_left1, int n1, const uint32_t* node_ids2, const int32_t* node_start2,
                                      const uint32_t* feature_idx2, int n_nodes2, const orbx_keypoint* kps2, const uint8_t* desc2,
                                      const uint8_t* has_map_point2, int n_left2, int n2, const float* level_sigma2_1,
                                      const float* level_sigma2_2, int nlevels, const orbx_tri_rig* rig, int only_stereo, int coarse,
                                      int check_orientation, int32_t* matches12);

/* Replaces ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:766-884;
 * LoopClosing).  *_1 / *_2 = pKF1 / pKF2: mFeatVec as CSR, mvKeysUn (the angle is read), mDescriptors and valid[i] =
 * (vpMapPoints[i] && !vpMapPoints[i]->isBad() && !(NLeft != -1 && i >= mvKeysUn.size())) (:799-806,:817-825).  matches12[idx1] =
 * the feature of pKF2 whose map point vpMatches12[idx1] receives, or -1: per vocabulary node, pKF1's features in list order, best
 * and second best over the partner node's features that are valid and not taken yet (vbMatched2), accepted if bestDist1 < TH_LOW
 * (strict) and bestDist1 < nnratio * bestDist2; then the rotation-consistency cull.  Returns nmatches or a negative error. */
int orbx_search_by_bow_keyframes(int device, const uint32_t* node_ids1, const int32_t* node_start1, const uint32_t* feature_idx1,
                                 int n_nodes1, const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* valid1, int n1,
                                 const uint32_t* node_ids2, const int32_t* node_start2, const uint32_t* feature_idx2, int n_nodes2,
                                 const orbx_keypoint* kps2, const uint8_t* desc2, const uint8_t* valid2, int n2, float nnratio,
                                 int check_orientation, int32_t* matches12);

/* One map point of ORBmatcher::Fuse after the reference's projection and gates (src/ORBmatcher.cc:1141-1192: not bad, not already
 * in the key frame, positive depth, inside the image, distance inside the scale-invariance range, viewing angle below 60 degrees):
 * uv, ur = u - mbf * invz, radius = th * mvScaleFactors[nPredictedLevel], nPredictedLevel, GetDescriptor().  valid = 0 for the
 * points the reference skips.  56 bytes. */
typedef struct orbx_fuse_point {
  float u, v, ur, radius;
  int32_t predicted_level;
  uint8_t valid, pad_[3];
  uint8_t desc[32];
} orbx_fuse_point;

/* Replaces the search of ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th, bRight)
 * (src/ORBmatcher.cc:1108-1277; LocalMapping::SearchInNeighbors): per map point KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:
 * 705-749), the level window [nPredictedLevel - 1, nPredictedLevel], the chi-square gate on the reprojection error (5.99
 * monocular, 7.8 with mvuRight[idx] >= 0; float arithmetic as :1217-1237) and the first strict minimum of the descriptor distance
 * (:1195-1256).  kps / desc / u_right = the camera searched: mvKeysUn + mvuRight, or mvKeys / mvKeysRight of a two-camera rig
 * with u_right = NULL (the caller adds NLeft to the indices of the right camera, :1239); bounds = mnMinX .. mnMaxY;
 * inv_level_sigma2 = mvInvLevelSigma2; max_dist = TH_LOW (50).  best_idx[i] = the keypoint point i fuses into (distance <=
 * max_dist) or -1; best_dist[i]
 * (optional) = the minimum over the gated candidates, 256 if there were none.  The Replace / AddObservation / AddMapPoint
 * bookkeeping of a hit (:1259-1271) does not feed back into the search and stays with the caller, in point order.
 * Returns nFused or a negative error.
 * Fuse(KeyFrame* pKF, Sim3f& Scw, vpPoints, th, vpReplacePoint) of loop closing (:1279-1390) runs the same search without the
 * chi-square gate: pass inv_level_sigma2 = 0 for every level (e2 * 0 > 5.99 never holds).
 * ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (:1392-1592) is two such searches -- pKF1's map points in pKF2
 * (:1437-1497) and pKF2's in pKF1 (:1499-1567), no chi-square gate, max_dist = TH_HIGH (100) -- followed by the agreement
 * check vnMatch2[vnMatch1[i1]] == i1 (:1569-1583) on the caller's side (ORB_SLAM3::SearchBySim3 in csrc/ORBmatcher.h). */
int orbx_fuse_search(int device, const orbx_keypoint* kps, const uint8_t* desc, const float* u_right, int n, float min_x,
                     float min_y, float max_x, float max_y, const float* inv_level_sigma2, int nlevels,
                     const orbx_fuse_point* points, int n_points, int max_dist, int32_t* best_idx, int32_t* best_dist);

/* Stereo-fisheye frames (F.Nleft != -1): the frame holds N = n_left + n_right keypoints (mvKeys then mvKeysRight), one
 * descriptor row each in the same order, mGrid over the left and mGridRight over the right keypoints, and the stereo
 * association mvLeftToRightMatch / mvRightToLeftMatch (orbx_fisheye_stereo_match).  orbx_map_point_right carries the
 * right-camera members of MapPoint next to orbx_map_point_view (whose proj_xr is mTrackProjXR): mTrackProjYR,
 * mTrackViewCosR, mnTrackScaleLevelR (-1 = none), mbTrackInViewR. */
typedef struct orbx_map_point_right {
  float proj_yr, view_cos_r;
  int32_t predicted_level_r;
  uint8_t in_view_r, pad_[3];
} orbx_map_point_right;

/* Device-side projection for stereo-fisheye frames (Nleft != -1; the pinhole form is orbx_project_map_points_batch, same uploaded map):
 * Frame::isInFrustum runs isInFrustumChecks (src/Frame.cc:689-697, :1333-1410)
 * once per camera with KannalaBrandt8::project (src/CameraModels/KannalaBrandt8.cpp:67-86).  A camera's pose = the (mR, mt, twc) the
 * reference forms at :1342-1351 -- (mRcw, mtcw, mOw) for the left camera, (Rrl * mRcw, Rrl * mtcw + trl, mRwc * mTlr.translation()
 * + mOw) for the right one, computed by the caller in float -- plus that camera's eight KB8 parameters.  The views of both
 * cameras stay on the device for orbx_search_by_projection_fisheye_batch(map_points = NULL, map_points_right = NULL); views_out /
 * views_right_out (may be NULL) receive host copies in the matcher's input form ([n_frames][n] each).  Tolerance parity (device
 * atan2f / cosf / sinf / logf). */
typedef struct orbx_frame_pose_kb8 {
  float R[9], t[3], twc[3], kb8[8];
} orbx_frame_pose_kb8;
int orbx_project_map_points_fisheye_batch(orbx_extractor* ex, int n_frames, const orbx_frame_pose_kb8* left_poses,
                                          const orbx_frame_pose_kb8* right_poses, float min_x, float min_y, float max_x, float max_y,
                                          float viewing_cos_limit, const uint8_t* skip, orbx_map_point_view* views_out,
                                          orbx_map_point_right* views_right_out);


/* Replaces ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
 * (src/ORBmatcher.cc:41-221) for F.Nleft != -1: left-camera search, right-camera search (radius not scaled by th, :144),
 * and the assignments to the stereo partner slots (:126-132, :199-204).  occupied / match have N entries, the right
 * keypoint i at n_left + i; semantics as orbx_search_by_projection.  Returns nmatches or an error. */
int orbx_search_by_projection_fisheye(int device, const orbx_keypoint* kps, const uint8_t* desc, int n_left, int n_right,
                                      float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                                      int nlevels, const orbx_map_point_view* map_points,
                                      const orbx_map_point_right* map_points_right, int n_map_points, float th,
                                      int far_points, float th_far_points, float nnratio, const int32_t* left_to_right,
                                      const int32_t* right_to_left, uint8_t* occupied, int32_t* match);

/* Replaces the matching part of ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * (src/ORBmatcher.cc:1594-1806) for CurrentFrame.Nleft != -1: uv_right = the caller's projection of every point into the
 * right camera (2 floats per point, :1705-1706); the right search runs only when the left one found candidates
 * (:1651).  Returns nmatches after the rotation-consistency cull. */
int orbx_search_by_projection_frame_fisheye(int device, const orbx_keypoint* kps, const uint8_t* desc, int n_left,
                                            int n_right, float min_x, float min_y, float max_x, float max_y,
                                            const orbx_projected_point* points, const float* uv_right, int n_points,
                                            int check_orientation, uint8_t* occupied, int32_t* match);
/* The two stereo-fisheye flavours above on the two-camera frames of an extraction BATCH: frame f = left image first_left + f and
 * right image first_right + f of `ex`'s last batch (keypoints / descriptors stay in HBM; scale factors = the handle's).  Points of
 * frame f are [f * points_stride .. + n_points[f]) of map_points / map_points_right (points / uv_right [.][2] for the frame
 * flavour); left_to_right / right_to_left are [n_frames][cap] (mvLeftToRightMatch / mvRightToLeftMatch of every frame, cap =
 * orbx_batch_results_device's cap); occupied_in (may be NULL = all free) / occupied / match are [n_frames][2 cap], row = [left
 * keypoints | right keypoints] like the one-shot calls' N = Nleft + Nright arrays; n_matches [n_frames].  Every kernel of the
 * chain runs ONCE for all frames and both cameras with a fixed number of fixed-point rounds enqueued blindly; a frame that needs
 * more (or larger candidate / writer lists) is redone through the one-shot path: results are those of n_frames separate calls.
 * map_points == NULL && map_points_right == NULL (round 6): the views of both cameras come from the handle's last
 * orbx_project_map_points_fisheye_batch (points_stride and every n_map_points[f] must equal the uploaded map's n).
 * Returns the total number of matches or a negative error. */
int orbx_search_by_projection_fisheye_batch(orbx_extractor* ex, int first_left, int first_right, int n_frames, float min_x, float min_y,
                                            float max_x, float max_y, const orbx_map_point_view* map_points,
                                            const orbx_map_point_right* map_points_right, const int32_t* n_map_points,
                                            int points_stride, float th, int far_points, float th_far_points, float nnratio,
                                            const int32_t* left_to_right, const int32_t* right_to_left, const uint8_t* occupied_in,
                                            uint8_t* occupied, int32_t* match, int32_t* n_matches);
int orbx_search_by_projection_frame_fisheye_batch(orbx_extractor* ex, int first_left, int first_right, int n_frames, float min_x,
                                                  float min_y, float max_x, float max_y, const orbx_projected_point* points,
                                                  const float* uv_right, const int32_t* n_points, int points_stride,
                                                  int check_orientation, const uint8_t* occupied_in, uint8_t* occupied,
                                                  int32_t* match, int32_t* n_matches);


/* ---- measurement ------------------------------------------------------------------------------------ */

/* Per-kernel timing with HIP events recorded on the handle's own stream around every kernel launch (the
 * numbers bench.py's roofline object is built from).  Stages: */
#define ORBX_STAGE_RESIZE 0        /* 7 launches per extraction (one per pyramid level >= 1) */
#define ORBX_STAGE_DETECT 1
#define ORBX_STAGE_OCTREE 2
#define ORBX_STAGE_BLUR 3
#define ORBX_STAGE_SLOTS 4
#define ORBX_STAGE_DESCRIBE 5
#define ORBX_STAGE_STEREO_MATCH 6
#define ORBX_STAGE_STEREO_FILTER 7
#define ORBX_NUM_STAGES 8
/* on: 0 = off, 1 = bracket every kernel launch, 2 + s = bracket only the launches of stage s. */
int orbx_profile_enable(orbx_extractor* ex, int on);
/* Synchronises the stream, adds up the elapsed milliseconds / launch counts per stage since the last
 * collect and resets the log.  ms and launches hold ORBX_NUM_STAGES entries. */
int orbx_profile_collect(orbx_extractor* ex, double* ms, int32_t* launches);
const char* orbx_stage_name(int stage);
/* Pyramid geometry of the last configured image size: w/h per level (nlevels entries each) and the number
 * of FAST candidates / selected keypoints of image `image` per level (device counters, synchronises). */
int orbx_level_stats(orbx_extractor* ex, int image, int32_t* w, int32_t* h, int32_t* n_candidates,
                     int32_t* n_selected);

/* ---- test hooks -------------------------------------------------------------------------------------- */

/* Measurement aid: the shader clock while other work runs.  _start launches ONE wave on a private stream that spins for
 * spin_us microseconds between readings of s_memtime (shader cycles) and s_memrealtime (100 MHz); _finish waits for it and
 * returns cycles / ns = GHz of the shader clock domain during that interval (the probe handle is consumed).  bench.py
 * prints it as roofline.shader_clock_ghz instead of assuming the 2.4 GHz peak. */
int orbx_clock_probe_start(int device, int spin_us, void** probe);
int orbx_clock_probe_finish(void* probe, double* ghz);

/* Measurement aid: the device-copy rate the roofline fractions are ALSO quoted against.  Copies `bytes` (a multiple
 * of 16, >= 1 MiB) device to device `iters` times with a 16-byte-per-lane kernel and returns (bytes read + bytes written) / time in
 * GB/s -- the hardware guide measures 6.29 TB/s this way on MI355X (79 % of the 8 TB/s HBM3E peak). */
int orbx_copy_probe(int device, size_t bytes, int iters, double* gbps);

/* Runs the quadtree's host/device introsort replica (csrc/orbx_introsort.h) on the host: sorts n 64-bit
 * elements by their key bits 16..63, payload bits 0..15 ride along.  tests/ compare it with std::sort
 * (the tie order DistributeOctTree depends on, src/ORBextractor.cc:686). */
void orbx_debug_introsort(uint64_t* v, int n);
/* The wave-cooperative device version the quadtree kernel actually runs (n <= 4000). */
int orbx_debug_introsort_device(int device, uint64_t* v, int n);

/* Shrinks (>= 320 entries) or restores (any larger value) the capacity of k_detect's LDS corner + survivor list so
 * that tests can force the paths natural images rarely reach: mid-cell flushes, the corner limit and the tile-scan
 * NMS behind it. */
void orbx_debug_set_detect_list_cap(int cap);
/* Test hook: != 0 forces k_octree's global-memory candidate path (normally taken only when one (image, level) has more
 * than 16384 FAST candidates); 0 restores the register-resident path. */
void orbx_debug_set_octree_global(int on);
/* Test hook of ComputeStereoMatches' two forms (src/Frame.cc:921-1084): calls with up to max_pairs pairs (and at most 4096
 * result slots per image) run the DIRECT form -- k_stereo_band selects its keypoints from the unsorted arrays itself, no
 * k_stereo_sort launch in front --, larger ones the row-sorted form.  Default 1 (the single-frame path); 0 = never; < 0
 * restores the default.  Environment: ORBX_STEREO_DIRECT_PAIRS. */
void orbx_debug_set_stereo_direct(int max_pairs);
/* Test hook of orbx_clahe's two apply forms (cv::CLAHE::apply, Examples/Stereo/stereo_tum_vi.cc:100,142-143): 1 (default) = one
 * workgroup per interpolation cell with the cell's table in LDS where the geometry allows it, 0 = the per-pixel table gathers
 * everywhere. */
/* Test hook of the batched, device-resident association paths (orbx_fisheye_stereo_match_batch, orbx_stereo_match_batch, the
 * batched matchers): overwrites image `image` of the handle's LAST extraction batch with n keypoints / descriptors given by the
 * caller (serial-order slots, mono_index = first lapping row), so that crafted sets -- ties, empty and one-row lapping areas,
 * sizes around the kernels' tile edges -- reach the kernels that normally only see extractor output.  Synchronises the stream. */
int orbx_debug_upload_results(orbx_extractor* ex, int image, const orbx_keypoint* kps, const uint8_t* desc, int n, int mono_index);
void orbx_debug_set_clahe_cell_kernel(int on);
/* Test hook of the pre-processing plans' two cv::remap forms (src/System.cc:294-295): 1 (default) = the source footprint of
 * every 128 x 8 output tile staged through LDS (k_remap_lds) where the plan's maps allow it, 0 = the per-thread window
 * gathers (k_remap1) everywhere. */
void orbx_debug_set_remap_lds(int on);
/* Test hook of the pyramid's fused small-level launches (k_resize_tail: up to three consecutive levels of
 * ComputePyramid, src/ORBextractor.cc:1108-1145, per launch).  first_level: -1 = the library's policy, 0 = no fusion (every
 * level through k_resize), >= 2 = fuse from that level on; max_levels / band_rows: levels per launch and rows of the last
 * level per workgroup (<= 0: defaults).  Applies to handles (re)configured afterwards, i.e. to the next image SIZE a handle
 * sees.  orbx_debug_resize_plan reports the fused segments of the handle's current size (returns their number). */
void orbx_debug_set_resize_tail(int first_level, int max_levels, int band_rows);
int orbx_debug_resize_plan(const orbx_extractor* ex, int32_t* first_level, int32_t* n_levels, int32_t* n_bands, int cap);
/* Test tap of k_detect: with enable != 0 every following extraction also writes, per pyramid level, the FAST score
 * (cornerScore, 0 = not a corner) of every detectable pixel at iniThFAST -- the corner set of cv::FAST BEFORE non-max
 * suppression (src/ORBextractor.cc:810-815).  orbx_debug_score_level copies one level (w x h bytes) to the host.
 * tests/test_pin_skimage.py compares it with scikit-image's independent segment test. */
int orbx_debug_score_map(orbx_extractor* ex, int enable);
int orbx_debug_score_level(orbx_extractor* ex, int image, int level, uint8_t* dst, ptrdiff_t dst_stride);
/* The device's sinf / cosf of the descriptor steering (computeOrbDescriptor, src/ORBextractor.cc:106-107: libm cosf / sinf;
 * csrc/orbx_sincos.h restates glibc's two x86-64 ifunc variants): evaluates n angles (radians, host arrays) with the FMA
 * (fused != 0) or the SSE2 variant.  tests/ compare both with the host's libm bit for bit. */
int orbx_debug_sincos(int device, const float* angles, int n, int fused, float* sin_out, float* cos_out);

#ifdef __cplusplus
}
#endif
#endif /* ORBX_H_ */
