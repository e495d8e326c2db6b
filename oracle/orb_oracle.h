// orb_oracle.h — CPU oracle for the ORB front-end hot path.  TEST INFRASTRUCTURE ONLY.
//
// This is a from-scratch, single-threaded restatement of the *serial* semantics of the reference
// (hellovuong/ORB_SLAM3_FAST: src/ORBextractor.cc, src/ORBmatcher.cc, src/Frame.cc) and of the OpenCV 4.x
// kernels that path calls (resize, FAST, GaussianBlur, fastAtan2, norm, BFMatcher::knnMatch).
//
// PARITY UNPINNED: the reference holds no tests or golden vectors for this path, and neither the
// reference nor OpenCV can be built in this image (no OpenCV/TBB/Eigen/Boost headers, no network).
// The OpenCV kernels are restated from their published algorithm (generic C++ paths of OpenCV >= 4.5.1);
// what IS pinned: the rBRIEF pattern (sha256 + scikit-image's independent copy), the umax / quota /
// pyramid-size tables of SURVEY.md Appendix C, and the hand-derivable KATs of Appendix B.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
// The product (orb_slam3_fast_amd/) never links, imports or calls it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace orbo {

// Layout-identical to cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id) = 28 bytes.
struct KeyPoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct Image {
  int w = 0, h = 0;
  std::vector<uint8_t> px;  // row-major, stride == w
  Image() {}
  Image(int w_, int h_) : w(w_), h(h_), px((size_t)w_ * h_) {}
  uint8_t* row(int y) { return px.data() + (size_t)y * w; }
  const uint8_t* row(int y) const { return px.data() + (size_t)y * w; }
};

// ---- OpenCV kernels restated -------------------------------------------------
int cv_round(float v);
int cv_round(double v);
float fast_atan2(float y, float x);                                   // B5
void orb_sincosf(float ang, float* s, float* c);                      // B8: host libm sinf/cosf (mode 0) or a glibc model
void orb_set_sincos_mode(int mode);  // 0 = host libm, 1 = glibc FMA-variant model (default: deterministic, = the device), 2 = glibc SSE2-variant model
int orb_get_sincos_mode();
float glibc_sinf_model(float y, bool fused);
float glibc_cosf_model(float y, bool fused);
int orb_host_libm_variant();         // 1 = FMA, 2 = SSE2, 0 = no model matches the host libm
long long orb_sincos_check(uint64_t seed, long long n, int fused, float* first_bad);
void resize_linear_u8(const Image& src, Image& dst, int dw, int dh);  // B2
void resize_axis_coefs(int s, int d, bool clampX, std::vector<int>& ofs, std::vector<short>& ab, std::vector<uint8_t>* plain);  // B2: one axis's tables
// FAST-9-16 with non-max suppression on a standalone image (ROI already cut); out: x,y,score triples.
struct FastPt { int x, y, score; };
void fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold, bool nms,
              std::vector<FastPt>& out);                              // B3
int fast_corner_score16(const uint8_t* ptr, const int pixel[25], int threshold);
void gaussian_blur7(const Image& src, Image& dst, const int taps[7], int simd_vec = 0);  // B4 (simd_vec: orb_oracle.cpp)
extern const int kBlurTaps451[7];  // OpenCV >= 4.5.1 : 18,34,48,56,48,34,18
extern const int kBlurTaps440[7];  // OpenCV 4.0-4.5.0: 18,34,49,55,49,34,18

// ---- ORBextractor (reference src/ORBextractor.cc) --------------------------------------------------
struct ExtractorTables {
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> nfeat_level;
  std::vector<int> umax;  // 16 entries
};

class Extractor {
 public:
  Extractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
  // operator(): returns monoIndex, or -1 on empty image.  lap0/lap1 = vLappingArea.
  int extract(const uint8_t* img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
              std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc);

  // Same result with the fork's thread structure (one task per pyramid level and stage) -- the cpu_mt timing baseline.
  int extract_mt(const uint8_t* img, int w, int h, ptrdiff_t stride, int lap0, int lap1, std::vector<KeyPoint>& kps,
                 std::vector<uint8_t>& desc);

  // Stages, exposed for stage-level differential tests.
  void compute_pyramid(const uint8_t* img, int w, int h, ptrdiff_t stride);
  void detect_level_candidates(int level, std::vector<KeyPoint>& cand) const;  // literal per-cell FAST
  std::vector<KeyPoint> distribute_octtree(const std::vector<KeyPoint>& cand, int minX, int maxX,
                                           int minY, int maxY, int N) const;
  void compute_keypoints(std::vector<std::vector<KeyPoint>>& all) const;

  ExtractorTables t;
  std::vector<Image> pyramid;  // mvImagePyramid (no 19-px border: it is never read)
  std::vector<Image> blurred;  // per level, filled by extract() for levels with keypoints
  int nfeatures, nlevels, iniTh, minTh;
  double scaleFactor;  // the reference keeps it as double (include/ORBextractor.h:106)
  const int* blur_taps = kBlurTaps451;
  int blur_simd_vec = 0;  // 16 / 32: OpenCV 4.0 .. 4.5.0's vectorised vertical pass floors its body columns (gaussian_blur7)
};

float ic_angle(const Image& im, int cx, int cy, const std::vector<int>& umax);
void orb_descriptor(const Image& blurred, float px, float py, float angle_deg, uint8_t out[32]);
extern const int8_t kPattern[1024];

// ---- ORBmatcher / Frame matching (reference src/ORBmatcher.cc, src/Frame.cc) -----------------------
int descriptor_distance(const uint8_t* a, const uint8_t* b);

// Frame::ComputeStereoMatches (src/Frame.cc:921-1084). maxD = bf / b (intended semantics).
void compute_stereo_matches(const std::vector<Image>& pyrL, const std::vector<Image>& pyrR,
                            const std::vector<KeyPoint>& kL, const uint8_t* dL,
                            const std::vector<KeyPoint>& kR, const uint8_t* dR,
                            const std::vector<float>& scale, const std::vector<float>& inv_scale,
                            float bf, float b, std::vector<float>& uRight, std::vector<float>& depth);

// cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) + Lowe ratio (src/Frame.cc:1293-1302).
// idx2/dist2: nQ x 2 (-1 when fewer than k train rows). ratio_ok[q] = accepted by d0 < d1*0.7.
void bf_knn2(const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, std::vector<int>& idx2,
             std::vector<int>& dist2, std::vector<uint8_t>& ratio_ok);

// ---- KannalaBrandt8 + Frame::ComputeStereoFishEyeMatches tail (FLOAT: tolerance parity, see tests) ------------
// KannalaBrandt8 camera: mvParameters = fx fy cx cy k0 k1 k2 k3 and the Newton stop `precision`
// (include/CameraModels/KannalaBrandt8.h:42-57,102).
struct KB8 {
  float p[8];
  float precision = 1e-6f;
};
void kb8_project(const KB8& c, const float X[3], float uv[2]);   // src/CameraModels/KannalaBrandt8.cpp:67-86
void kb8_unproject(const KB8& c, float u, float v, float ray[3]);  // :116-147
// Right singular vector of the smallest singular value of a row-major 4x4 (what Eigen::JacobiSVD(...).matrixV().col(3)
// returns, :420-432), by one-sided Jacobi in double.  Eigen is not in this image: the reference's float JacobiSVD is
// matched to its own rounding noise only.
void smallest_right_singular_vector(const float A[16], float v[4]);
// KannalaBrandt8::TriangulateMatches (:341-417).  Returns z1 (> 0), or -1..-5 for the rejecting gate.  gate[5] (may
// be null) receives the gated quantities {cosParallax, z1, z2, err1/(5.991 sigma1), err2/(5.991 sigma2)} so that a
// tolerance test can tell borderline decisions from wrong ones.
float kb8_triangulate_matches(const KB8& c1, const KB8& c2, float u1, float v1, float u2, float v2, const float R12[9],
                              const float t12[3], float sigmaLevel, float unc, float p3D[3], float* gate,
                              const float* xh_override = nullptr, float* A_out = nullptr);
// Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1273-1331): BF 2-NN on the lapping-area rows [mono, n) of both
// eyes, Lowe 0.7, triangulation gates; serial order (a right keypoint claimed twice keeps the later left index).
// Outputs sized nL / nR / nL / 3 nL; returns nMatches, *descMatches = pairs that passed the ratio test.
int compute_stereo_fisheye_matches(const std::vector<KeyPoint>& kL, const uint8_t* dL, int monoL,
                                   const std::vector<KeyPoint>& kR, const uint8_t* dR, int monoR, const KB8& c1,
                                   const KB8& c2, const float R12[9], const float t12[3],
                                   const std::vector<float>& levelSigma2, std::vector<int>& leftToRight,
                                   std::vector<int>& rightToLeft, std::vector<float>& depth,
                                   std::vector<float>& p3D, int* descMatches, std::vector<float>* gates);

// ---- image pre-processing in front of the extractor (the gray conversion and the input resize) ------------
// cv::cvtColor(src, dst, COLOR_{RGB,BGR,RGBA,BGRA}2GRAY) for 8U (src/Tracking.cc:1394-1412,1441-1459,1481-1499):
// fixed point with 15-bit coefficients RY15 = 9798, GY15 = 19235, BY15 = 3735, (sum + 16384) >> 15 (OpenCV >= 3.4.2 /
// 4.x; variant 14 = the older 14-bit set 4899 / 9617 / 1868).  cn = 3 or 4 interleaved channels, rgb != 0: R first.
void cvt_gray_u8(const uint8_t* src, int w, int h, ptrdiff_t src_stride, int cn, bool rgb, uint8_t* dst, ptrdiff_t dst_stride,
                 int variant = 15);
// cv::resize(src, dst, Size(dw, dh)) with INTER_LINEAR on 8UC1 / 8UC3 / 8UC4 (src/System.cc:297-298,369-370,437-438;
// src/Settings.cc:330-343): the B2 arithmetic of resize_linear_u8 on every interleaved channel.
void resize_linear_u8c(const uint8_t* src, int sw, int sh, ptrdiff_t src_stride, int cn, uint8_t* dst, int dw, int dh,
                       ptrdiff_t dst_stride);

// cv::remap(src, dst, mapx, mapy, INTER_LINEAR) with CV_32FC1 maps, BORDER_CONSTANT 0, 8UC1 (src/System.cc:294-295): the
// stereo rectification of every raw frame.  map_stride in floats.
void remap_linear_u8(const uint8_t* src, int sw, int sh, ptrdiff_t src_stride, const float* mapx, const float* mapy,
                     ptrdiff_t map_stride, uint8_t* dst, int dw, int dh, ptrdiff_t dst_stride);
// cv::createCLAHE(clip_limit, Size(tiles_x, tiles_y))->apply(src, dst), 8UC1 (Examples/Stereo/stereo_tum_vi.cc:100,142-143).
void clahe_u8(const uint8_t* src, int w, int h, ptrdiff_t src_stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t* dst, ptrdiff_t dst_stride);

// ---- Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:853-919) --------------------------------------------
// cv::undistortPoints(src, dst, K, distCoeffs, noArray(), P = K) with its default TermCriteria(MAX_ITER, 5, 0.01): five
// fixed-point iterations of the inverse distortion in double, then x' = fx x + cx (OpenCV calib3d undistort.dispatch.cpp,
// restated from the published algorithm -- not pinnable here, see the header note).  K = fx fy cx cy (float, as
// Pinhole::toK()); dist = up to 14 OpenCV coefficients k1 k2 p1 p2 k3 k4 k5 k6 s1..s4 tx ty (ORB-SLAM3 passes 4 or 5;
// tx / ty must be 0).  xy_in / xy_out: n interleaved float pairs.
void undistort_points(const float* xy_in, int n, const float K[4], const float* dist, int n_dist, float* xy_out);
// mvKeysUn from mvKeys (identity when dist[0] == 0, :854-857).
void undistort_keypoints(const std::vector<KeyPoint>& kps, const float K[4], const float* dist, int n_dist,
                         std::vector<KeyPoint>& out);
// mnMinX, mnMinY, mnMaxX, mnMaxY (:887-919).
void compute_image_bounds(int cols, int rows, const float K[4], const float* dist, int n_dist, float bounds[4]);

// Frame grid (src/Frame.cc:520-547,765-844).
struct FrameGrid {
  float minX, minY, maxX, maxY, invW, invH;
  std::vector<std::vector<int>> cells;  // [ix*48+iy]
  void build(const std::vector<KeyPoint>& kps, float minX, float minY, float maxX, float maxY);
  std::vector<int> features_in_area(const std::vector<KeyPoint>& kps, float x, float y, float r,
                                    int minLevel, int maxLevel) const;
};

// ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:618-764), serial i1 order.
int search_for_initialization(const std::vector<KeyPoint>& k1, const uint8_t* d1,
                              const std::vector<KeyPoint>& k2, const uint8_t* d2, const FrameGrid& g2,
                              std::vector<float>& prevMatched /*2*n1 in/out*/,
                              std::vector<int>& matches12, int windowSize, float nnratio,
                              bool checkOri);

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints)
// (src/ORBmatcher.cc:41-221), pinhole case (F.Nleft == -1), serial iMP order.  MapPointView carries the MapPoint
// members that routine reads (mbTrackInView, mTrackDepth, isBad(), mnTrackScaleLevel, mTrackViewCos,
// mTrackProjX/Y/XR, GetDescriptor(), Observations() > 0).
struct MapPointView {
  float proj_x, proj_y, proj_xr, view_cos, track_depth;
  int32_t predicted_level;
  uint8_t in_view, bad, has_observations, pad_;
  uint8_t desc[32];
};
static_assert(sizeof(MapPointView) == 60, "POD layout shared with orbx_map_point_view");
// occupied[i] != 0  <=>  F.mvpMapPoints[i] != NULL && ->Observations() > 0 (in/out: assignments update it).
// match[i] = index of the map point assigned to keypoint i by this call, or -1.  Returns nmatches.
// Frame::isInFrustum (src/Frame.cc:632-690, pinhole case Nleft == -1) with MapPoint::PredictScale (src/MapPoint.cc:559-573):
// the members SearchByProjection reads back from the MapPoint, as a MapPointView.  pose = mRcw (row-major), mtcw, mOw, the Pinhole
// parameters, mbf.  Float arithmetic in the reference's expression order (Eigen's 3x3 product and dot / norm reductions evaluate
// left to right; no contraction).  margin (may be NULL) receives how far the decisive quantities are from their gates, for the
// tolerance tests of the device kernel: [0] min over the gates passed / failed of |value - threshold| / max(1, |threshold|),
// [1] distance of log(ratio) / logScaleFactor from the nearest integer (ceil's rounding point).
struct FramePose {
  float Rcw[9], tcw[3], Ow[3], fx, fy, cx, cy, bf;
};
MapPointView is_in_frustum(const FramePose& T, const float P[3], const float Pn[3], float minDistance, float maxDistance,
                           float minX, float minY, float maxX, float maxY, float viewingCosLimit, float logScaleFactor,
                           int nlevels, double margin[2]);
// Frame::isInFrustumChecks (src/Frame.cc:1333-1410) for ONE camera of a stereo-fisheye frame: pose = (mR, mt, twc) of that camera --
// (mRcw, mtcw, mOw) for the left one, (Rrl * mRcw, Rrl * mtcw + trl, mRwc * mTlr.translation() + mOw) for the right one, formed by
// the caller in float as :1342-1351 does -- and its KannalaBrandt8 parameters; the projection is KannalaBrandt8::project
// (src/CameraModels/KannalaBrandt8.cpp:67-86).  Result as a MapPointView (proj_x / proj_y / view_cos / track_depth /
// predicted_level / in_view of THAT camera); margin as is_in_frustum.
struct FramePoseKB8 {
  float R[9], t[3], Ow[3], kb8[8];
};
MapPointView is_in_frustum_kb8(const FramePoseKB8& T, const float P[3], const float Pn[3], float minDistance, float maxDistance,
                               float minX, float minY, float maxX, float maxY, float viewingCosLimit, float logScaleFactor,
                               int nlevels, double margin[2]);
int search_by_projection_map(const std::vector<KeyPoint>& kpsUn, const uint8_t* desc, const float* uRight,
                             const FrameGrid& grid, const std::vector<float>& scaleFactors,
                             const std::vector<MapPointView>& mps, float th, bool bFarPoints, float thFarPoints,
                             float nnratio, std::vector<uint8_t>& occupied, std::vector<int>& match);

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
// (src/ORBmatcher.cc:1594-1806), pinhole case.  The pose / camera projection of the reference (Eigen float code,
// :1606-1636) stays on the caller's side; ProjectedPoint carries its results per LastFrame point: uv, the
// right-image coordinate ur = u - mbf*invz, radius = th * mvScaleFactors[nLastOctave], the level window chosen by
// bForward / bBackward, the last-frame keypoint angle, the MapPoint descriptor and Observations() > 0.
struct ProjectedPoint {
  float u, v, ur, radius, angle;
  int32_t min_level, max_level;
  uint8_t valid, has_observations, pad_[2];
  uint8_t desc[32];
};
static_assert(sizeof(ProjectedPoint) == 64, "POD layout shared with orbx_projected_point");
// The projection block of that matcher (src/ORBmatcher.cc:1606-1648) for ONE LastFrame point: x3Dc = Tcw * x3Dw with Tcw a
// Sophus::SE3f, i.e. the unit-quaternion sandwich of Thirdparty/Sophus/sophus/so3.hpp:358-366 (uv = q.vec x p; uv += uv;
// p + q.w * uv + q.vec x uv) plus the translation (se3.hpp:321-324) -- NOT a matrix product --; invzc = 1.0 / z in double,
// narrowed; `invzc < 0` skip; Pinhole::project; the image-bounds skips; radius = th * mvScaleFactors[nLastOctave]; the level
// window by bForward / bBackward (direction 1 / 2; decided by the caller from tlc(2) and mb, :1611-1612); ur = u - mbf * invzc
// (:1669).  Float arithmetic in Eigen's coefficient order (cross product a1 b2 - a2 b1, ..), no contraction.
// margin (may be NULL): min |value - threshold| / max(1, |threshold|) over the gates evaluated, for the device kernel's tolerance tests.
struct FramePoseQ {
  float q[4] /* x y z w */, t[3], fx, fy, cx, cy, bf;
  int32_t direction;   // 0: neither, 1: bForward, 2: bBackward
};
ProjectedPoint project_last_frame_point(const FramePoseQ& T, const float Pw[3], int lastOctave, float lastAngle, float th,
                                        const std::vector<float>& scaleFactors, float minX, float minY, float maxX, float maxY,
                                        double* margin);
// match[i2] = index of the LastFrame point assigned to CurrentFrame keypoint i2, or -1.  Returns nmatches.
int search_by_projection_frame(const std::vector<KeyPoint>& kpsUn, const uint8_t* desc, const float* uRight,
                               const FrameGrid& grid, const std::vector<ProjectedPoint>& pts, bool checkOri,
                               std::vector<uint8_t>& occupied, std::vector<int>& match);

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
// (src/ORBmatcher.cc:1808-1918, relocalisation).  pts[i] = pKF's i-th map point after the caller's projection and gates
// (:1823-1851): valid, (u, v), radius = th * mvScaleFactors[nPredictedLevel], level window (nPredictedLevel -/+ 1), angle =
// pKF->mvKeysUn[i].angle, descriptor; ur / has_observations are not read.  occupied[i2] <=> CurrentFrame.mvpMapPoints[i2] != NULL
// (in/out; every assignment occupies, culled slots are free again).  match[i2] = point index or -1.  Returns nmatches.
int search_by_projection_keyframe(const std::vector<KeyPoint>& kpsUn, const uint8_t* desc, const FrameGrid& grid,
                                  const std::vector<ProjectedPoint>& pts, int ORBdist, bool checkOri,
                                  std::vector<uint8_t>& occupied, std::vector<int>& match);

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:886-1106) for
// single-camera key frames.  Feature vectors as CSR (ascending node ids); k1 / k2 = mvKeysUn; hasMP = GetMapPoint(idx) != NULL;
// uRight = mvuRight (nullptr: no stereo observations); scaleFactors2 / levelSigma2_2 = pKF2's tables; ep = pKF2's projection
// of pKF1's camera centre (:897-901); F12 row-major = K1^-T [t12]x R12 K2^-1 (Pinhole.cpp:130-133).  vMatches12[idx1] = idx2
// or -1 (vMatchedPairs = its non-negative entries in ascending idx1).  Returns nmatches.
int search_for_triangulation(const std::vector<uint32_t>& nodes1, const std::vector<int>& start1, const std::vector<uint32_t>& feat1,
                             const std::vector<KeyPoint>& k1, const uint8_t* d1, const uint8_t* hasMP1, const float* uRight1,
                             const std::vector<uint32_t>& nodes2, const std::vector<int>& start2, const std::vector<uint32_t>& feat2,
                             const std::vector<KeyPoint>& k2, const uint8_t* d2, const uint8_t* hasMP2, const float* uRight2,
                             const std::vector<float>& scaleFactors2, const std::vector<float>& levelSigma2_2, const float ep[2],
                             const float F12[9], bool bOnlyStereo, bool bCoarse, bool checkOri, std::vector<int>& vMatches12);

// ORBmatcher::SearchForTriangulation for two-camera rigs (pKF1->mpCamera2 && pKF2->mpCamera2; src/ORBmatcher.cc:906-923,1007-1064):
// k1 / k2 = mvKeys | mvKeysRight (nLeft = KeyFrame::NLeft), the epipolar test is KannalaBrandt8::epipolarConstrain =
// TriangulateMatches(...) > 0.0001f (src/CameraModels/KannalaBrandt8.cpp:240-250) with (R12, t12, cameras) picked by which eye each
// of the two features sits in.  TriRig: cam[0..3] = pKF1->mpCamera, pKF1->mpCamera2, pKF2->mpCamera, pKF2->mpCamera2; R / t [0..3] =
// Tll = T1w * Tw2, Tlr = T1w * Twr2, Trl = Tr1w * Tw2, Trr = Tr1w * Twr2 (rotation row-major, translation).  FLOAT: tolerance
// parity like compute_stereo_fisheye_matches; borderline[idx1] (may be null) = some evaluated candidate's gated quantity lies
// within the tolerance of tests/test_fisheye.py of its threshold.
struct TriRig {
  float cam[4][8];
  float precision;
  float R[4][9], t[4][3];
};
int search_for_triangulation_rig(const std::vector<uint32_t>& nodes1, const std::vector<int>& start1, const std::vector<uint32_t>& feat1,
                                 const std::vector<KeyPoint>& k1, const uint8_t* d1, const uint8_t* hasMP1, int nLeft1,
                                 const std::vector<uint32_t>& nodes2, const std::vector<int>& start2, const std::vector<uint32_t>& feat2,
                                 const std::vector<KeyPoint>& k2, const uint8_t* d2, const uint8_t* hasMP2, int nLeft2,
                                 const std::vector<float>& levelSigma2_1, const std::vector<float>& levelSigma2_2, const TriRig& rig,
                                 bool bOnlyStereo, bool bCoarse, bool checkOri, std::vector<int>& vMatches12,
                                 std::vector<uint8_t>* borderline);

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:766-884).  valid =
// the feature holds a good map point (and lies below mvKeysUn.size() for two-camera rigs); vMatches12[idx1] = idx2 or -1.
int search_by_bow_keyframes(const std::vector<uint32_t>& nodes1, const std::vector<int>& start1, const std::vector<uint32_t>& feat1,
                            const uint8_t* d1, const float* angle1, const uint8_t* valid1, int n1, const std::vector<uint32_t>& nodes2,
                            const std::vector<int>& start2, const std::vector<uint32_t>& feat2, const uint8_t* d2, const float* angle2,
                            const uint8_t* valid2, int n2, float nnratio, bool checkOri, std::vector<int>& vMatches12);

// ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th, bRight) (src/ORBmatcher.cc:1108-1277): the per-point
// search (:1195-1256).  FusePoint = a map point after the caller's projection and gates (:1141-1192): valid, uv, ur = u - bf * invz,
// radius = th * mvScaleFactors[nPredictedLevel], nPredictedLevel, GetDescriptor().  kps / desc / uRight = the camera searched
// (mvKeysUn + mvuRight; mvKeys or mvKeysRight of a two-camera rig).  bestIdx[i] = the keypoint point i fuses into (bestDist <=
// maxDist = TH_LOW) or -1; bestDist[i] = the minimum over the gated candidates (256: none).  Returns nFused (:1262: counted for every hit).
// The map-point bookkeeping that follows a hit (Replace / AddObservation / AddMapPoint, :1259-1271) does not feed back into the
// search and stays with the caller.
struct FusePoint {
  float u, v, ur, radius;
  int32_t predicted_level;
  uint8_t valid, pad_[3];
  uint8_t desc[32];
};
static_assert(sizeof(FusePoint) == 56, "POD layout shared with orbx_fuse_point");
int fuse_search(const std::vector<KeyPoint>& kps, const uint8_t* desc, const float* uRight, const FrameGrid& grid,
                const std::vector<float>& invLevelSigma2, const std::vector<FusePoint>& pts, int maxDist,
                std::vector<int>& bestIdx, std::vector<int>& bestDist);

// ---- stereo-fisheye (F.Nleft != -1) branches of the two SearchByProjection matchers ---------------------------------------
// The frame holds N = nLeft + nRight keypoints (mvKeys then mvKeysRight), descriptors in the same order, two grids
// (mGrid over the left keypoints, mGridRight over the right ones with indices relative to mvKeysRight), and the stereo
// association mvLeftToRightMatch / mvRightToLeftMatch.  MapPointRight carries the right-camera members
// (mTrackProjXR = MapPointView::proj_xr, mTrackProjYR, mTrackViewCosR, mnTrackScaleLevelR, mbTrackInViewR).
struct MapPointRight {
  float proj_yr, view_cos_r;
  int32_t predicted_level_r;
  uint8_t in_view_r, pad_[3];
};
static_assert(sizeof(MapPointRight) == 16, "POD layout shared with orbx_map_point_right");
// src/ORBmatcher.cc:41-221 with F.Nleft != -1.  occupied / match have N entries (right slots at nLeft + i).
int search_by_projection_map_fisheye(const std::vector<KeyPoint>& kps, const uint8_t* desc, int nLeft, const FrameGrid& gridL,
                                     const FrameGrid& gridR, const std::vector<float>& scaleFactors,
                                     const std::vector<MapPointView>& mps, const std::vector<MapPointRight>& mpsR, float th,
                                     bool bFarPoints, float thFarPoints, float nnratio, const std::vector<int>& leftToRight,
                                     const std::vector<int>& rightToLeft, std::vector<uint8_t>& occupied,
                                     std::vector<int>& match);
// src/ORBmatcher.cc:1594-1806 with CurrentFrame.Nleft != -1: uvRight = the projection of each point into the right camera
// (2 floats per point, :1705-1706).
int search_by_projection_frame_fisheye(const std::vector<KeyPoint>& kps, const uint8_t* desc, int nLeft, const FrameGrid& gridL,
                                       const FrameGrid& gridR, const std::vector<ProjectedPoint>& pts, const float* uvRight,
                                       bool checkOri, std::vector<uint8_t>& occupied, std::vector<int>& match);

// ---- bag of words (Thirdparty/DBoW2) --------------------------------------------------------------------
// The vocabulary tree as TemplatedVocabulary::loadFromTextFile leaves it (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:
// 1338-1421): node 0 is the root, nodes 1.. in file order with parent[i] < i, a node's children in file order, words
// numbered in the file order of the leaves.  The ORB vocabulary file itself (ORBvoc.txt, 10^6 words) is not in the reference
// tree; every test runs on synthetic trees.
struct Vocabulary {
  int k = 0, L = 0, scoring = 0, weighting = 0;  // ScoringType / WeightingType of BowVector.h:39-56
  std::vector<int> parent, childStart, children, wordId;  // childStart has n + 1 entries; wordId -1 for inner nodes
  std::vector<uint8_t> desc;                              // n x 32
  std::vector<double> weight;
  int nWords = 0;
  bool is_leaf(int n) const { return childStart[n + 1] == childStart[n]; }
  // build from the per-node file columns (parent, isLeaf, descriptor, weight); entry 0 = root (ignored columns)
  void build(int k_, int L_, int scoring_, int weighting_, int n, const int* parent_, const uint8_t* isLeaf,
             const uint8_t* desc_, const double* weight_);
  bool load_text(const char* path);   // the ORBvoc.txt format of loadFromTextFile
  bool save_text(const char* path) const;
};
// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (:1202-1250): descend by first-minimum Hamming
// distance among the children; *nid = the node of the path at level L - levelsup (root if that is <= 0).  A leaf above
// that level leaves *nid uninitialised in the reference; here it is the leaf itself.
void bow_transform_one(const Vocabulary& v, const uint8_t* d, int levelsup, int& wordId, double& weight, int& nodeId);
// TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup) (:1125-1188) as Frame::ComputeBoW calls it
// (src/Frame.cc:846-851, levelsup 4): BowVector = ascending (word, value) with the map's sequential += in feature order and
// the scoring's normalisation (BowVector.cpp:58-80); FeatureVector = ascending node id -> feature indices in order.
void bow_transform(const Vocabulary& v, const uint8_t* desc, int n, int levelsup, std::vector<uint32_t>& words,
                   std::vector<double>& values, std::vector<uint32_t>& nodes, std::vector<int>& nodeStart,
                   std::vector<uint32_t>& features);
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:230-404).  Feature vectors as CSR
// (ascending node ids); kfValid[i] = the keyframe feature holds a good map point; angles = kp.angle of the concatenated
// (left | right) keypoints; nLeftF = F.Nleft (-1: monocular / rectified).  match[iF] = keyframe feature index or -1.
int search_by_bow(const std::vector<uint32_t>& kfNodes, const std::vector<int>& kfStart, const std::vector<uint32_t>& kfFeat,
                  const uint8_t* kfDesc, const float* kfAngle, const uint8_t* kfValid, const std::vector<uint32_t>& fNodes,
                  const std::vector<int>& fStart, const std::vector<uint32_t>& fFeat, const uint8_t* fDesc, const float* fAngle,
                  int nF, int nLeftF, float nnratio, bool checkOri, std::vector<int>& match);

}  // namespace orbo
