// ORBmatcher.h — C++ host mirror of the hot-path part of ORB_SLAM3::ORBmatcher and of the two Frame
// stereo-association members, over the C ABI of include/orbx.h.
//
// Reference: include/ORBmatcher.h:36-101, src/ORBmatcher.cc:618-764,1920-1973 (SearchForInitialization,
// ComputeThreeMaxima, DescriptorDistance); src/Frame.cc:921-1084 (ComputeStereoMatches),
// src/Frame.cc:1273-1331 (ComputeStereoFishEyeMatches: brute-force kNN + KannalaBrandt8 triangulation).
//
// The reference's methods take Frame&; Frame itself (poses, map points, IMU) is out of scope, so the
// mirror takes the few Frame members those methods read (FrameView).  INTEGRATION.md shows the three-line
// forwarding overloads a maintainer adds to keep `matcher.SearchForInitialization(F1, F2, ...)` call sites.
#ifndef ORBX_SHIM_ORBMATCHER_H
#define ORBX_SHIM_ORBMATCHER_H

#include <algorithm>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ORBextractor.h"

namespace ORB_SLAM3 {

// The members of Frame that SearchForInitialization reads: mvKeysUn, mDescriptors and the image bounds
// mnMinX/mnMinY/mnMaxX/mnMaxY used by the 64x48 feature grid (include/Frame.h:249-272,314-319).
struct FrameView {
  const ocv::KeyPoint* mvKeysUn = nullptr;
  const uint8_t* mDescriptors = nullptr;  // N x 32, continuous
  int N = 0;
  float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  const float* mvuRight = nullptr;        // optional (SearchByProjection stereo-consistency gate)
  const float* mvScaleFactors = nullptr;  // nLevels entries (SearchByProjection)
  int nLevels = 0;
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50;    // src/ORBmatcher.cc:35-37
  static const int TH_HIGH = 100;
  static const int HISTO_LENGTH = 30;

  ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0)
      : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}

  // src/ORBmatcher.cc:1959-1973 (two 32-byte descriptor rows).
  static int DescriptorDistance(const ocv::Mat& a, const ocv::Mat& b) { return orbx_hamming256(a.ptr(0), b.ptr(0)); }
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return orbx_hamming256(a, b); }

  // src/ORBmatcher.cc:618-764, serial-order semantics.
  int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<ocv::Point2f>& vbPrevMatched,
                              std::vector<int>& vnMatches12, int windowSize = 10) {
    if ((int)vbPrevMatched.size() != F1.N) throw std::invalid_argument("vbPrevMatched.size() != F1.N");
    vnMatches12.assign(F1.N, -1);
    static_assert(sizeof(ocv::Point2f) == 8, "Point2f is two floats");
    const int n = orbx_search_for_initialization(
        device_, reinterpret_cast<const orbx_keypoint*>(F1.mvKeysUn), F1.mDescriptors, F1.N,
        reinterpret_cast<const orbx_keypoint*>(F2.mvKeysUn), F2.mDescriptors, F2.N, F2.mnMinX, F2.mnMinY, F2.mnMaxX,
        F2.mnMaxY, reinterpret_cast<float*>(vbPrevMatched.data()), vnMatches12.data(), windowSize, mfNNratio,
        mbCheckOrientation ? 1 : 0);
    if (n < 0) throw std::runtime_error(std::string("SearchForInitialization: ") + orbx_last_error());
    return n;
  }

  // The reference's own call shape -- matcher.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize),
  // include/ORBmatcher.h:97-101, called from Tracking::MonocularInitialization (src/Tracking.cc:2438-2440) -- for any
  // Frame-like type that exposes the members the routine reads: mvKeysUn (std::vector<cv::KeyPoint>), mDescriptors
  // (N x 32 CV_8U Mat, continuous), N, and the image bounds mnMinX / mnMinY / mnMaxX / mnMaxY (static members of Frame,
  // include/Frame.h:314-319).  ORB_SLAM3::Frame itself satisfies this, so the call site compiles unchanged.
  template <class FrameT, class = decltype(std::declval<const FrameT&>().mvKeysUn.data())>
  int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<ocv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10) {
    return SearchForInitialization(ViewOf(F1), ViewOf(F2), vbPrevMatched, vnMatches12, windowSize);
  }
  template <class FrameT>
  static FrameView ViewOf(const FrameT& F) {
    if (F.N > 1 && (size_t)F.mDescriptors.step != 32) throw std::invalid_argument("mDescriptors must be a continuous N x 32 matrix");
    if ((int)F.mvKeysUn.size() < F.N || F.mDescriptors.rows < F.N) throw std::invalid_argument("Frame holds fewer than N features");
    FrameView v;
    v.mvKeysUn = F.mvKeysUn.data();
    v.mDescriptors = F.mDescriptors.ptr(0);
    v.N = F.N;
    v.mnMinX = F.mnMinX; v.mnMinY = F.mnMinY; v.mnMaxX = F.mnMaxX; v.mnMaxY = F.mnMaxY;
    return v;
  }

  // SearchByProjection(Frame& F, const vector<MapPoint*>&, th, bFarPoints, thFarPoints), src/ORBmatcher.cc:41-221,
  // pinhole case.  vpMapPoints[i] is given as the POD view of the members the routine reads; occupied[i] != 0 <=>
  // F.mvpMapPoints[i] holds a point with Observations() > 0 (updated); match[i] = map point index newly assigned to
  // keypoint i, or -1 (the caller stores F.mvpMapPoints[i] = vpMapPoints[match[i]]).
  int SearchByProjection(const FrameView& F, const std::vector<orbx_map_point_view>& vpMapPoints,
                         std::vector<uint8_t>& occupied, std::vector<int>& match, const float th = 3,
                         const bool bFarPoints = false, const float thFarPoints = 50.0f) {
    match.assign(F.N, -1);
    const int n = orbx_search_by_projection(
        device_, reinterpret_cast<const orbx_keypoint*>(F.mvKeysUn), F.mDescriptors, F.mvuRight, F.N, F.mnMinX, F.mnMinY,
        F.mnMaxX, F.mnMaxY, F.mvScaleFactors, F.nLevels, vpMapPoints.data(), (int)vpMapPoints.size(), th,
        bFarPoints ? 1 : 0, thFarPoints, mfNNratio, occupied.data(), match.data());
    if (n < 0) throw std::runtime_error(std::string("SearchByProjection: ") + orbx_last_error());
    return n;
  }

  // Matching part of SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono),
  // src/ORBmatcher.cc:1594-1806, pinhole case: the caller keeps the pose / camera projection (:1606-1648) and passes
  // one orbx_projected_point per LastFrame point.
  int SearchByProjection(const FrameView& CurrentFrame, const std::vector<orbx_projected_point>& lastFramePoints,
                         std::vector<uint8_t>& occupied, std::vector<int>& match) {
    match.assign(CurrentFrame.N, -1);
    const int n = orbx_search_by_projection_frame(
        device_, reinterpret_cast<const orbx_keypoint*>(CurrentFrame.mvKeysUn), CurrentFrame.mDescriptors,
        CurrentFrame.mvuRight, CurrentFrame.N, CurrentFrame.mnMinX, CurrentFrame.mnMinY, CurrentFrame.mnMaxX,
        CurrentFrame.mnMaxY, lastFramePoints.data(), (int)lastFramePoints.size(), mbCheckOrientation ? 1 : 0,
        occupied.data(), match.data());
    if (n < 0) throw std::runtime_error(std::string("SearchByProjection: ") + orbx_last_error());
    return n;
  }

  // Matching part of the relocalisation matcher SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF,
  // const set<MapPoint*>& sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1808-1918 (Tracking::Relocalization,
  // src/Tracking.cc:3631-3632,3645-3646): the caller keeps the projection and its gates (:1823-1851) and passes one
  // orbx_projected_point per map point of pKF (valid = 0 for the skipped ones; radius = th * mvScaleFactors[nPredictedLevel],
  // levels nPredictedLevel -/+ 1, angle = pKF->mvKeysUn[i].angle).  occupied[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2] != NULL
  // (updated); match[i2] = index into pKF's map points or -1.
  int SearchByProjection(const FrameView& CurrentFrame, const std::vector<orbx_projected_point>& keyFramePoints,
                         const int ORBdist, std::vector<uint8_t>& occupied, std::vector<int>& match) {
    match.assign(CurrentFrame.N, -1);
    const int n = orbx_search_by_projection_keyframe(
        device_, reinterpret_cast<const orbx_keypoint*>(CurrentFrame.mvKeysUn), CurrentFrame.mDescriptors, CurrentFrame.N,
        CurrentFrame.mnMinX, CurrentFrame.mnMinY, CurrentFrame.mnMaxX, CurrentFrame.mnMaxY, keyFramePoints.data(),
        (int)keyFramePoints.size(), ORBdist, mbCheckOrientation ? 1 : 0, occupied.data(), match.data());
    if (n < 0) throw std::runtime_error(std::string("SearchByProjection: ") + orbx_last_error());
    return n;
  }

  // The search of Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th, bRight), src/ORBmatcher.cc:1108-1277
  // (LocalMapping::SearchInNeighbors): KF = the members of the key frame's searched camera (mvKeysUn or mvKeys / mvKeysRight,
  // mDescriptors rows of that camera, mvuRight, bounds), mvInvLevelSigma2, one orbx_fuse_point per map point after the caller's
  // projection and gates (:1141-1192).  bestIdx[i] = keypoint the point fuses into or -1: the caller then runs the reference's
  // bookkeeping in point order -- `MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[i]); if (pMPinKF) {...Replace...} else
  // {pMP->AddObservation(pKF, bestIdx[i]); pKF->AddMapPoint(pMP, bestIdx[i]);}` (:1259-1271).  Returns nFused.
  int Fuse(const FrameView& KF, const std::vector<float>& mvInvLevelSigma2, const std::vector<orbx_fuse_point>& vpMapPoints,
           std::vector<int>& bestIdx) {
    bestIdx.assign(vpMapPoints.size(), -1);
    const int n = orbx_fuse_search(device_, reinterpret_cast<const orbx_keypoint*>(KF.mvKeysUn), KF.mDescriptors, KF.mvuRight, KF.N,
                                   KF.mnMinX, KF.mnMinY, KF.mnMaxX, KF.mnMaxY, mvInvLevelSigma2.data(), (int)mvInvLevelSigma2.size(),
                                   vpMapPoints.data(), (int)vpMapPoints.size(), TH_LOW, bestIdx.data(), nullptr);
    if (n < 0) throw std::runtime_error(std::string("Fuse: ") + orbx_last_error());
    return n;
  }

  // SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const Sim3f& S12, th), src/ORBmatcher.cc:1392-1592
  // (LoopClosing): two Fuse-type searches without the chi-square gate and with TH_HIGH -- pKF1's map points in pKF2 (:1437-1497),
  // pKF2's in pKF1 (:1499-1567) -- and the agreement check (:1569-1583).  points1in2[i1] = map point i1 of pKF1 after the caller's
  // Sim3 projection into pKF2 and its gates (valid = 0 for !pMP, vbAlreadyMatched1[i1], isBad(), negative depth, outside the image,
  // distance outside the invariance range); points2in1 likewise.  vnMatch12[i1] = the feature of pKF2 whose map point
  // vpMatches12[i1] receives, or -1.  Returns nFound.
  int SearchBySim3(const FrameView& KF1, const FrameView& KF2, const std::vector<orbx_fuse_point>& points1in2,
                   const std::vector<orbx_fuse_point>& points2in1, std::vector<int>& vnMatch12) {
    auto levels = [](const FrameView& KF) {
      int m = 0;
      for (int i = 0; i < KF.N; i++) m = std::max(m, KF.mvKeysUn[i].octave);
      return m + 1;
    };
    auto search = [&](const FrameView& KF, const std::vector<orbx_fuse_point>& pts, std::vector<int>& best) {
      const std::vector<float> noGate((size_t)levels(KF), 0.0f);
      best.assign(pts.size(), -1);
      const int n = orbx_fuse_search(device_, reinterpret_cast<const orbx_keypoint*>(KF.mvKeysUn), KF.mDescriptors, nullptr, KF.N, KF.mnMinX,
                                     KF.mnMinY, KF.mnMaxX, KF.mnMaxY, noGate.data(), (int)noGate.size(), pts.data(), (int)pts.size(),
                                     TH_HIGH, best.data(), nullptr);
      if (n < 0) throw std::runtime_error(std::string("SearchBySim3: ") + orbx_last_error());
    };
    std::vector<int> vnMatch1, vnMatch2;
    search(KF2, points1in2, vnMatch1);
    search(KF1, points2in1, vnMatch2);
    vnMatch12.assign(points1in2.size(), -1);
    int nFound = 0;
    for (size_t i1 = 0; i1 < vnMatch1.size(); i1++) {
      const int idx2 = vnMatch1[i1];
      if (idx2 >= 0 && idx2 < (int)vnMatch2.size() && vnMatch2[idx2] == (int)i1) {
        vnMatch12[i1] = idx2;
        nFound++;
      }
    }
    return nFound;
  }

  // The same two searches for stereo-fisheye frames (F.Nleft != -1, src/ORBmatcher.cc:41-221 / :1594-1806): F holds
  // N = Nleft + Nright keypoints (mvKeys then mvKeysRight; FrameView::mvKeysUn points at that array, N at the total),
  // vpMapPointsRight the right-camera members of every MapPoint, and the stereo association of the frame.
  int SearchByProjection(const FrameView& F, int Nleft, const std::vector<orbx_map_point_view>& vpMapPoints,
                         const std::vector<orbx_map_point_right>& vpMapPointsRight, const std::vector<int>& mvLeftToRightMatch,
                         const std::vector<int>& mvRightToLeftMatch, std::vector<uint8_t>& occupied, std::vector<int>& match,
                         const float th = 3, const bool bFarPoints = false, const float thFarPoints = 50.0f) {
    if (vpMapPoints.size() != vpMapPointsRight.size()) throw std::invalid_argument("one right-camera record per map point");
    match.assign(F.N, -1);
    const int n = orbx_search_by_projection_fisheye(
        device_, reinterpret_cast<const orbx_keypoint*>(F.mvKeysUn), F.mDescriptors, Nleft, F.N - Nleft, F.mnMinX, F.mnMinY,
        F.mnMaxX, F.mnMaxY, F.mvScaleFactors, F.nLevels, vpMapPoints.data(), vpMapPointsRight.data(), (int)vpMapPoints.size(), th,
        bFarPoints ? 1 : 0, thFarPoints, mfNNratio, mvLeftToRightMatch.data(), mvRightToLeftMatch.data(), occupied.data(),
        match.data());
    if (n < 0) throw std::runtime_error(std::string("SearchByProjection: ") + orbx_last_error());
    return n;
  }
  int SearchByProjection(const FrameView& CurrentFrame, int Nleft, const std::vector<orbx_projected_point>& lastFramePoints,
                         const std::vector<float>& uvRight, std::vector<uint8_t>& occupied, std::vector<int>& match) {
    if (uvRight.size() != 2 * lastFramePoints.size()) throw std::invalid_argument("two right-camera coordinates per point");
    match.assign(CurrentFrame.N, -1);
    const int n = orbx_search_by_projection_frame_fisheye(
        device_, reinterpret_cast<const orbx_keypoint*>(CurrentFrame.mvKeysUn), CurrentFrame.mDescriptors, Nleft,
        CurrentFrame.N - Nleft, CurrentFrame.mnMinX, CurrentFrame.mnMinY, CurrentFrame.mnMaxX, CurrentFrame.mnMaxY,
        lastFramePoints.data(), uvRight.data(), (int)lastFramePoints.size(), mbCheckOrientation ? 1 : 0, occupied.data(),
        match.data());
    if (n < 0) throw std::runtime_error(std::string("SearchByProjection: ") + orbx_last_error());
    return n;
  }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
  int device_;
};

// Frame::ComputeStereoMatches (src/Frame.cc:921-1084) on the device-resident results of the two
// extractors' last operator() calls.  mbf / mb as in Frame; fills mvuRight / mvDepth (size N = left count).
inline void ComputeStereoMatches(ORBextractor& left, ORBextractor& right, int N, float mbf, float mb,
                                 std::vector<float>& mvuRight, std::vector<float>& mvDepth) {
  if (orbx_stereo_match_batch(left.handle(), 0, right.handle(), 0, 1, mbf, mb) != ORBX_OK)
    throw std::runtime_error(std::string("ComputeStereoMatches: ") + orbx_last_error());
  mvuRight.assign(N, -1.0f);
  mvDepth.assign(N, -1.0f);
  if (N && orbx_stereo_download(left.handle(), 0, mvuRight.data(), mvDepth.data(), N) != ORBX_OK)
    throw std::runtime_error(std::string("ComputeStereoMatches: ") + orbx_last_error());
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) + Lowe ratio of Frame::ComputeStereoFishEyeMatches
// (src/Frame.cc:46,1293-1302).  idx2/dist2: nQ x 2; ratio_ok[q] = 1 when (*it)[0].distance < (*it)[1].distance*0.7.
inline void BFKnnMatch2(const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, std::vector<int>& idx2,
                        std::vector<int>& dist2, std::vector<uint8_t>& ratio_ok, int device = 0) {
  idx2.assign((size_t)nQ * 2, -1);
  dist2.assign((size_t)nQ * 2, -1);
  ratio_ok.assign(nQ, 0);
  if (orbx_bf_knn2(device, descQ, nQ, descT, nT, idx2.data(), dist2.data(), ratio_ok.data()) != ORBX_OK)
    throw std::runtime_error(std::string("BFKnnMatch2: ") + orbx_last_error());
}

// Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1273-1331) incl. KannalaBrandt8::TriangulateMatches
// (src/CameraModels/KannalaBrandt8.cpp:341-432).  Arguments are the Frame members the routine reads (mvKeys,
// mDescriptors, monoLeft, mvKeysRight, mDescriptorsRight, monoRight, mvLevelSigma2) and the rig (both cameras'
// mvParameters, mRlr, mtlr); outputs are the members it fills.  mvStereo3Dpoints holds x, y, z per left keypoint
// (Eigen::Vector3f in the reference; zeros where the reference leaves the entry uninitialised).  Returns nMatches.
inline int ComputeStereoFishEyeMatches(const std::vector<ocv::KeyPoint>& mvKeys, const uint8_t* mDescriptors, int monoLeft,
                                       const std::vector<ocv::KeyPoint>& mvKeysRight, const uint8_t* mDescriptorsRight,
                                       int monoRight, const orbx_kb8_rig& rig, const std::vector<float>& mvLevelSigma2,
                                       std::vector<int>& mvLeftToRightMatch, std::vector<int>& mvRightToLeftMatch,
                                       std::vector<float>& mvDepth, std::vector<float>& mvuRight,
                                       std::vector<float>& mvStereo3Dpoints, int device = 0) {
  const int Nleft = (int)mvKeys.size(), Nright = (int)mvKeysRight.size();
  mvLeftToRightMatch.assign(Nleft, -1);
  mvRightToLeftMatch.assign(Nright, -1);
  mvDepth.assign(Nleft, -1.0f);
  mvuRight.assign(Nleft, -1.0f);  // :1288, never written afterwards
  mvStereo3Dpoints.assign((size_t)Nleft * 3, 0.0f);
  const int n = orbx_fisheye_stereo_match(
      device, reinterpret_cast<const orbx_keypoint*>(mvKeys.data()), mDescriptors, Nleft, monoLeft,
      reinterpret_cast<const orbx_keypoint*>(mvKeysRight.data()), mDescriptorsRight, Nright, monoRight, &rig,
      mvLevelSigma2.data(), (int)mvLevelSigma2.size(), mvLeftToRightMatch.data(), mvRightToLeftMatch.data(), mvDepth.data(),
      mvStereo3Dpoints.data(), nullptr);
  if (n < 0) throw std::runtime_error(std::string("ComputeStereoFishEyeMatches: ") + orbx_last_error());
  return n;
}

// Frame::UndistortKeyPoints (src/Frame.cc:853-885): mvKeysUn from mvKeys.  K = fx fy cx cy of Pinhole::toK(),
// mDistCoef = the 4 or 5 OpenCV coefficients (float).
inline void UndistortKeyPoints(const std::vector<ocv::KeyPoint>& mvKeys, const float K[4], const std::vector<float>& mDistCoef,
                               std::vector<ocv::KeyPoint>& mvKeysUn, int device = 0) {
  mvKeysUn.resize(mvKeys.size());
  if (orbx_undistort_keypoints(device, reinterpret_cast<const orbx_keypoint*>(mvKeys.data()), (int)mvKeys.size(), K,
                               mDistCoef.data(), (int)mDistCoef.size(), reinterpret_cast<orbx_keypoint*>(mvKeysUn.data())) !=
      ORBX_OK)
    throw std::runtime_error(std::string("UndistortKeyPoints: ") + orbx_last_error());
}

// Frame::ComputeImageBounds (src/Frame.cc:887-919): fills mnMinX, mnMinY, mnMaxX, mnMaxY.
inline void ComputeImageBounds(int cols, int rows, const float K[4], const std::vector<float>& mDistCoef, float& mnMinX,
                               float& mnMinY, float& mnMaxX, float& mnMaxY, int device = 0) {
  float b[4];
  if (orbx_compute_image_bounds(device, cols, rows, K, mDistCoef.data(), (int)mDistCoef.size(), b) != ORBX_OK)
    throw std::runtime_error(std::string("ComputeImageBounds: ") + orbx_last_error());
  mnMinX = b[0]; mnMinY = b[1]; mnMaxX = b[2]; mnMaxY = b[3];
}

}  // namespace ORB_SLAM3

#endif  // ORBX_SHIM_ORBMATCHER_H
