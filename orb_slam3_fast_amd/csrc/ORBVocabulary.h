// ORBVocabulary.h — C++ host mirror of ORB_SLAM3::ORBVocabulary (include/ORBVocabulary.h: DBoW2::TemplatedVocabulary<
// FORB::TDescriptor, FORB>) for the calls on the per-frame path, over the C ABI of include/orbx.h:
//   mpVocabulary->loadFromTextFile(strVocFile)                         src/System.cc:131
//   mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)     src/Frame.cc:846-851, src/KeyFrame.cc:100-107
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)     src/ORBmatcher.cc:230-404  (free function below)
// The tree lives on the device, the descents / map assembly / matching run in the HIP kernels of liborbx.so; the std::map
// types of DBoW2 (BowVector.h:58, FeatureVector.h:23) are filled from the sorted arrays the library returns.
#ifndef ORBX_SHIM_ORBVOCABULARY_H
#define ORBX_SHIM_ORBVOCABULARY_H

#include <algorithm>
#include <map>

#include "ORBextractor.h"

namespace DBoW2 {
#ifndef __D_T_BOW_VECTOR__  // DBoW2's own BowVector.h / FeatureVector.h take precedence when they were included first
typedef unsigned int WordId;
typedef double WordValue;
typedef unsigned int NodeId;
class BowVector : public std::map<WordId, WordValue> {};
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
#endif
}  // namespace DBoW2

namespace ORB_SLAM3 {

class ORBVocabulary {
 public:
  explicit ORBVocabulary(int device = 0) : device_(device) {}
  ~ORBVocabulary() { orbx_vocabulary_destroy(h_); }
  ORBVocabulary(const ORBVocabulary&) = delete;
  ORBVocabulary& operator=(const ORBVocabulary&) = delete;

  // TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1421): false when the file cannot be read or parsed.
  bool loadFromTextFile(const std::string& filename) {
    orbx_vocabulary_destroy(h_);
    h_ = nullptr;
    return orbx_vocabulary_load_text(device_, filename.c_str(), &h_) == ORBX_OK;
  }
  bool empty() const { return h_ == nullptr; }
  unsigned int size() const {  // number of words
    int32_t info[6] = {0, 0, 0, 0, 0, 0};
    if (h_) orbx_vocabulary_info(h_, info);
    return (unsigned int)info[3];
  }

  // transform(features, v, fv, levelsup) for the rows of an N x 32 descriptor matrix (Frame::ComputeBoW converts mDescriptors
  // into a vector of rows first; the matrix is taken directly here).
  void transform(const uint8_t* descriptors, int n, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
    v.clear();
    fv.clear();
    if (!h_ || n <= 0) return;
    std::vector<uint32_t> words(n), nodes(n), feats(n);
    std::vector<double> values(n);
    std::vector<int32_t> start((size_t)n + 1);
    int nw = 0, nn = 0;
    if (orbx_bow_transform(h_, descriptors, n, levelsup, words.data(), values.data(), &nw, nodes.data(), start.data(),
                           feats.data(), &nn) < 0)
      throw std::runtime_error(std::string("ORBVocabulary::transform: ") + orbx_last_error());
    for (int i = 0; i < nw; i++) v.insert(v.end(), std::make_pair(words[i], values[i]));
    for (int j = 0; j < nn; j++)
      fv.insert(fv.end(), std::make_pair(nodes[j], std::vector<unsigned int>(feats.begin() + start[j], feats.begin() + start[j + 1])));
  }
  void transform(const ocv::Mat& mDescriptors, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup) const {
    transform(mDescriptors.data, mDescriptors.rows, v, fv, levelsup);
  }
  orbx_vocabulary* handle() const { return h_; }

 private:
  orbx_vocabulary* h_ = nullptr;
  int device_;
};

// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:230-404) on plain
// views of the members it reads.  kfHasGoodMapPoint[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad(); vnMatches[iF] = the
// keyframe feature whose map point the frame's feature iF receives (-1: none) -- the caller maps it to the MapPoint*.
inline int SearchByBoW(const DBoW2::FeatureVector& vFeatVecKF, const std::vector<ocv::KeyPoint>& kfKeys, const uint8_t* kfDescriptors,
                       const std::vector<uint8_t>& kfHasGoodMapPoint, const DBoW2::FeatureVector& vFeatVecF,
                       const std::vector<ocv::KeyPoint>& fKeys, const uint8_t* fDescriptors, int FNleft, float mfNNratio,
                       bool mbCheckOrientation, std::vector<int>& vnMatches, int device = 0) {
  auto flatten = [](const DBoW2::FeatureVector& fv, std::vector<uint32_t>& nodes, std::vector<int32_t>& start,
                    std::vector<uint32_t>& feats) {
    start.assign(1, 0);
    for (const auto& e : fv) {
      nodes.push_back(e.first);
      feats.insert(feats.end(), e.second.begin(), e.second.end());
      start.push_back((int32_t)feats.size());
    }
  };
  std::vector<uint32_t> kn, kf, fn, ff;
  std::vector<int32_t> ks, fs;
  flatten(vFeatVecKF, kn, ks, kf);
  flatten(vFeatVecF, fn, fs, ff);
  vnMatches.assign(fKeys.size(), -1);
  const int n = orbx_search_by_bow(device, kn.data(), ks.data(), kf.data(), (int)kn.size(),
                                   reinterpret_cast<const orbx_keypoint*>(kfKeys.data()), kfDescriptors, kfHasGoodMapPoint.data(),
                                   (int)kfKeys.size(), fn.data(), fs.data(), ff.data(), (int)fn.size(),
                                   reinterpret_cast<const orbx_keypoint*>(fKeys.data()), fDescriptors, (int)fKeys.size(), FNleft,
                                   mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches.data());
  if (n < 0) throw std::runtime_error(std::string("SearchByBoW: ") + orbx_last_error());
  return n;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:766-884; LoopClosing)
// on plain views: hasGoodMapPoint[i] = vpMapPoints[i] && !isBad(), one per feature (two-camera rigs: NLeft + NRight >= mvKeysUn.size());
// vnMatches12[idx1] = the feature of pKF2 whose map point vpMatches12[idx1] receives (-1: none).
inline int SearchByBoW(const DBoW2::FeatureVector& vFeatVec1, const std::vector<ocv::KeyPoint>& vKeysUn1, const uint8_t* Descriptors1,
                       const std::vector<uint8_t>& hasGoodMapPoint1, const DBoW2::FeatureVector& vFeatVec2,
                       const std::vector<ocv::KeyPoint>& vKeysUn2, const uint8_t* Descriptors2,
                       const std::vector<uint8_t>& hasGoodMapPoint2, float mfNNratio, bool mbCheckOrientation,
                       std::vector<int>& vnMatches12, int device = 0) {
  auto flatten = [](const DBoW2::FeatureVector& fv, std::vector<uint32_t>& nodes, std::vector<int32_t>& start,
                    std::vector<uint32_t>& feats) {
    start.assign(1, 0);
    for (const auto& e : fv) {
      nodes.push_back(e.first);
      feats.insert(feats.end(), e.second.begin(), e.second.end());
      start.push_back((int32_t)feats.size());
    }
  };
  std::vector<uint32_t> n1, f1, n2, f2;
  std::vector<int32_t> s1, s2;
  flatten(vFeatVec1, n1, s1, f1);
  flatten(vFeatVec2, n2, s2, f2);
  // One flag per FEATURE (= GetMapPointMatches().size() = descriptor rows).  A two-camera key frame has more features (NLeft +
  // NRight, all in mFeatVec) than mvKeysUn entries: the reference skips idx >= mvKeysUn.size() (:799, :816), so those features
  // are flagged invalid here and the keypoint array is padded to the feature count (only matched features' angles are read).
  if (hasGoodMapPoint1.size() < vKeysUn1.size() || hasGoodMapPoint2.size() < vKeysUn2.size())
    throw std::invalid_argument("SearchByBoW: one map-point flag per feature (at least one per keypoint)");
  const size_t N1 = hasGoodMapPoint1.size(), N2 = hasGoodMapPoint2.size();
  std::vector<uint8_t> v1(hasGoodMapPoint1), v2(hasGoodMapPoint2);
  std::fill(v1.begin() + (std::ptrdiff_t)vKeysUn1.size(), v1.end(), (uint8_t)0);
  std::fill(v2.begin() + (std::ptrdiff_t)vKeysUn2.size(), v2.end(), (uint8_t)0);
  std::vector<ocv::KeyPoint> k1pad, k2pad;
  const ocv::KeyPoint* k1 = vKeysUn1.data();
  const ocv::KeyPoint* k2 = vKeysUn2.data();
  if (N1 > vKeysUn1.size()) { k1pad = vKeysUn1; k1pad.resize(N1); k1 = k1pad.data(); }
  if (N2 > vKeysUn2.size()) { k2pad = vKeysUn2; k2pad.resize(N2); k2 = k2pad.data(); }
  vnMatches12.assign(N1, -1);
  const int n = orbx_search_by_bow_keyframes(
      device, n1.data(), s1.data(), f1.data(), (int)n1.size(), reinterpret_cast<const orbx_keypoint*>(k1), Descriptors1, v1.data(),
      (int)N1, n2.data(), s2.data(), f2.data(), (int)n2.size(), reinterpret_cast<const orbx_keypoint*>(k2), Descriptors2, v2.data(),
      (int)N2, mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data());
  if (n < 0) throw std::runtime_error(std::string("SearchByBoW: ") + orbx_last_error());
  return n;
}

// The members of a KeyFrame that ORBmatcher::SearchForTriangulation reads (single-camera key frames: mpCamera2 == NULL,
// NLeft == -1): mFeatVec, mvKeysUn, mDescriptors (N x 32, continuous), hasMapPoint[i] = (GetMapPoint(i) != NULL), mvuRight (empty:
// no stereo observations), mvScaleFactors, mvLevelSigma2.
struct KeyFrameView {
  const DBoW2::FeatureVector* mFeatVec = nullptr;
  const std::vector<ocv::KeyPoint>* mvKeysUn = nullptr;
  const uint8_t* mDescriptors = nullptr;
  const std::vector<uint8_t>* hasMapPoint = nullptr;
  const std::vector<float>* mvuRight = nullptr;
  const std::vector<float>* mvScaleFactors = nullptr;
  const std::vector<float>* mvLevelSigma2 = nullptr;
};

// ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vector<pair<size_t, size_t>>& vMatchedPairs, bOnlyStereo,
// bCoarse) (src/ORBmatcher.cc:886-1106; LocalMapping::CreateNewMapPoints).  ep = pKF2->mpCamera->project(T2w * Cw1) (:897-901);
// F12 = K1^-T [t12]x R12 K2^-1, the matrix Pinhole::epipolarConstrain forms per candidate (src/CameraModels/Pinhole.cpp:130-133),
// row-major -- with Eigen: `Eigen::Matrix<float, 3, 3, Eigen::RowMajor> F = K1.transpose().inverse() * Sophus::SO3f::hat(t12) * R12
// * K2.inverse();` and F.data().  Returns nmatches; vMatchedPairs in ascending idx1 like :1095-1103.
inline int SearchForTriangulation(const KeyFrameView& KF1, const KeyFrameView& KF2, const float ep[2], const float F12[9],
                                  std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse,
                                  bool mbCheckOrientation = true, int device = 0) {
  auto flatten = [](const DBoW2::FeatureVector& fv, std::vector<uint32_t>& nodes, std::vector<int32_t>& start,
                    std::vector<uint32_t>& feats) {
    start.assign(1, 0);
    for (const auto& e : fv) {
      nodes.push_back(e.first);
      feats.insert(feats.end(), e.second.begin(), e.second.end());
      start.push_back((int32_t)feats.size());
    }
  };
  std::vector<uint32_t> n1, f1, n2, f2;
  std::vector<int32_t> s1, s2;
  flatten(*KF1.mFeatVec, n1, s1, f1);
  flatten(*KF2.mFeatVec, n2, s2, f2);
  const int N1 = (int)KF1.mvKeysUn->size(), N2 = (int)KF2.mvKeysUn->size();
  if ((int)KF1.hasMapPoint->size() != N1 || (int)KF2.hasMapPoint->size() != N2)
    throw std::invalid_argument("SearchForTriangulation: one hasMapPoint flag per keypoint");
  if (KF2.mvScaleFactors->size() != KF2.mvLevelSigma2->size()) throw std::invalid_argument("SearchForTriangulation: level tables differ");
  const float* u1 = KF1.mvuRight && (int)KF1.mvuRight->size() == N1 && N1 ? KF1.mvuRight->data() : nullptr;
  const float* u2 = KF2.mvuRight && (int)KF2.mvuRight->size() == N2 && N2 ? KF2.mvuRight->data() : nullptr;
  std::vector<int> vMatches12(N1, -1);
  const int n = orbx_search_for_triangulation(
      device, n1.data(), s1.data(), f1.data(), (int)n1.size(), reinterpret_cast<const orbx_keypoint*>(KF1.mvKeysUn->data()),
      KF1.mDescriptors, KF1.hasMapPoint->data(), u1, N1, n2.data(), s2.data(), f2.data(), (int)n2.size(),
      reinterpret_cast<const orbx_keypoint*>(KF2.mvKeysUn->data()), KF2.mDescriptors, KF2.hasMapPoint->data(), u2, N2,
      KF2.mvScaleFactors->data(), KF2.mvLevelSigma2->data(), (int)KF2.mvScaleFactors->size(), ep, F12, bOnlyStereo ? 1 : 0,
      bCoarse ? 1 : 0, mbCheckOrientation ? 1 : 0, vMatches12.data());
  if (n < 0) throw std::runtime_error(std::string("SearchForTriangulation: ") + orbx_last_error());
  vMatchedPairs.clear();
  vMatchedPairs.reserve(n);
  for (size_t i = 0; i < vMatches12.size(); i++)
    if (vMatches12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
  return n;
}

// The same call for two-camera (stereo-fisheye) key frames, src/ORBmatcher.cc:906-923,1007-1064: the views hold mvKeys | mvKeysRight
// (mvKeysUn points at the concatenation, NLeft = KeyFrame::NLeft), rig = both key frames' KannalaBrandt8 parameters and the four
// relative poses Tll, Tlr, Trl, Trr; mvLevelSigma2 of BOTH key frames is read (sigmaLevel / unc of TriangulateMatches).
inline int SearchForTriangulation(const KeyFrameView& KF1, int NLeft1, const KeyFrameView& KF2, int NLeft2, const orbx_tri_rig& rig,
                                  std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo, const bool bCoarse,
                                  bool mbCheckOrientation = true, int device = 0) {
  auto flatten = [](const DBoW2::FeatureVector& fv, std::vector<uint32_t>& nodes, std::vector<int32_t>& start,
                    std::vector<uint32_t>& feats) {
    start.assign(1, 0);
    for (const auto& e : fv) {
      nodes.push_back(e.first);
      feats.insert(feats.end(), e.second.begin(), e.second.end());
      start.push_back((int32_t)feats.size());
    }
  };
  std::vector<uint32_t> n1, f1, n2, f2;
  std::vector<int32_t> s1, s2;
  flatten(*KF1.mFeatVec, n1, s1, f1);
  flatten(*KF2.mFeatVec, n2, s2, f2);
  const int N1 = (int)KF1.mvKeysUn->size(), N2 = (int)KF2.mvKeysUn->size();
  if ((int)KF1.hasMapPoint->size() != N1 || (int)KF2.hasMapPoint->size() != N2)
    throw std::invalid_argument("SearchForTriangulation: one hasMapPoint flag per keypoint");
  if (KF1.mvLevelSigma2->size() != KF2.mvLevelSigma2->size()) throw std::invalid_argument("SearchForTriangulation: level tables differ");
  std::vector<int> vMatches12(N1, -1);
  const int n = orbx_search_for_triangulation_rig(
      device, n1.data(), s1.data(), f1.data(), (int)n1.size(), reinterpret_cast<const orbx_keypoint*>(KF1.mvKeysUn->data()),
      KF1.mDescriptors, KF1.hasMapPoint->data(), NLeft1, N1, n2.data(), s2.data(), f2.data(), (int)n2.size(),
      reinterpret_cast<const orbx_keypoint*>(KF2.mvKeysUn->data()), KF2.mDescriptors, KF2.hasMapPoint->data(), NLeft2, N2,
      KF1.mvLevelSigma2->data(), KF2.mvLevelSigma2->data(), (int)KF1.mvLevelSigma2->size(), &rig, bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0,
      mbCheckOrientation ? 1 : 0, vMatches12.data());
  if (n < 0) throw std::runtime_error(std::string("SearchForTriangulation: ") + orbx_last_error());
  vMatchedPairs.clear();
  vMatchedPairs.reserve(n);
  for (size_t i = 0; i < vMatches12.size(); i++)
    if (vMatches12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)vMatches12[i]));
  return n;
}

}  // namespace ORB_SLAM3
#endif
