// ORBVocabulary.h — C++ host mirror of ORB_SLAM3::ORBVocabulary (include/ORBVocabulary.h: DBoW2::TemplatedVocabulary<
// FORB::TDescriptor, FORB>) for the calls on the per-frame path, over the C ABI of include/orbx.h:
//   mpVocabulary->loadFromTextFile(strVocFile)                         src/System.cc:131
//   mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)     src/Frame.cc:846-851, src/KeyFrame.cc:100-107
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)     src/ORBmatcher.cc:230-404  (free function below)
// The tree lives on the device, the descents / map assembly / matching run in the HIP kernels of liborbx.so; the std::map
// types of DBoW2 (BowVector.h:58, FeatureVector.h:23) are filled from the sorted arrays the library returns.
#ifndef ORBX_SHIM_ORBVOCABULARY_H
#define ORBX_SHIM_ORBVOCABULARY_H

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

}  // namespace ORB_SLAM3
#endif
