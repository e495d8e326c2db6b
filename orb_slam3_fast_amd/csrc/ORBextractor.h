// ORBextractor.h — C++ host mirror of ORB_SLAM3::ORBextractor over the C ABI of include/orbx.h.
//
// Same class name, constructor arguments, operator() signature, getters and public mvImagePyramid member as
// the reference (include/ORBextractor.h:49-118 of hellovuong/ORB_SLAM3_FAST), so Frame / Tracking call sites
// (src/Frame.cc:549-560, src/Tracking.cc:628-637) compile unchanged against it.  Header-only; it contains no
// pixel arithmetic — every stage runs in the HIP kernels of liborbx.so, and construction throws when no
// MI355X is visible (no CPU fallback).
//
// With OpenCV headers on the include path the cv:: types are used directly; without them (this repo's own
// tests) a minimal layout-compatible stand-in (orbx::cvlite) keeps the signatures.
#ifndef ORBX_SHIM_ORBEXTRACTOR_H
#define ORBX_SHIM_ORBEXTRACTOR_H

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/orbx.h"

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>) && !defined(ORBX_NO_OPENCV)
#include <opencv2/core.hpp>
#define ORBX_HAVE_OPENCV 1
#endif
#endif

namespace orbx {
namespace cvlite {
// Minimal stand-ins with cv::KeyPoint's memory layout and the handful of cv::Mat members the shim needs.
struct Point2f {
  float x = 0, y = 0;
  Point2f() {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct KeyPoint {
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == sizeof(orbx_keypoint), "cv::KeyPoint layout");
class Mat {
 public:
  int rows = 0, cols = 0;
  size_t step = 0;
  uint8_t* data = nullptr;
  Mat() {}
  Mat(int r, int c, uint8_t* ext, size_t step_) : rows(r), cols(c), step(step_), data(ext) {}
  void create(int r, int c) {
    buf_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * c], std::default_delete<uint8_t[]>());
    rows = r;
    cols = c;
    step = (size_t)c;
    data = buf_.get();
  }
  void release() {
    buf_.reset();
    rows = cols = 0;
    step = 0;
    data = nullptr;
  }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  uint8_t* ptr(int r = 0) { return data + (size_t)r * step; }
  const uint8_t* ptr(int r = 0) const { return data + (size_t)r * step; }

 private:
  std::shared_ptr<uint8_t> buf_;
};
}  // namespace cvlite
}  // namespace orbx

namespace ORB_SLAM3 {

#ifdef ORBX_HAVE_OPENCV
namespace ocv {
using Mat = cv::Mat;
using KeyPoint = cv::KeyPoint;
using Point2f = cv::Point2f;
using InputArray = cv::InputArray;
using OutputArray = cv::OutputArray;
}  // namespace ocv
#else
namespace ocv {
using Mat = orbx::cvlite::Mat;
using KeyPoint = orbx::cvlite::KeyPoint;
using Point2f = orbx::cvlite::Point2f;
using InputArray = const orbx::cvlite::Mat&;
using OutputArray = orbx::cvlite::Mat&;
}  // namespace ocv
#endif
static_assert(sizeof(ocv::KeyPoint) == sizeof(orbx_keypoint), "KeyPoint must be 28 bytes like cv::KeyPoint");

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  // src/ORBextractor.cc:408-469.  max_width/max_height bound the images this instance will see (device
  // buffers are sized once); the reference has no such limit because it reallocates per call.
  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int max_width = 1920,
               int max_height = 1200, int device = 0)
      : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST),
        minThFAST(minThFAST) {
    orbx_params p{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
    int rc = orbx_extractor_create(&p, max_width, max_height, 2, device, &h_);  // 2: room for ExtractStereo
    if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBextractor: ") + orbx_last_error());
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    umax.resize(16);
    orbx_get_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                    mvInvLevelSigma2.data(), mnFeaturesPerLevel.data(), umax.data());
    mvImagePyramid.resize(nlevels);
    // rows of the handle's own result arrays: nfeatures + (4 * 8 + 4) per level -- the reference's first DistributeOctTree pass
    // can leave up to 4 * nIni nodes on a level whose quota is smaller (src/ORBextractor.cc:575-601)
    orbx_batch_results_device(h_, nullptr, nullptr, nullptr, nullptr, &capacity_);
  }
  ~ORBextractor() { orbx_extractor_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // src/ORBextractor.cc:1015-1106.  Returns monoIndex, -1 for an empty image; throws on device errors
  // (the reference can throw cv::Exception from the same call).  The mask is ignored, as in the reference.
  int operator()(ocv::InputArray _image, ocv::InputArray _mask, std::vector<ocv::KeyPoint>& _keypoints,
                 ocv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    (void)_mask;
#ifdef ORBX_HAVE_OPENCV
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    const uint8_t* data = image.data;
    const int w = image.cols, h = image.rows;
    const ptrdiff_t step = (ptrdiff_t)image.step;
#else
    if (_image.empty()) return -1;
    const uint8_t* data = _image.data;
    const int w = _image.cols, h = _image.rows;
    const ptrdiff_t step = (ptrdiff_t)_image.step;
#endif
    int n = 0;
    const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0;
    const int lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
    ApplyHostPyramidMode();
    // (NULL output arrays: the results stay in the handle's page-locked block and are copied ONCE, into the caller's containers)
    const int mono = orbx_extract(h_, data, w, h, step, lap0, lap1, nullptr, nullptr, 0, &n);
    if (mono == ORBX_E_EMPTY) return -1;
    if (mono < 0) throw std::runtime_error(std::string("ORBextractor::operator(): ") + orbx_last_error());
    const orbx_keypoint* rk = nullptr;
    const uint8_t* rd = nullptr;
    if (orbx_host_results(h_, 0, &rk, &rd, nullptr, nullptr, nullptr, nullptr) != ORBX_OK)
      throw std::runtime_error(std::string("ORBextractor::operator(): ") + orbx_last_error());
    FillOutputs(_keypoints, _descriptors, rk, rd, n);
    if (mbKeepHostPyramid) ViewImagePyramid(0);
    return mono;
  }

  // Both eyes of one stereo frame through ONE batched pipeline and one synchronisation (orbx_extract_stereo): replaces
  // the two threaded ExtractORB calls of the stereo Frame constructor (src/Frame.cc:200-203) and, with mbf > 0, the
  // ComputeStereoMatches that follows (:921-1084) -- 0.37 ms instead of 0.62 ms per 1280x720 frame.  The left eye is
  // image 0, the right eye image 1 of this instance afterwards (mvImagePyramid refers to the left eye).
  void ExtractStereo(ocv::InputArray imLeft, ocv::InputArray imRight, std::vector<ocv::KeyPoint>& keysLeft,
                     ocv::OutputArray descLeft, std::vector<ocv::KeyPoint>& keysRight, ocv::OutputArray descRight,
                     const std::vector<int>& lapLeft, const std::vector<int>& lapRight, int& monoLeft, int& monoRight,
                     float mbf = 0.f, float mb = 0.f, std::vector<float>* mvuRight = nullptr,
                     std::vector<float>* mvDepth = nullptr) {
#ifdef ORBX_HAVE_OPENCV
    cv::Mat L = imLeft.getMat(), R = imRight.getMat();
    CV_Assert(L.type() == CV_8UC1 && R.type() == CV_8UC1 && L.size() == R.size());
    const uint8_t *pl = L.data, *pr = R.data;
    const int w = L.cols, h = L.rows;
    const ptrdiff_t sl = (ptrdiff_t)L.step, sr = (ptrdiff_t)R.step;
#else
    const uint8_t *pl = imLeft.data, *pr = imRight.data;
    const int w = imLeft.cols, h = imLeft.rows;
    const ptrdiff_t sl = (ptrdiff_t)imLeft.step, sr = (ptrdiff_t)imRight.step;
    if (imRight.cols != w || imRight.rows != h) throw std::invalid_argument("ExtractStereo: image sizes differ");
#endif
    const int32_t ll[2] = {lapLeft.size() > 0 ? lapLeft[0] : 0, lapLeft.size() > 1 ? lapLeft[1] : 0};
    const int32_t lr[2] = {lapRight.size() > 0 ? lapRight[0] : 0, lapRight.size() > 1 ? lapRight[1] : 0};
    int nl = 0, nr = 0;
    const bool stereo = mbf > 0.f && mvuRight && mvDepth;
    ApplyHostPyramidMode();
    // The outputs are handed to the library as ITS output arrays, sized to the extractor's capacity and trimmed afterwards: the
    // library copies keypoints and descriptors into them while the stereo association still runs (include/orbx.h), so the copies
    // cost nothing at the end of the call.  (Descriptor outputs that are not plain matrices take the copy-after path below.)
    const int cap = capacity_;   // (nfeatures + 3 * nlevels is too small for tiny quotas)
    uint8_t* dl = DescBuffer(descLeft, cap);
    uint8_t* dr = dl ? DescBuffer(descRight, cap) : nullptr;
    if (dl && dr) {
      keysLeft.resize((size_t)cap);
      keysRight.resize((size_t)cap);
      if (stereo) {
        mvuRight->resize((size_t)cap);
        mvDepth->resize((size_t)cap);
      }
      const int rc = orbx_extract_stereo(h_, pl, pr, w, h, sl, sr, ll, lr, reinterpret_cast<orbx_keypoint*>(keysLeft.data()), dl, cap, &nl,
                                         &monoLeft, reinterpret_cast<orbx_keypoint*>(keysRight.data()), dr, cap, &nr, &monoRight,
                                         stereo ? mbf : 0.f, mb, stereo ? mvuRight->data() : nullptr, stereo ? mvDepth->data() : nullptr);
      if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBextractor::ExtractStereo: ") + orbx_last_error());
      keysLeft.resize((size_t)nl);
      keysRight.resize((size_t)nr);
      TrimDesc(descLeft, nl);
      TrimDesc(descRight, nr);
      if (stereo) {
        mvuRight->resize((size_t)nl);
        mvDepth->resize((size_t)nl);
      }
    } else {
      if (orbx_extract_stereo(h_, pl, pr, w, h, sl, sr, ll, lr, nullptr, nullptr, 0, &nl, &monoLeft, nullptr, nullptr, 0, &nr,
                              &monoRight, stereo ? mbf : 0.f, mb, nullptr, nullptr) != ORBX_OK)
        throw std::runtime_error(std::string("ORBextractor::ExtractStereo: ") + orbx_last_error());
      const orbx_keypoint *kl = nullptr, *kr = nullptr;
      const uint8_t *dl2 = nullptr, *dr2 = nullptr;
      const float *ur = nullptr, *dp = nullptr;
      if (orbx_host_results(h_, 0, &kl, &dl2, nullptr, nullptr, &ur, &dp) != ORBX_OK ||
          orbx_host_results(h_, 1, &kr, &dr2, nullptr, nullptr, nullptr, nullptr) != ORBX_OK)
        throw std::runtime_error(std::string("ORBextractor::ExtractStereo: ") + orbx_last_error());
      FillOutputs(keysLeft, descLeft, kl, dl2, nl);
      FillOutputs(keysRight, descRight, kr, dr2, nr);
      if (stereo) {
        mvuRight->assign(ur, ur + nl);
        mvDepth->assign(dp, dp + nl);
      }
    }
    if (mbKeepHostPyramid) {
      ViewImagePyramid(0);
      ViewImagePyramid(1);
    }
  }

  // Not in the reference (its OpenCV is chosen at link time): which OpenCV's GaussianBlur taps the descriptors follow --
  // 440 = OpenCV 4.0 .. 4.5.0 (README.md:101 "tested with 4.4.0"; 44016 / 44032 = with the flooring 16- / 32-lane vector body of
  // its vertical pass), 451 = OpenCV >= 4.5.1 (default).  include/orbx.h.
  void SetOpenCVCompat(int opencv_version) {
    if (orbx_set_opencv_compat(h_, opencv_version) != ORBX_OK)
      throw std::invalid_argument(std::string("ORBextractor::SetOpenCVCompat: ") + orbx_last_error());
  }

  int inline GetLevels() { return nlevels; }
  float inline GetScaleFactor() { return (float)scaleFactor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  // Public in the reference (include/ORBextractor.h:86) and read by an unmodified Frame::ComputeStereoMatches
  // (src/Frame.cc:927,1011,1024,1029).  The pyramid lives in HBM; with mbKeepHostPyramid (the DEFAULT, so that
  // unmodified readers of mvImagePyramid keep working) every operator() / ExtractStereo keeps a host copy current:
  // orbx_set_host_pyramid -- the library copies the levels (~2.9 MB per 1280x720 eye) into page-locked memory with the DMA
  // engines BESIDE the frame's kernels, and mvImagePyramid[l] is a cv::Mat HEADER over that memory (no host-side copy;
  // valid until the next call on this extractor, which is the reference's own lifetime: ComputePyramid overwrites it,
  // src/ORBextractor.cc:1108-1145 -- clone() a level to keep it longer).  A caller that has replaced ComputeStereoMatches
  // by the device version of this repo (ORBmatcher.h) may set mbKeepHostPyramid = false and call SyncImagePyramid() only
  // when it needs pixels.  After ExtractStereo, mvImagePyramid is the LEFT eye (image 0) and mvImagePyramidRight the
  // right eye (image 1).
  std::vector<ocv::Mat> mvImagePyramid;
  std::vector<ocv::Mat> mvImagePyramidRight;
  bool mbKeepHostPyramid = true;
  void ViewImagePyramid(int image = 0) {
    std::vector<ocv::Mat>& dst = image == 0 ? mvImagePyramid : mvImagePyramidRight;
    dst.resize(nlevels);
    for (int l = 0; l < nlevels; l++) {
      const uint8_t* p = nullptr;
      int w = 0, h = 0;
      ptrdiff_t st = 0;
      if (orbx_host_pyramid_level(h_, image, l, &p, &w, &h, &st) != ORBX_OK)
        throw std::runtime_error(std::string("mvImagePyramid: ") + orbx_last_error());
#ifdef ORBX_HAVE_OPENCV
      dst[l] = cv::Mat(h, w, CV_8UC1, const_cast<uint8_t*>(p), (size_t)st);
#else
      dst[l] = ocv::Mat(h, w, const_cast<uint8_t*>(p), (size_t)st);
#endif
    }
  }
  void SyncImagePyramid(int image = 0) {
    std::vector<ocv::Mat>& dst = image == 0 ? mvImagePyramid : mvImagePyramidRight;
    dst.resize(nlevels);
    std::vector<uint8_t*> ptrs(nlevels);
    std::vector<ptrdiff_t> strides(nlevels);
    for (int l = 0; l < nlevels; l++) {
      int w = 0, h = 0;
      if (orbx_pyramid_level(h_, image, l, 0, nullptr, 0, &w, &h) != ORBX_OK)   // size query only: no copy, no sync
        throw std::runtime_error(std::string("mvImagePyramid: ") + orbx_last_error());
      dst[l].release();  // (a header over the library's page-locked copy, ViewImagePyramid, must not be written through)
#ifdef ORBX_HAVE_OPENCV
      dst[l].create(h, w, CV_8UC1);
#else
      dst[l].create(h, w);
#endif
      ptrs[l] = dst[l].ptr(0);
      strides[l] = (ptrdiff_t)dst[l].step;
    }
    // all levels with asynchronous copies and ONE synchronisation (was: 2 x nlevels blocking calls)
    if (orbx_pyramid_download(h_, image, nlevels, ptrs.data(), strides.data()) != ORBX_OK)
      throw std::runtime_error(std::string("mvImagePyramid: ") + orbx_last_error());
  }

  orbx_extractor* handle() { return h_; }

 protected:
  int nfeatures;
  double scaleFactor;
  int nlevels;
  int iniThFAST;
  int minThFAST;
  std::vector<int> mnFeaturesPerLevel;
  std::vector<int> umax;
  std::vector<float> mvScaleFactor;
  std::vector<float> mvInvScaleFactor;
  std::vector<float> mvLevelSigma2;
  std::vector<float> mvInvLevelSigma2;

 private:
  // a cap x 32 byte descriptor matrix behind `desc` for the library to fill (nullptr: `desc` is not a plain matrix), and its trim to n rows
  static uint8_t* DescBuffer(ocv::OutputArray desc, int cap) {
#ifdef ORBX_HAVE_OPENCV
    if (!desc.isMat()) return nullptr;
    desc.create(cap, 32, CV_8U);
    cv::Mat& m = desc.getMatRef();
    return m.isContinuous() ? m.data : nullptr;
#else
    desc.create(cap, 32);
    return desc.data;
#endif
  }
  static void TrimDesc(ocv::OutputArray desc, int n) {
    if (n == 0) {
      desc.release();
      return;
    }
#ifdef ORBX_HAVE_OPENCV
    cv::Mat& m = desc.getMatRef();
    m = m.rowRange(0, n);   // (a header over the first n rows: no copy)
#else
    desc.rows = n;
#endif
  }
  static void FillOutputs(std::vector<ocv::KeyPoint>& keys, ocv::OutputArray desc, const orbx_keypoint* k, const uint8_t* d, int n) {
    keys.resize(n);
    if (n) std::memcpy(static_cast<void*>(keys.data()), k, (size_t)n * sizeof(orbx_keypoint));
    if (n == 0) {
      desc.release();
      return;
    }
#ifdef ORBX_HAVE_OPENCV
    desc.create(n, 32, CV_8U);
    cv::Mat m = desc.getMat();
    for (int i = 0; i < n; i++) std::memcpy(m.ptr(i), d + (size_t)i * 32, 32);
#else
    desc.create(n, 32);
    std::memcpy(desc.data, d, (size_t)n * 32);
#endif
  }
  void ApplyHostPyramidMode() {  // (mbKeepHostPyramid is a public flag: follow it lazily)
    if ((int)mbKeepHostPyramid != keepSet_) {
      orbx_set_host_pyramid(h_, mbKeepHostPyramid ? 1 : 0);
      keepSet_ = (int)mbKeepHostPyramid;
    }
  }
  orbx_extractor* h_ = nullptr;
  int keepSet_ = -1;
  int capacity_ = 0;
};

}  // namespace ORB_SLAM3

#endif  // ORBX_SHIM_ORBEXTRACTOR_H
