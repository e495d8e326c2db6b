// TEST-ONLY stand-in for <opencv2/core.hpp> (this image has no OpenCV; profiles/r3_opencv_probe_*.txt).
//
// Purpose: let a compiler see the `#ifdef ORBX_HAVE_OPENCV` branches of the C++ mirror headers
// (orb_slam3_fast_amd/csrc/ORBextractor.h, ORBmatcher.h, Preprocess.h, ORBVocabulary.h) -- i.e. the reference's own
// signature  int ORBextractor::operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&, cv::OutputArray,
// std::vector<int>&)  (include/ORBextractor.h:64-68 of the reference) -- and run them through tests/cpp/frame_like.cpp.
// It models only the members those branches touch, with OpenCV's documented semantics (reference-counted Mat header,
// _InputArray / _OutputArray proxies, 28-byte KeyPoint).  No arithmetic of OpenCV is restated here; it is NOT used to
// build the reference and never ships with the product.
#ifndef ORBX_TEST_OPENCV_STUB_CORE_HPP
#define ORBX_TEST_OPENCV_STUB_CORE_HPP
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#define CV_CN_SHIFT 3
#define CV_MAT_DEPTH_MASK 7
#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)

namespace cv {
class Exception : public std::runtime_error {
 public:
  explicit Exception(const std::string& m) : std::runtime_error(m) {}
};
#define CV_Assert(expr)                                                                     \
  do {                                                                                      \
    if (!(expr)) throw cv::Exception(std::string("CV_Assert failed: ") + #expr);            \
  } while (0)

enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1 };

struct Size {
  int width = 0, height = 0;
  Size() {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
};
struct Point2f {
  float x = 0, y = 0;
  Point2f() {}
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
class KeyPoint {
 public:
  Point2f pt;
  float size = 0, angle = -1, response = 0;
  int octave = 0, class_id = -1;
  KeyPoint() {}
  KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
      : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint is 28 bytes");

class _OutputArray;
class Mat {
 public:
  static const size_t AUTO_STEP = 0;
  int flags = 0, rows = 0, cols = 0;
  uint8_t* data = nullptr;
  struct MatStep {
    size_t v = 0;
    operator size_t() const { return v; }
    MatStep& operator=(size_t s) { v = s; return *this; }
  } step;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void* ext, size_t step_ = AUTO_STEP)
      : flags(type), rows(r), cols(c), data(static_cast<uint8_t*>(ext)) {
    step = step_ ? step_ : (size_t)c * elemSize();
  }
  int type() const { return flags & 0xFFF; }
  int depth() const { return flags & CV_MAT_DEPTH_MASK; }
  int channels() const { return ((flags & 0xFFF) >> CV_CN_SHIFT) + 1; }
  size_t elemSize() const { return (size_t)channels() * (depth() == CV_32F ? 4 : 1); }
  Size size() const { return Size(cols, rows); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  bool isContinuous() const { return rows <= 1 || (size_t)step == (size_t)cols * elemSize(); }
  void create(int r, int c, int type) {
    if (data && r == rows && c == cols && type == this->type() && buf_) return;
    flags = type;
    rows = r;
    cols = c;
    step = (size_t)c * elemSize();
    buf_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)step * (size_t)(r > 0 ? r : 1)], std::default_delete<uint8_t[]>());
    data = buf_.get();
  }
  void release() {
    buf_.reset();
    data = nullptr;
    rows = cols = 0;
    step = 0;
  }
  uint8_t* ptr(int r = 0) { return data + (size_t)r * (size_t)step; }
  const uint8_t* ptr(int r = 0) const { return data + (size_t)r * (size_t)step; }
  template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(ptr(r)); }
  template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(ptr(r)); }
  Mat rowRange(int r0, int r1) const {   // header over rows [r0, r1) of the same pixels
    Mat m = *this;
    m.data = data + (size_t)r0 * (size_t)step;
    m.rows = r1 - r0;
    return m;
  }
  Mat clone() const {
    Mat m;
    m.create(rows, cols, type());
    for (int r = 0; r < rows; r++) std::memcpy(m.ptr(r), ptr(r), (size_t)cols * elemSize());
    return m;
  }
  inline void copyTo(const _OutputArray& dst) const;

 private:
  std::shared_ptr<uint8_t> buf_;  // shared header semantics: copies of a Mat refer to the same pixels
};

class _InputArray {
 public:
  _InputArray() {}
  _InputArray(const Mat& m) : m_(&m) {}
  Mat getMat() const { return m_ ? *m_ : Mat(); }
  bool isMat() const { return m_ != nullptr; }
  bool empty() const { return !m_ || m_->empty(); }
  int type() const { return m_ ? m_->type() : 0; }

 protected:
  const Mat* m_ = nullptr;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray() {}
  _OutputArray(Mat& m) : _InputArray(m), w_(&m) {}
  void create(int rows, int cols, int type) const {
    if (!w_) throw Exception("create() on noArray()");
    w_->create(rows, cols, type);
  }
  void release() const {
    if (w_) w_->release();
  }
  Mat getMat() const { return w_ ? *w_ : Mat(); }  // shares the pixels with the caller's Mat
  Mat& getMatRef() const {
    if (!w_) throw Exception("getMatRef() on noArray()");
    return *w_;
  }
 private:
  Mat* w_ = nullptr;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline InputArray noArray() {
  static _InputArray none;
  return none;
}
inline void Mat::copyTo(const _OutputArray& dst) const {
  dst.create(rows, cols, type());
  Mat d = dst.getMat();
  for (int r = 0; r < rows; r++) std::memcpy(d.ptr(r), ptr(r), (size_t)cols * elemSize());
}
}  // namespace cv
#endif
