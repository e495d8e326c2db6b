// Preprocess.h — C++ host mirror of the image pre-processing calls in front of ORB_SLAM3's extractor, over the C ABI of
// include/orbx.h:
//   cv::remap(imLeft, imLeftToFeed, M1l, M2l, cv::INTER_LINEAR)          src/System.cc:294-295
//   cv::createCLAHE(3.0, cv::Size(8, 8)); clahe->apply(im, im)           Examples/Stereo/stereo_tum_vi.cc:100,142-143
// Header-only, no pixel arithmetic here: the HIP kernels of liborbx.so do the work and every call throws when no MI355X is
// visible (no CPU fallback).  With OpenCV headers present the cv:: types are accepted directly.
#ifndef ORBX_SHIM_PREPROCESS_H
#define ORBX_SHIM_PREPROCESS_H

#include "ORBextractor.h"

namespace ORB_SLAM3 {

// cv::remap with CV_32FC1 maps and INTER_LINEAR (the only form the reference uses).  dst is (re)allocated to the map size.
inline void remap(const ocv::Mat& src, ocv::Mat& dst, const float* map1, const float* map2, int mapCols, int mapRows,
                  size_t mapStepFloats = 0, int device = 0) {
#ifdef ORBX_HAVE_OPENCV
  CV_Assert(src.depth() == CV_8U);
  const int cn = src.channels();
  dst.create(mapRows, mapCols, src.type());
#else
  const int cn = 1;
  dst.create(mapRows, mapCols);
#endif
  if (orbx_remap_linear(device, src.data, src.cols, src.rows, (ptrdiff_t)src.step, cn, map1, map2,
                        (ptrdiff_t)(mapStepFloats ? mapStepFloats : (size_t)mapCols), dst.data, mapCols, mapRows,
                        (ptrdiff_t)dst.step) != ORBX_OK)
    throw std::runtime_error(std::string("remap: ") + orbx_last_error());
}
#ifdef ORBX_HAVE_OPENCV
inline void remap(cv::InputArray src, cv::OutputArray dst, cv::InputArray map1, cv::InputArray map2, int interpolation) {
  CV_Assert(interpolation == cv::INTER_LINEAR && map1.type() == CV_32FC1 && map2.type() == CV_32FC1);
  cv::Mat s = src.getMat(), m1 = map1.getMat(), m2 = map2.getMat(), d;
  CV_Assert(m1.size() == m2.size() && m1.step == m2.step);
  remap(s, d, m1.ptr<float>(), m2.ptr<float>(), m1.cols, m1.rows, m1.step / sizeof(float));
  d.copyTo(dst);
}
#endif

// cv::CLAHE as the TUM-VI front ends use it.
class CLAHE {
 public:
  explicit CLAHE(double clipLimit = 40.0, int tilesX = 8, int tilesY = 8, int device = 0)
      : clip_(clipLimit), tx_(tilesX), ty_(tilesY), device_(device) {}
  void apply(const ocv::Mat& src, ocv::Mat& dst) const {
    ocv::Mat out;
#ifdef ORBX_HAVE_OPENCV
    CV_Assert(src.type() == CV_8UC1);
    out.create(src.rows, src.cols, CV_8UC1);
#else
    out.create(src.rows, src.cols);
#endif
    if (orbx_clahe(device_, src.data, src.cols, src.rows, (ptrdiff_t)src.step, clip_, tx_, ty_, out.data, (ptrdiff_t)out.step) !=
        ORBX_OK)
      throw std::runtime_error(std::string("CLAHE::apply: ") + orbx_last_error());
    dst = out;  // src and dst may be the same Mat, as in clahe->apply(imLeft, imLeft)
  }
  void setClipLimit(double c) { clip_ = c; }
  double getClipLimit() const { return clip_; }

 private:
  double clip_;
  int tx_, ty_, device_;
};
inline std::shared_ptr<CLAHE> createCLAHE(double clipLimit = 40.0, int tilesX = 8, int tilesY = 8) {
  return std::make_shared<CLAHE>(clipLimit, tilesX, tilesY);
}

// The rectification of System::TrackStereo for both eyes with the maps resident on the device (uploaded once, as
// Settings::precomputeRectificationMaps computes them once): left frame -> maps (M1l, M2l), right frame -> (M1r, M2r);
// optionally CLAHE first.  handle() can be passed to orbx_extract_batch_raw_device to keep the frames on the device.
class StereoRectifier {
 public:
  StereoRectifier(int srcW, int srcH, int outW, int outH, const float* M1l, const float* M2l, const float* M1r, const float* M2r,
                  double claheClip = 0.0, int claheTiles = 0, int device = 0) {
    std::vector<float> mx((size_t)2 * outW * outH), my(mx.size());
    const size_t per = (size_t)outW * outH;
    std::memcpy(mx.data(), M1l, per * 4);
    std::memcpy(mx.data() + per, M1r, per * 4);
    std::memcpy(my.data(), M2l, per * 4);
    std::memcpy(my.data() + per, M2r, per * 4);
    orbx_preproc_params p{};
    p.src_w = srcW; p.src_h = srcH; p.channels = 1; p.out_w = outW; p.out_h = outH;
    p.map_x = mx.data(); p.map_y = my.data(); p.map_stride = outW; p.n_maps = 2;
    p.clahe_clip_limit = claheClip; p.clahe_tiles_x = p.clahe_tiles_y = claheTiles;
    if (orbx_preproc_create(&p, 2, device, &h_) != ORBX_OK)
      throw std::runtime_error(std::string("StereoRectifier: ") + orbx_last_error());
    w_ = outW; h_out_ = outH;
  }
  ~StereoRectifier() { orbx_preproc_destroy(h_); }
  StereoRectifier(const StereoRectifier&) = delete;
  StereoRectifier& operator=(const StereoRectifier&) = delete;
  void operator()(const ocv::Mat& imLeft, const ocv::Mat& imRight, ocv::Mat& outLeft, ocv::Mat& outRight) {
    run(imLeft, 0, outLeft);
    run(imRight, 1, outRight);
  }
  orbx_preproc* handle() const { return h_; }

 private:
  void run(const ocv::Mat& im, int eye, ocv::Mat& out) {
#ifdef ORBX_HAVE_OPENCV
    out.create(h_out_, w_, CV_8UC1);
#else
    out.create(h_out_, w_);
#endif
    if (orbx_preproc_run(h_, im.data, (ptrdiff_t)im.step, eye, out.data, (ptrdiff_t)out.step) != ORBX_OK)
      throw std::runtime_error(std::string("StereoRectifier: ") + orbx_last_error());
  }
  orbx_preproc* h_ = nullptr;
  int w_ = 0, h_out_ = 0;
};

}  // namespace ORB_SLAM3
#endif
