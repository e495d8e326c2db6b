// C-ABI face of the CPU oracle for ctypes (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
// TEST INFRASTRUCTURE ONLY — see orb_oracle.h.
#include <algorithm>
#include <cstring>

#include "orb_oracle.h"

#include <math.h>

#include <chrono>
#include <thread>

using namespace orbo;

extern "C" {

void* oro_create(int nfeatures, float scaleFactor, int nlevels, int iniTh, int minTh) {
  return new Extractor(nfeatures, scaleFactor, nlevels, iniTh, minTh);
}
void oro_destroy(void* h) { delete (Extractor*)h; }
// variant: 451 (OpenCV >= 4.5.1), 440 (4.0 .. 4.5.0, scalar model), 44016 / 44032 (4.0 .. 4.5.0 with the 16- / 32-lane vector body)
static const int* blur_variant_taps(int variant) { return variant == 440 || variant == 44016 || variant == 44032 ? kBlurTaps440 : kBlurTaps451; }
static int blur_variant_vec(int variant) { return variant == 44016 ? 16 : variant == 44032 ? 32 : 0; }
void oro_set_blur_taps(void* h, int variant) {
  ((Extractor*)h)->blur_taps = blur_variant_taps(variant);
  ((Extractor*)h)->blur_simd_vec = blur_variant_vec(variant);
}
void oro_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* nfeat,
                int* umax) {
  Extractor* e = (Extractor*)h;
  for (int i = 0; i < e->nlevels; i++) {
    scale[i] = e->t.scale[i];
    inv_scale[i] = e->t.inv_scale[i];
    sigma2[i] = e->t.sigma2[i];
    inv_sigma2[i] = e->t.inv_sigma2[i];
    nfeat[i] = e->t.nfeat_level[i];
  }
  for (int i = 0; i < 16; i++) umax[i] = e->t.umax[i];
}
const int8_t* oro_pattern() { return kPattern; }

int oro_extract(void* h, const uint8_t* img, int w, int h_, long stride, int lap0, int lap1, KeyPoint* kps,
                uint8_t* desc, int cap, int* n_out) {
  Extractor* e = (Extractor*)h;
  std::vector<KeyPoint> k;
  std::vector<uint8_t> d;
  int mono = e->extract(img, w, h_, stride, lap0, lap1, k, d);
  if (mono < 0) { *n_out = 0; return mono; }
  *n_out = (int)k.size();
  if ((int)k.size() > cap) return -3;
  std::memcpy(kps, k.data(), k.size() * sizeof(KeyPoint));
  std::memcpy(desc, d.data(), d.size());
  return mono;
}

// One stereo frame with the reference's thread structure and timer placement (`cpu_mt`): both eyes extracted
// concurrently, one std::thread per eye (src/Frame.cc:200-203) with per-level tasks inside (Extractor::extract_mt), then
// ComputeStereoMatches; ms[0] = wall time of the both-eye extraction, ms[1] = of the stereo association -- the two
// brackets of REGISTER_TIMES (src/Frame.cc:196-232).  Results as oro_extract / oro_stereo_match.
int oro_stereo_frame_mt(void* hl, void* hr, const uint8_t* imgL, const uint8_t* imgR, int w, int h_, long stride, float bf,
                        float b, KeyPoint* kL, uint8_t* dL, int* nL, KeyPoint* kR, uint8_t* dR, int* nR, int cap,
                        float* uRight, float* depth, double* ms) {
  Extractor *L = (Extractor*)hl, *R = (Extractor*)hr;
  std::vector<KeyPoint> kl, kr;
  std::vector<uint8_t> dl, dr;
  const auto t0 = std::chrono::steady_clock::now();
  std::thread tl([&]() { L->extract_mt(imgL, w, h_, stride, 0, 0, kl, dl); });
  std::thread tr([&]() { R->extract_mt(imgR, w, h_, stride, 0, 0, kr, dr); });
  tl.join();
  tr.join();
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<float> u, d;
  compute_stereo_matches(L->pyramid, R->pyramid, kl, dl.data(), kr, dr.data(), L->t.scale, L->t.inv_scale, bf, b, u, d);
  const auto t2 = std::chrono::steady_clock::now();
  ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
  ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
  *nL = (int)kl.size();
  *nR = (int)kr.size();
  if ((int)kl.size() > cap || (int)kr.size() > cap) return -3;
  std::memcpy(kL, kl.data(), kl.size() * sizeof(KeyPoint));
  std::memcpy(dL, dl.data(), dl.size());
  std::memcpy(kR, kr.data(), kr.size() * sizeof(KeyPoint));
  std::memcpy(dR, dr.data(), dr.size());
  std::memcpy(uRight, u.data(), u.size() * sizeof(float));
  std::memcpy(depth, d.data(), d.size() * sizeof(float));
  return 0;
}

void oro_compute_pyramid(void* h, const uint8_t* img, int w, int h_, long stride) {
  ((Extractor*)h)->compute_pyramid(img, w, h_, stride);
}
void oro_level_size(void* h, int level, int* w, int* h_) {
  Extractor* e = (Extractor*)h;
  *w = e->pyramid[level].w;
  *h_ = e->pyramid[level].h;
}
const uint8_t* oro_level_ptr(void* h, int level) { return ((Extractor*)h)->pyramid[level].px.data(); }
const uint8_t* oro_blurred_ptr(void* h, int level) { return ((Extractor*)h)->blurred[level].px.data(); }

// candidates of one level (coordinates relative to the (16,16) window origin, as in the reference).
int oro_detect_candidates(void* h, int level, KeyPoint* out, int cap) {
  std::vector<KeyPoint> c;
  ((Extractor*)h)->detect_level_candidates(level, c);
  if ((int)c.size() > cap) return -(int)c.size();
  std::memcpy(out, c.data(), c.size() * sizeof(KeyPoint));
  return (int)c.size();
}
int oro_distribute(void* h, const KeyPoint* cand, int n, int minX, int maxX, int minY, int maxY, int N,
                   KeyPoint* out, int cap) {
  std::vector<KeyPoint> c(cand, cand + n);
  std::vector<KeyPoint> r = ((Extractor*)h)->distribute_octtree(c, minX, maxX, minY, maxY, N);
  if ((int)r.size() > cap) return -3;
  std::memcpy(out, r.data(), r.size() * sizeof(KeyPoint));
  return (int)r.size();
}

void oro_resize(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh) {
  Image s(sw, sh), d;
  std::memcpy(s.px.data(), src, (size_t)sw * sh);
  resize_linear_u8(s, d, dw, dh);
  std::memcpy(dst, d.px.data(), (size_t)dw * dh);
}
int oro_fast(const uint8_t* img, int stride, int cols, int rows, int threshold, int nms, int* xys, int cap) {
  std::vector<FastPt> out;
  fast9_16(img, stride, cols, rows, threshold, nms != 0, out);
  int n = (int)out.size();
  for (int i = 0; i < n && i < cap; i++) {
    xys[3 * i] = out[i].x;
    xys[3 * i + 1] = out[i].y;
    xys[3 * i + 2] = out[i].score;
  }
  return n;
}
// cornerScore<16> of every pixel that passes the segment test at `threshold` (0 elsewhere): what cv::FAST writes into
// its rolling score rows before non-max suppression.  img is a standalone cols x rows image.
void oro_fast_score_map(const uint8_t* img, int stride, int cols, int rows, int threshold, uint8_t* out) {
  std::memset(out, 0, (size_t)cols * rows);
  std::vector<FastPt> pts;
  fast9_16(img, stride, cols, rows, threshold, false, pts);
  int pixel[25];
  static const int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
  static const int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
  for (int k = 0; k < 25; k++) pixel[k] = dy[k % 16] * stride + dx[k % 16];
  for (const FastPt& q : pts)
    out[(size_t)q.y * cols + q.x] = (uint8_t)fast_corner_score16(img + (size_t)q.y * stride + q.x, pixel, threshold);
}
void oro_blur(const uint8_t* src, int w, int h, uint8_t* dst, int variant) {
  Image s(w, h), d;
  std::memcpy(s.px.data(), src, (size_t)w * h);
  gaussian_blur7(s, d, blur_variant_taps(variant), blur_variant_vec(variant));
  std::memcpy(dst, d.px.data(), (size_t)w * h);
}
float oro_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void oro_fast_atan2_n(const float* y, const float* x, float* out, long n) {
  for (long i = 0; i < n; i++) out[i] = fast_atan2(y[i], x[i]);
}
// the tables cv::resize's linear path builds for one axis (see resize_axis_coefs): ofs[d], ab[2 d] = {a0, a1} per destination index
void oro_resize_coefs(int s, int d, int clamp_x, int* ofs, short* ab) {
  std::vector<int> o;
  std::vector<short> a;
  resize_axis_coefs(s, d, clamp_x != 0, o, a, nullptr);
  std::memcpy(ofs, o.data(), (size_t)d * sizeof(int));
  std::memcpy(ab, a.data(), (size_t)2 * d * sizeof(short));
}
void oro_sincosf(float a, float* s, float* c) { orb_sincosf(a, s, c); }
void oro_set_sincos_mode(int mode) { orb_set_sincos_mode(mode); }
int oro_get_sincos_mode() { return orb_get_sincos_mode(); }
int oro_host_libm_variant() { return orb_host_libm_variant(); }
void oro_sincos_model(float a, int fused, float* s, float* c) {
  *s = glibc_sinf_model(a, fused != 0);
  *c = glibc_cosf_model(a, fused != 0);
}
// host libm over an array (the reference's dependency itself; used to check the device's restatement)
void oro_libm_sincos_array(const float* a, long long n, float* s, float* c) {
  for (long long i = 0; i < n; i++) {
    s[i] = sinf(a[i]);
    c[i] = cosf(a[i]);
  }
}
// the angles orb_sincos_check draws: fastAtan2 of random integer moments times factorPI
void oro_reachable_angles(uint64_t seed, long long n, float* out) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  uint64_t st = seed;
  for (long long i = 0; i < n; i++) {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const int sh = (int)((z >> 60) & 15);
    const int m01 = (int)((int64_t)(z & 0x3FFFFF) - 0x200000) >> sh, m10 = (int)((int64_t)((z >> 24) & 0x3FFFFF) - 0x200000) >> sh;
    out[i] = fast_atan2((float)m01, (float)m10) * factorPI;
  }
}
long long oro_sincos_check(uint64_t seed, long long n, int fused, float* first_bad) {
  return orb_sincos_check(seed, n, fused, first_bad);
}
int oro_cv_round_f(float v) { return cv_round(v); }
float oro_ic_angle(const uint8_t* img, int w, int h, int cx, int cy) {
  Image s(w, h);
  std::memcpy(s.px.data(), img, (size_t)w * h);
  Extractor e(1000, 1.2f, 8, 20, 7);
  return ic_angle(s, cx, cy, e.t.umax);
}
void oro_descriptor(const uint8_t* blurred, int w, int h, float px, float py, float angle, uint8_t* out) {
  Image s(w, h);
  std::memcpy(s.px.data(), blurred, (size_t)w * h);
  orb_descriptor(s, px, py, angle, out);
}
int oro_hamming(const uint8_t* a, const uint8_t* b) { return descriptor_distance(a, b); }

// Stereo association on the pyramids held by two oracle extractors (after oro_extract on each).
void oro_stereo_match(void* hl, void* hr, const KeyPoint* kL, const uint8_t* dL, int nL, const KeyPoint* kR,
                      const uint8_t* dR, int nR, float bf, float b, float* uRight, float* depth) {
  Extractor* L = (Extractor*)hl;
  Extractor* R = (Extractor*)hr;
  std::vector<KeyPoint> kl(kL, kL + nL), kr(kR, kR + nR);
  std::vector<float> u, d;
  compute_stereo_matches(L->pyramid, R->pyramid, kl, dL, kr, dR, L->t.scale, L->t.inv_scale, bf, b, u, d);
  std::memcpy(uRight, u.data(), nL * sizeof(float));
  std::memcpy(depth, d.data(), nL * sizeof(float));
}

void oro_bf_knn2(const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, int* idx2, int* dist2, uint8_t* ok) {
  std::vector<int> i2, d2;
  std::vector<uint8_t> r;
  bf_knn2(dQ, nQ, dT, nT, i2, d2, r);
  std::memcpy(idx2, i2.data(), i2.size() * sizeof(int));
  std::memcpy(dist2, d2.data(), d2.size() * sizeof(int));
  std::memcpy(ok, r.data(), r.size());
}

int oro_search_init(const KeyPoint* k1, const uint8_t* d1, int n1, const KeyPoint* k2, const uint8_t* d2,
                    int n2, float minX, float minY, float maxX, float maxY, float* prevMatched,
                    int* matches12, int windowSize, float nnratio, int checkOri) {
  std::vector<KeyPoint> a(k1, k1 + n1), b(k2, k2 + n2);
  FrameGrid g;
  g.build(b, minX, minY, maxX, maxY);
  std::vector<float> prev(prevMatched, prevMatched + 2 * n1);
  std::vector<int> m;
  int n = search_for_initialization(a, d1, b, d2, g, prev, m, windowSize, nnratio, checkOri != 0);
  std::memcpy(prevMatched, prev.data(), prev.size() * sizeof(float));
  std::memcpy(matches12, m.data(), m.size() * sizeof(int));
  return n;
}

int oro_features_in_area(const KeyPoint* k, int n, float minX, float minY, float maxX, float maxY, float x,
                         float y, float r, int minLevel, int maxLevel, int* out, int cap) {
  std::vector<KeyPoint> a(k, k + n);
  FrameGrid g;
  g.build(a, minX, minY, maxX, maxY);
  std::vector<int> v = g.features_in_area(a, x, y, r, minLevel, maxLevel);
  for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
  return (int)v.size();
}

int oro_search_by_projection(const KeyPoint* k, const uint8_t* desc, const float* uRight, int n, float minX, float minY,
                             float maxX, float maxY, const float* scaleFactors, int nlevels, const MapPointView* mps,
                             int nmp, float th, int bFarPoints, float thFarPoints, float nnratio, uint8_t* occupied,
                             int* match) {
  std::vector<KeyPoint> a(k, k + n);
  FrameGrid g;
  g.build(a, minX, minY, maxX, maxY);
  std::vector<float> sf(scaleFactors, scaleFactors + nlevels);
  std::vector<MapPointView> m(mps, mps + nmp);
  std::vector<uint8_t> occ(occupied, occupied + n);
  std::vector<int> mt;
  const int nm = search_by_projection_map(a, desc, uRight, g, sf, m, th, bFarPoints != 0, thFarPoints, nnratio, occ, mt);
  std::memcpy(occupied, occ.data(), n);
  std::memcpy(match, mt.data(), n * sizeof(int));
  return nm;
}

int oro_search_by_projection_frame(const KeyPoint* k, const uint8_t* desc, const float* uRight, int n, float minX,
                                   float minY, float maxX, float maxY, const ProjectedPoint* pts, int npts, int checkOri,
                                   uint8_t* occupied, int* match) {
  std::vector<KeyPoint> a(k, k + n);
  FrameGrid g;
  g.build(a, minX, minY, maxX, maxY);
  std::vector<ProjectedPoint> p(pts, pts + npts);
  std::vector<uint8_t> occ(occupied, occupied + n);
  std::vector<int> mt;
  const int nm = search_by_projection_frame(a, desc, uRight, g, p, checkOri != 0, occ, mt);
  std::memcpy(occupied, occ.data(), n);
  std::memcpy(match, mt.data(), n * sizeof(int));
  return nm;
}

// cam = fx fy cx cy k0 k1 k2 k3 ; rig = cam1[8] cam2[8] precision R12[9] t12[3] (29 floats, = orbx_kb8_rig)
static KB8 kb8_from(const float* cam, float precision) {
  KB8 c;
  std::memcpy(c.p, cam, sizeof(c.p));
  c.precision = precision;
  return c;
}
void oro_kb8_project(const float* cam, const float* X, float* uv) { kb8_project(kb8_from(cam, 1e-6f), X, uv); }
void oro_kb8_unproject(const float* cam, float precision, float u, float v, float* ray) {
  kb8_unproject(kb8_from(cam, precision), u, v, ray);
}
void oro_null_vector4(const float* A, float* v) { smallest_right_singular_vector(A, v); }
float oro_kb8_triangulate(const float* rig, float u1, float v1, float u2, float v2, float sigma1, float sigma2, float* p3D,
                          float* gate) {
  return kb8_triangulate_matches(kb8_from(rig, rig[16]), kb8_from(rig + 8, rig[16]), u1, v1, u2, v2, rig + 17, rig + 26,
                                 sigma1, sigma2, p3D, gate);
}
// Study hook (tools/svd_gate_study.py): the same routine with the 4x4 system returned (A_out, may be null) and / or the
// null vector supplied by the caller (xh, may be null) instead of the oracle's double-precision Jacobi.
float oro_kb8_triangulate_ex(const float* rig, float u1, float v1, float u2, float v2, float sigma1, float sigma2,
                             const float* xh, float* A_out, float* p3D, float* gate) {
  return kb8_triangulate_matches(kb8_from(rig, rig[16]), kb8_from(rig + 8, rig[16]), u1, v1, u2, v2, rig + 17, rig + 26,
                                 sigma1, sigma2, p3D, gate, xh, A_out);
}
int oro_fisheye_stereo_match(const KeyPoint* kL, const uint8_t* dL, int nL, int monoL, const KeyPoint* kR, const uint8_t* dR,
                             int nR, int monoR, const float* rig, const float* levelSigma2, int nLevels, int* leftToRight,
                             int* rightToLeft, float* depth, float* p3D, int* descMatches, float* gates /* nL x 6 or null */) {
  std::vector<KeyPoint> a(kL, kL + nL), b(kR, kR + nR);
  std::vector<float> s2(levelSigma2, levelSigma2 + nLevels), dep, pts, gt;
  std::vector<int> l2r, r2l;
  const int nm = compute_stereo_fisheye_matches(a, dL, monoL, b, dR, monoR, kb8_from(rig, rig[16]), kb8_from(rig + 8, rig[16]),
                                                rig + 17, rig + 26, s2, l2r, r2l, dep, pts, descMatches, gates ? &gt : nullptr);
  std::memcpy(leftToRight, l2r.data(), nL * sizeof(int));
  std::memcpy(rightToLeft, r2l.data(), nR * sizeof(int));
  std::memcpy(depth, dep.data(), nL * sizeof(float));
  std::memcpy(p3D, pts.data(), (size_t)nL * 3 * sizeof(float));
  if (gates) std::memcpy(gates, gt.data(), (size_t)nL * 6 * sizeof(float));
  return nm;
}

int oro_search_by_projection_fisheye(const KeyPoint* k, const uint8_t* desc, int nLeft, int nRight, float minX, float minY,
                                      float maxX, float maxY, const float* scaleFactors, int nlevels, const MapPointView* mps,
                                      const MapPointRight* mpsR, int nmp, float th, int bFarPoints, float thFarPoints,
                                      float nnratio, const int* l2r, const int* r2l, uint8_t* occupied, int* match) {
  const int n = nLeft + nRight;
  std::vector<KeyPoint> a(k, k + n), kl(k, k + nLeft), kr(k + nLeft, k + n);
  FrameGrid gl, gr;
  gl.build(kl, minX, minY, maxX, maxY);
  gr.build(kr, minX, minY, maxX, maxY);
  std::vector<float> sf(scaleFactors, scaleFactors + nlevels);
  std::vector<MapPointView> m(mps, mps + nmp);
  std::vector<MapPointRight> mr(mpsR, mpsR + nmp);
  std::vector<int> a12(l2r, l2r + nLeft), a21(r2l, r2l + nRight), mt;
  std::vector<uint8_t> occ(occupied, occupied + n);
  const int nm = search_by_projection_map_fisheye(a, desc, nLeft, gl, gr, sf, m, mr, th, bFarPoints != 0, thFarPoints, nnratio, a12,
                                                  a21, occ, mt);
  std::memcpy(occupied, occ.data(), n);
  std::memcpy(match, mt.data(), n * sizeof(int));
  return nm;
}

int oro_search_by_projection_frame_fisheye(const KeyPoint* k, const uint8_t* desc, int nLeft, int nRight, float minX, float minY,
                                            float maxX, float maxY, const ProjectedPoint* pts, const float* uvRight, int npts,
                                            int checkOri, uint8_t* occupied, int* match) {
  const int n = nLeft + nRight;
  std::vector<KeyPoint> a(k, k + n), kl(k, k + nLeft), kr(k + nLeft, k + n);
  FrameGrid gl, gr;
  gl.build(kl, minX, minY, maxX, maxY);
  gr.build(kr, minX, minY, maxX, maxY);
  std::vector<ProjectedPoint> p(pts, pts + npts);
  std::vector<uint8_t> occ(occupied, occupied + n);
  std::vector<int> mt;
  const int nm = search_by_projection_frame_fisheye(a, desc, nLeft, gl, gr, p, uvRight, checkOri != 0, occ, mt);
  std::memcpy(occupied, occ.data(), n);
  std::memcpy(match, mt.data(), n * sizeof(int));
  return nm;
}

void oro_cvt_gray(const uint8_t* src, int w, int h, long src_stride, int cn, int rgb, uint8_t* dst, long dst_stride, int variant) {
  cvt_gray_u8(src, w, h, src_stride, cn, rgb != 0, dst, dst_stride, variant);
}
void oro_resize_c(const uint8_t* src, int sw, int sh, long src_stride, int cn, uint8_t* dst, int dw, int dh, long dst_stride) {
  resize_linear_u8c(src, sw, sh, src_stride, cn, dst, dw, dh, dst_stride);
}

void oro_remap(const uint8_t* src, int sw, int sh, long src_stride, const float* mapx, const float* mapy, long map_stride,
               uint8_t* dst, int dw, int dh, long dst_stride) {
  remap_linear_u8(src, sw, sh, src_stride, mapx, mapy, map_stride, dst, dw, dh, dst_stride);
}
void oro_clahe(const uint8_t* src, int w, int h, long src_stride, double clip, int tx, int ty, uint8_t* dst, long dst_stride) {
  clahe_u8(src, w, h, src_stride, clip, tx, ty, dst, dst_stride);
}

void oro_undistort_keypoints(const KeyPoint* k, int n, const float* K, const float* dist, int n_dist, KeyPoint* out) {
  std::vector<KeyPoint> a(k, k + n), o;
  undistort_keypoints(a, K, dist, n_dist, o);
  std::memcpy(out, o.data(), (size_t)n * sizeof(KeyPoint));
}
void oro_image_bounds(int cols, int rows, const float* K, const float* dist, int n_dist, float* bounds) {
  compute_image_bounds(cols, rows, K, dist, n_dist, bounds);
}

// libstdc++ std::sort with the (count, UL.x) comparator of compareNodes (src/ORBextractor.cc:542-555) on
// packed 64-bit elements (key = bits 16..63): the tie order the device quadtree's replica must reproduce.
void oro_std_sort_keys(uint64_t* v, int n) {
  std::sort(v, v + n, [](uint64_t a, uint64_t b) { return (a >> 16) < (b >> 16); });
}

// ---- f4: bag of words ------------------------------------------------------------------------------------------------
void* oro_voc_create(int k, int L, int scoring, int weighting, int n, const int* parent, const uint8_t* isLeaf,
                     const uint8_t* desc, const double* weight) {
  Vocabulary* v = new Vocabulary();
  v->build(k, L, scoring, weighting, n, parent, isLeaf, desc, weight);
  return v;
}
void* oro_voc_load(const char* path) {
  Vocabulary* v = new Vocabulary();
  if (!v->load_text(path)) { delete v; return nullptr; }
  return v;
}
int oro_voc_save(void* h, const char* path) { return static_cast<Vocabulary*>(h)->save_text(path) ? 0 : -1; }
void oro_voc_destroy(void* h) { delete static_cast<Vocabulary*>(h); }
void oro_voc_info(void* h, int* out) {
  const Vocabulary* v = static_cast<Vocabulary*>(h);
  out[0] = v->k; out[1] = v->L; out[2] = (int)v->parent.size(); out[3] = v->nWords; out[4] = v->scoring; out[5] = v->weighting;
}
void oro_voc_export(void* h, int* parent, uint8_t* isLeaf, uint8_t* desc, double* weight) {
  const Vocabulary* v = static_cast<Vocabulary*>(h);
  const int n = (int)v->parent.size();
  for (int i = 0; i < n; i++) {
    parent[i] = v->parent[i];
    isLeaf[i] = i > 0 && v->wordId[i] >= 0;
    weight[i] = v->weight[i];
  }
  std::memcpy(desc, v->desc.data(), (size_t)n * 32);
}
void oro_bow_transform_one(void* h, const uint8_t* desc, int n, int levelsup, int* word, double* weight, int* node) {
  for (int i = 0; i < n; i++) bow_transform_one(*static_cast<Vocabulary*>(h), desc + (size_t)i * 32, levelsup, word[i], weight[i], node[i]);
}
// outputs sized n (words, values, nodes, feats) and n + 1 (nodeStart); counts[0] = words, counts[1] = nodes, counts[2] = features
void oro_bow_transform(void* h, const uint8_t* desc, int n, int levelsup, uint32_t* words, double* values, uint32_t* nodes,
                       int* nodeStart, uint32_t* feats, int* counts) {
  std::vector<uint32_t> w, nd, ft;
  std::vector<double> val;
  std::vector<int> st;
  bow_transform(*static_cast<Vocabulary*>(h), desc, n, levelsup, w, val, nd, st, ft);
  std::copy(w.begin(), w.end(), words);
  std::copy(val.begin(), val.end(), values);
  std::copy(nd.begin(), nd.end(), nodes);
  std::copy(st.begin(), st.end(), nodeStart);
  std::copy(ft.begin(), ft.end(), feats);
  counts[0] = (int)w.size(); counts[1] = (int)nd.size(); counts[2] = (int)ft.size();
}
int oro_search_by_bow(const uint32_t* kfNodes, int nKfNodes, const int* kfStart, const uint32_t* kfFeat, const uint8_t* kfDesc,
                      const float* kfAngle, const uint8_t* kfValid, const uint32_t* fNodes, int nFNodes, const int* fStart,
                      const uint32_t* fFeat, const uint8_t* fDesc, const float* fAngle, int nF, int nLeftF, float nnratio,
                      int checkOri, int* match) {
  std::vector<uint32_t> a(kfNodes, kfNodes + nKfNodes), af(kfFeat, kfFeat + kfStart[nKfNodes]);
  std::vector<uint32_t> b(fNodes, fNodes + nFNodes), bf(fFeat, fFeat + fStart[nFNodes]);
  std::vector<int> as(kfStart, kfStart + nKfNodes + 1), bs(fStart, fStart + nFNodes + 1), m;
  const int n = search_by_bow(a, as, af, kfDesc, kfAngle, kfValid, b, bs, bf, fDesc, fAngle, nF, nLeftF, nnratio, checkOri != 0, m);
  std::copy(m.begin(), m.end(), match);
  return n;
}

int oro_search_by_projection_keyframe(const KeyPoint* k, const uint8_t* desc, int n, float minX, float minY, float maxX,
                                      float maxY, const ProjectedPoint* pts, int npts, int orbDist, int checkOri,
                                      uint8_t* occupied, int* match) {
  std::vector<KeyPoint> a(k, k + n);
  FrameGrid g;
  g.build(a, minX, minY, maxX, maxY);
  std::vector<ProjectedPoint> p(pts, pts + npts);
  std::vector<uint8_t> occ(occupied, occupied + n);
  std::vector<int> mt;
  const int nm = search_by_projection_keyframe(a, desc, g, p, orbDist, checkOri != 0, occ, mt);
  std::memcpy(occupied, occ.data(), n);
  std::memcpy(match, mt.data(), n * sizeof(int));
  return nm;
}

int oro_search_for_triangulation(const uint32_t* nodes1, int nNodes1, const int* start1, const uint32_t* feat1, const KeyPoint* k1,
                                 const uint8_t* d1, const uint8_t* hasMP1, const float* uRight1, int n1, const uint32_t* nodes2,
                                 int nNodes2, const int* start2, const uint32_t* feat2, const KeyPoint* k2, const uint8_t* d2,
                                 const uint8_t* hasMP2, const float* uRight2, int n2, const float* scaleFactors2,
                                 const float* levelSigma2_2, int nLevels2, const float* ep, const float* F12, int onlyStereo,
                                 int coarse, int checkOri, int* matches12) {
  std::vector<uint32_t> a(nodes1, nodes1 + nNodes1), af(feat1, feat1 + start1[nNodes1]);
  std::vector<uint32_t> b(nodes2, nodes2 + nNodes2), bf(feat2, feat2 + start2[nNodes2]);
  std::vector<int> as(start1, start1 + nNodes1 + 1), bs(start2, start2 + nNodes2 + 1), m;
  std::vector<KeyPoint> ka(k1, k1 + n1), kb(k2, k2 + n2);
  std::vector<float> sf(scaleFactors2, scaleFactors2 + nLevels2), sg(levelSigma2_2, levelSigma2_2 + nLevels2);
  const int n = search_for_triangulation(a, as, af, ka, d1, hasMP1, uRight1, b, bs, bf, kb, d2, hasMP2, uRight2, sf, sg, ep, F12,
                                         onlyStereo != 0, coarse != 0, checkOri != 0, m);
  std::copy(m.begin(), m.end(), matches12);
  return n;
}

int oro_fuse_search(const KeyPoint* k, const uint8_t* desc, const float* uRight, int n, float minX, float minY, float maxX, float maxY,
                    const float* invSigma2, int nLevels, const FusePoint* pts, int npts, int maxDist, int* bestIdx, int* bestDist) {
  std::vector<KeyPoint> a(k, k + n);
  FrameGrid g;
  g.build(a, minX, minY, maxX, maxY);
  std::vector<FusePoint> p(pts, pts + npts);
  std::vector<float> is(invSigma2, invSigma2 + nLevels);
  std::vector<int> bi, bd;
  const int nf = fuse_search(a, desc, uRight, g, is, p, maxDist, bi, bd);
  std::copy(bi.begin(), bi.end(), bestIdx);
  std::copy(bd.begin(), bd.end(), bestDist);
  return nf;
}

int oro_search_by_bow_keyframes(const uint32_t* nodes1, int nNodes1, const int* start1, const uint32_t* feat1, const uint8_t* d1,
                                const float* angle1, const uint8_t* valid1, int n1, const uint32_t* nodes2, int nNodes2,
                                const int* start2, const uint32_t* feat2, const uint8_t* d2, const float* angle2, const uint8_t* valid2,
                                int n2, float nnratio, int checkOri, int* matches12) {
  std::vector<uint32_t> a(nodes1, nodes1 + nNodes1), af(feat1, feat1 + start1[nNodes1]);
  std::vector<uint32_t> b(nodes2, nodes2 + nNodes2), bf(feat2, feat2 + start2[nNodes2]);
  std::vector<int> as(start1, start1 + nNodes1 + 1), bs(start2, start2 + nNodes2 + 1), m;
  const int n = search_by_bow_keyframes(a, as, af, d1, angle1, valid1, n1, b, bs, bf, d2, angle2, valid2, n2, nnratio, checkOri != 0, m);
  std::copy(m.begin(), m.end(), matches12);
  return n;
}

int oro_search_for_triangulation_rig(const uint32_t* nodes1, int nNodes1, const int* start1, const uint32_t* feat1, const KeyPoint* k1,
                                     const uint8_t* d1, const uint8_t* hasMP1, int nLeft1, int n1, const uint32_t* nodes2, int nNodes2,
                                     const int* start2, const uint32_t* feat2, const KeyPoint* k2, const uint8_t* d2,
                                     const uint8_t* hasMP2, int nLeft2, int n2, const float* sigma1, const float* sigma2, int nLevels,
                                     const TriRig* rig, int onlyStereo, int coarse, int checkOri, int* matches12, uint8_t* borderline) {
  std::vector<uint32_t> a(nodes1, nodes1 + nNodes1), af(feat1, feat1 + start1[nNodes1]);
  std::vector<uint32_t> b(nodes2, nodes2 + nNodes2), bf(feat2, feat2 + start2[nNodes2]);
  std::vector<int> as(start1, start1 + nNodes1 + 1), bs(start2, start2 + nNodes2 + 1), m;
  std::vector<KeyPoint> ka(k1, k1 + n1), kb(k2, k2 + n2);
  std::vector<float> s1(sigma1, sigma1 + nLevels), s2(sigma2, sigma2 + nLevels);
  std::vector<uint8_t> bl;
  const int n = search_for_triangulation_rig(a, as, af, ka, d1, hasMP1, nLeft1, b, bs, bf, kb, d2, hasMP2, nLeft2, s1, s2, *rig,
                                             onlyStereo != 0, coarse != 0, checkOri != 0, m, &bl);
  std::copy(m.begin(), m.end(), matches12);
  if (borderline) std::copy(bl.begin(), bl.end(), borderline);
  return n;
}


// Frame::isInFrustumChecks for n points x ONE camera pose of a stereo-fisheye frame (23 floats: R row-major, t, twc, 8 KB8
// parameters); views [n] receive that camera's proj_x / proj_y / view_cos / track_depth / predicted_level / in_view
void oro_is_in_frustum_kb8(const float* pose23, int n, const float* pos, const float* normal, const float* minDist, const float* maxDist,
                           float minX, float minY, float maxX, float maxY, float viewCosLimit, float logScaleFactor, int nlevels,
                           MapPointView* views, double* margins) {
  FramePoseKB8 T;
  static_assert(sizeof(FramePoseKB8) == 23 * sizeof(float), "pose = 23 packed floats");
  std::memcpy(&T, pose23, sizeof(T));
  for (int i = 0; i < n; i++) {
    double m[2];
    const MapPointView v = is_in_frustum_kb8(T, pos + 3 * i, normal + 3 * i, minDist[i], maxDist[i], minX, minY, maxX, maxY, viewCosLimit,
                                             logScaleFactor, nlevels, m);
    const MapPointView keep = views[i];
    views[i] = v;
    views[i].bad = keep.bad;
    views[i].has_observations = keep.has_observations;
    std::memcpy(views[i].desc, keep.desc, 32);
    if (margins) { margins[2 * i] = m[0]; margins[2 * i + 1] = m[1]; }
  }
}

// The projection block of SearchByProjection(CurrentFrame, LastFrame) for n LastFrame points x one pose (13 floats: quaternion
// x y z w, translation, fx fy cx cy, bf; + direction); flags bit 0 = has a MapPoint and is no outlier, bit 1 = Observations() > 0;
// views [n] (descriptors are the caller's), margins [n] (may be NULL)
void oro_project_last_frame(const float* pose12, int direction, int n, const float* pos, const int* octave, const float* angle,
                            const uint8_t* flags, float th, const float* sf, int nLevels, float minX, float minY, float maxX,
                            float maxY, ProjectedPoint* views, double* margins) {
  FramePoseQ T;
  std::memcpy(&T, pose12, 12 * sizeof(float));
  T.direction = direction;
  const std::vector<float> scale(sf, sf + nLevels);
  for (int i = 0; i < n; i++) {
    ProjectedPoint keep = views[i];
    double m = 1e30;
    ProjectedPoint v{};
    v.angle = angle[i];
    if (flags[i] & 1) v = project_last_frame_point(T, pos + 3 * i, octave[i], angle[i], th, scale, minX, minY, maxX, maxY, &m);
    v.has_observations = (flags[i] >> 1) & 1;
    std::memcpy(v.desc, keep.desc, 32);
    views[i] = v;
    if (margins) margins[i] = m;
  }
}

// Frame::isInFrustum for n points x one pose; views [n] (flags / descriptors are the caller's), margins [n][2] (may be NULL)
void oro_is_in_frustum(const float* pose20, int n, const float* pos, const float* normal, const float* minDist, const float* maxDist,
                       float minX, float minY, float maxX, float maxY, float viewCosLimit, float logScaleFactor, int nlevels,
                       MapPointView* views, double* margins) {
  FramePose T;
  std::memcpy(&T, pose20, sizeof(T));
  static_assert(sizeof(FramePose) == 20 * sizeof(float), "pose = 20 packed floats");
  for (int i = 0; i < n; i++) {
    double m[2];
    const MapPointView v = is_in_frustum(T, pos + 3 * i, normal + 3 * i, minDist[i], maxDist[i], minX, minY, maxX, maxY, viewCosLimit,
                                         logScaleFactor, nlevels, m);
    const MapPointView keep = views[i];
    views[i] = v;
    views[i].bad = keep.bad;
    views[i].has_observations = keep.has_observations;
    std::memcpy(views[i].desc, keep.desc, 32);
    if (margins) { margins[2 * i] = m[0]; margins[2 * i + 1] = m[1]; }
  }
}

}  // extern "C"
