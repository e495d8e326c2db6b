// orb_oracle.cpp — CPU oracle (see orb_oracle.h header: TEST INFRASTRUCTURE, PARITY UNPINNED).
// Every function cites the reference file:line (relative to /root/reference) or the SURVEY.md
// Appendix-B item (OpenCV kernel restated from its published algorithm) it follows.
// Build: g++ -O2 -std=c++17 -ffp-contract=off (x86-64 baseline: no FMA, like the reference's -O3 build).
#include "orb_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <thread>

namespace orbo {

// ======================================================================================= B1 rounding
// cvRound = round-half-to-even (SSE cvtss2si under the default MXCSR mode).
int cv_round(float v) { return (int)std::lrint(v); }
int cv_round(double v) { return (int)std::lrint(v); }
static inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(float v) { int i = (int)v; return i + (i < v); }
static inline short sat_short_from_float(float v) {
  int i = cv_round(v);
  return (short)(i < SHRT_MIN ? SHRT_MIN : i > SHRT_MAX ? SHRT_MAX : i);
}

// ======================================================================================= B5 fastAtan2
// cv::fastAtan2 scalar path (degrees, 7th-order odd polynomial, separate mul/add roundings).
float fast_atan2(float y, float x) {
  const float s = (float)(180.0 / 3.1415926535897932384626433832795);
  static const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s,
                     p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
  const float eps = (float)2.2204460492503131e-16;  // (float)DBL_EPSILON
  float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + eps);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + eps);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ======================================================================================= B8 sinf/cosf
// The reference calls libm: `(float)cos(angle), (float)sin(angle)` with a float argument under `using namespace std`
// (src/ORBextractor.cc:66-67,106-107) resolves to std::cos(float) / std::sin(float) = cosf / sinf (GCC may merge the
// pair into one sincosf call; glibc's three entry points run the same arithmetic per result).  The oracle's DEFAULT
// (mode 1) is its restatement of glibc >= 2.28's sinf / cosf below -- deterministic on every host and the very definition
// the device runs (csrc/orbx_sincos.h); mode 0 calls the host's libm instead, which tests/test_oracle_kernels.py uses to
// check that the restatement EQUALS the reference's own dependency on a glibc x86-64 host (skipped elsewhere).
//
// glibc's sinf/cosf (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, s_sincosf_data.c; ARM optimized-routines
// algorithm) evaluate in double: |y| < pi/4 directly, otherwise n = round(y * 2/pi) by a scaled int conversion,
// r = y - n * pi/2, then a degree-7 sine or degree-8 cosine polynomial picked by the quadrant.  x86-64 builds select
// one of two ifunc variants of the SAME C code at load time: __sinf_fma (every a + b*c fused, CPUs with FMA + AVX2)
// and __sinf_sse2 (separate multiply and add).  On the path's domain the two agree on EVERY argument: tools/sincos_sweep
// ran all 1.09e9 floats of [0, 2 pi] through both, 0 mismatches (they can differ in the last bit only for larger
// arguments, which fastAtan2 x factorPI never produces).  glibc_sinf_model / glibc_cosf_model below restate both variants
// (operation order read off the glibc 2.35 objects of this image: libm-2.35.a, s_sinf-fma.o / s_sinf-sse2.o /
// s_cosf-*.o / s_sincosf-fma.o; coefficients from s_sincosf_data.o); modes 1 / 2 select them so that fixtures can
// be generated for a stated variant, and tests check model == host libm over >= 10^7 fastAtan2-reachable angles and
// (tools/sincos_sweep) over every float in [0, 2 pi].  The device runs the same two variants (csrc/orbx_sincos.h).
namespace {
struct SinCosTab {
  double sign[4], hpi_inv, hpi, c0, c1, s1, c2, s2, c3, s3, c4;
};
const SinCosTab kSinCosTab[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2,
     -0x1.555545995a603p-3, 0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, -0x1.6c087e89a359dp-10,
     -0x1.994eb3774cf24p-13, 0x1.99343027bf8c3p-16},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2,
     -0x1.555545995a603p-3, -0x1.55553e1068f19p-5, 0x1.1107605230bc4p-7, 0x1.6c087e89a359dp-10,
     -0x1.994eb3774cf24p-13, -0x1.99343027bf8c3p-16}};
inline double mul_add(double a, double b, double c, bool fused) { return fused ? std::fma(a, b, c) : a * b + c; }
// sinf_poly of sincosf.h: n even -> sine of x (x2 = x*x up to the sign folded into x), n odd -> cosine
inline float sinf_poly_model(double x, double x2, const SinCosTab& p, int n, bool fused) {
  if ((n & 1) == 0) {
    const double x3 = x * x2;
    const double s1 = mul_add(x2, p.s3, p.s2, fused);
    const double x7 = x3 * x2;
    const double s = mul_add(x3, p.s1, x, fused);
    return (float)mul_add(x7, s1, s, fused);
  }
  const double x4 = x2 * x2;
  const double c2 = mul_add(x2, p.c4, p.c3, fused);
  const double c1 = mul_add(x2, p.c1, p.c0, fused);
  const double x6 = x4 * x2;
  const double c = mul_add(x4, p.c2, c1, fused);
  return (float)mul_add(x6, c2, c, fused);
}
inline uint32_t abstop12(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  return (u >> 20) & 0x7ff;
}
inline double reduce_fast_model(double x, const SinCosTab& p, int* np, bool fused) {
  const double r = x * p.hpi_inv;
  const int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return fused ? std::fma(-(double)n, p.hpi, x) : x - (double)n * p.hpi;
}
int g_sincos_mode = 1;  // 0 = host libm, 1 = glibc model with fused multiply-adds (default), 2 = glibc model, SSE2 variant
}  // namespace

// Valid for |y| < 120 (the path only produces y in [0, 2 pi]); larger arguments take glibc's reduce_large, not restated.
float glibc_sinf_model(float y, bool fused) {
  double x = y;
  const SinCosTab* p = &kSinCosTab[0];
  if (abstop12(y) < 0x3f4) {  // |y| < pi/4
    if (abstop12(y) < 0x398) return y;  // |y| < 2^-12
    return sinf_poly_model(x, x * x, *p, 0, fused);
  }
  int n;
  x = reduce_fast_model(x, *p, &n, fused);
  const double s = p->sign[n & 3];
  if (n & 2) p = &kSinCosTab[1];
  return sinf_poly_model(x * s, x * x, *p, n, fused);
}
float glibc_cosf_model(float y, bool fused) {
  double x = y;
  const SinCosTab* p = &kSinCosTab[0];
  if (abstop12(y) < 0x3f4) {
    if (abstop12(y) < 0x398) return 1.0f;
    return sinf_poly_model(x, x * x, *p, 1, fused);
  }
  int n;
  x = reduce_fast_model(x, *p, &n, fused);
  const double s = p->sign[n & 3];
  if (n & 2) p = &kSinCosTab[1];
  return sinf_poly_model(x * s, x * x, *p, n ^ 1, fused);
}

void orb_set_sincos_mode(int mode) { g_sincos_mode = mode; }
int orb_get_sincos_mode() { return g_sincos_mode; }
void orb_sincosf(float ang, float* s_out, float* c_out) {
  if (g_sincos_mode == 0) {  // the reference's own dependency on this machine
    *s_out = sinf(ang);
    *c_out = cosf(ang);
  } else {
    *s_out = glibc_sinf_model(ang, g_sincos_mode == 1);
    *c_out = glibc_cosf_model(ang, g_sincos_mode == 1);
  }
}

// Which glibc variant this host's libm runs: 1 = FMA, 2 = SSE2, 0 = neither model matches (not glibc >= 2.28 / x86-64).
// Sweeps every 2^11-th float of [0, 2 pi] (>= 5e5 arguments; the two variants differ on ~1e-3 of them).
int orb_host_libm_variant() {
  bool okF = true, okS = true;
  for (uint32_t u = 0; u <= 0x40C91000u; u += 2039u) {
    float y;
    std::memcpy(&y, &u, 4);
    const float hs = sinf(y), hc = cosf(y);
    okF = okF && hs == glibc_sinf_model(y, true) && hc == glibc_cosf_model(y, true);
    okS = okS && hs == glibc_sinf_model(y, false) && hc == glibc_cosf_model(y, false);
    if (!okF && !okS) return 0;
  }
  return okF ? 1 : (okS ? 2 : 0);
}

// Mismatch count of one model against the host libm over n angles the path can produce: fastAtan2 of random
// integer moment pairs (|m| < 2^21, the range of IC_Angle's sums) times factorPI, as computeOrbDescriptor forms it.
long long orb_sincos_check(uint64_t seed, long long n, int fused, float* first_bad) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  long long bad = 0;
  uint64_t st = seed;
  for (long long i = 0; i < n; i++) {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const int sh = (int)((z >> 60) & 15);  // mixed magnitudes: small moments give "round" angles
    const int m01 = (int)((int64_t)(z & 0x3FFFFF) - 0x200000) >> sh, m10 = (int)((int64_t)((z >> 24) & 0x3FFFFF) - 0x200000) >> sh;
    const float ang = fast_atan2((float)m01, (float)m10) * factorPI;
    if (sinf(ang) != glibc_sinf_model(ang, fused != 0) || cosf(ang) != glibc_cosf_model(ang, fused != 0)) {
      if (bad == 0 && first_bad) *first_bad = ang;
      bad++;
    }
  }
  return bad;
}

// ======================================================================================= B2 resize
// cv::resize(INTER_LINEAR) CV_8UC1, generic fixed-point path (11-bit coefficients).
// (For an exact 2x reduction OpenCV takes the INTER_AREA fast path instead; with fx = fy = 0.5 the formula below
// reduces to the same (s00 + s01 + s10 + s11 + 2) >> 2, see tests/test_oracle_kernels.py.)
// The per-axis tables of cv::resize's linear path: source index floor(f) and the 11-bit weights of S[ofs], S[ofs + 1], with
// f = (float)((d + 0.5) * scale - 0.5) -- a DOUBLE product rounded to float once.  clampX: the x axis folds the image edges into
// the table (first / last source column with weight 2048); the y axis clamps rows where they are read instead.  Exported as
// oro_resize_coefs so that tests/test_thirdparty_pins.py can hold the tables against an exact-rational model.
void resize_axis_coefs(int s, int d, bool clampX, std::vector<int>& ofs, std::vector<short>& ab, std::vector<uint8_t>* plain) {
  const double scale = 1.0 / ((double)d / s);
  ofs.assign(d, 0);
  ab.assign(2 * (size_t)d, 0);
  if (plain) plain->assign(d, 0);
  for (int i = 0; i < d; i++) {
    float f = (float)((i + 0.5) * scale - 0.5);
    int si = cv_floor(f);
    f -= si;
    if (clampX) {
      if (si < 0) { f = 0; si = 0; }
      if (si >= s - 1) { f = 0; si = s - 1; if (plain) (*plain)[i] = 1; }
    }
    ofs[i] = si;
    ab[2 * i] = sat_short_from_float((1.f - f) * 2048.f);
    ab[2 * i + 1] = sat_short_from_float(f * 2048.f);
  }
}

void resize_linear_u8(const Image& src, Image& dst, int dw, int dh) {
  dst = Image(dw, dh);
  const int sw = src.w, sh = src.h;
  std::vector<int> xofs, yofs;
  std::vector<short> alpha, beta;
  std::vector<uint8_t> xplain;  // dx >= xmax: D = S[sx] * 2048
  resize_axis_coefs(sw, dw, true, xofs, alpha, &xplain);
  resize_axis_coefs(sh, dh, false, yofs, beta, nullptr);
  std::vector<int> r0(dw), r1(dw);
  auto hrow = [&](int sy, std::vector<int>& out) {
    sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    const uint8_t* S = src.row(sy);
    for (int dx = 0; dx < dw; dx++) {
      int sx = xofs[dx];
      out[dx] = xplain[dx] ? S[sx] * 2048 : S[sx] * alpha[2 * dx] + S[sx + 1] * alpha[2 * dx + 1];
    }
  };
  for (int dy = 0; dy < dh; dy++) {
    hrow(yofs[dy], r0);
    hrow(yofs[dy] + 1, r1);
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst.row(dy);
    for (int dx = 0; dx < dw; dx++)
      D[dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
  }
}

void cvt_gray_u8(const uint8_t* src, int w, int h, ptrdiff_t src_stride, int cn, bool rgb, uint8_t* dst, ptrdiff_t dst_stride,
                 int variant) {
  const int shift = variant == 14 ? 14 : 15;
  const int cr = variant == 14 ? 4899 : 9798, cg = variant == 14 ? 9617 : 19235, cb = variant == 14 ? 1868 : 3735;
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + y * src_stride;
    uint8_t* D = dst + y * dst_stride;
    for (int x = 0; x < w; x++, S += cn) {
      const int r = rgb ? S[0] : S[2], g = S[1], b = rgb ? S[2] : S[0];
      D[x] = (uint8_t)((b * cb + g * cg + r * cr + (1 << (shift - 1))) >> shift);  // CV_DESCALE
    }
  }
}

void resize_linear_u8c(const uint8_t* src, int sw, int sh, ptrdiff_t src_stride, int cn, uint8_t* dst, int dw, int dh,
                       ptrdiff_t dst_stride) {
  const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> alpha(2 * dw), beta(2 * dh);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    alpha[2 * dx] = sat_short_from_float((1.f - fx) * 2048.f);
    alpha[2 * dx + 1] = sat_short_from_float(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    beta[2 * dy] = sat_short_from_float((1.f - fy) * 2048.f);
    beta[2 * dy + 1] = sat_short_from_float(fy * 2048.f);
  }
  auto tap = [&](int sy, int dx, int c) {
    sy = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    const uint8_t* S = src + sy * src_stride;
    const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;  // alpha[1] == 0 whenever sx is the last column
    return S[sx * cn + c] * alpha[2 * dx] + S[sx1 * cn + c] * alpha[2 * dx + 1];
  };
  for (int dy = 0; dy < dh; dy++) {
    const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
    uint8_t* D = dst + dy * dst_stride;
    for (int dx = 0; dx < dw; dx++)
      for (int c = 0; c < cn; c++)
        D[dx * cn + c] =
            (uint8_t)((((b0 * (tap(yofs[dy], dx, c) >> 4)) >> 16) + ((b1 * (tap(yofs[dy] + 1, dx, c) >> 4)) >> 16) + 2) >> 2);
  }
}

// cv::remap(src, dst, map1 (CV_32FC1 x), map2 (CV_32FC1 y), INTER_LINEAR, BORDER_CONSTANT, 0) on 8UC1
// (src/System.cc:294-295 with the maps of src/Settings.cc:557-572).  OpenCV imgwarp.cpp, restated from memory:
//  * RemapInvoker turns the float maps into fixed point per pixel: sx = cvRound(mapx * INTER_TAB_SIZE) (INTER_BITS = 5),
//    integer part saturate_cast<short>(sx >> 5), fraction index (sy & 31) * 32 + (sx & 31);
//  * remapBilinear reads the 2x2 weights from BilinearTab_i: float weights (1-fy)(1-fx) ... times 2^15
//    (INTER_REMAP_COEF_BITS), saturate_cast<short>; the only table entry whose sum is not 2^15 is fraction (0,0)
//    (32768 saturates to 32767) and initInterTab2D repairs it by adding the difference to the LAST tap: {32767,0,0,1};
//  * result = saturate_cast<uchar>((sum of tap * weight + 2^14) >> 15); taps outside the source count as borderValue 0.
static inline int sat_short_int(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
void remap_linear_u8(const uint8_t* src, int sw, int sh, ptrdiff_t src_stride, const float* mapx, const float* mapy,
                     ptrdiff_t map_stride, uint8_t* dst, int dw, int dh, ptrdiff_t dst_stride) {
  static short wtab[1024][4];
  static bool init = false;
  if (!init) {
    for (int fy = 0; fy < 32; fy++)
      for (int fx = 0; fx < 32; fx++) {
        const float vx[2] = {1.f - fx * (1.f / 32), fx * (1.f / 32)}, vy[2] = {1.f - fy * (1.f / 32), fy * (1.f / 32)};
        short* w = wtab[fy * 32 + fx];
        int isum = 0;
        for (int k1 = 0; k1 < 2; k1++)
          for (int k2 = 0; k2 < 2; k2++) {
            w[k1 * 2 + k2] = (short)sat_short_int(cv_round(vy[k1] * vx[k2] * 32768.f));
            isum += w[k1 * 2 + k2];
          }
        if (isum != 32768) w[3] = (short)(w[3] - (isum - 32768));  // ksize 2: the repaired tap is [1][1]
      }
    init = true;
  }
  for (int y = 0; y < dh; y++) {
    const float* MX = mapx + y * map_stride;
    const float* MY = mapy + y * map_stride;
    uint8_t* D = dst + y * dst_stride;
    for (int x = 0; x < dw; x++) {
      // cvRound(float) is cvtss2si: NaN and values outside int give INT_MIN (lrint would give a 64-bit indefinite)
      auto rnd = [](float t) { return std::fabs(t) < 2147483648.f ? (int)std::lrint(t) : INT_MIN; };
      const int fsx = rnd(MX[x] * 32.f), fsy = rnd(MY[x] * 32.f);
      const int sx = sat_short_int(fsx >> 5), sy = sat_short_int(fsy >> 5);
      const short* w = wtab[(fsy & 31) * 32 + (fsx & 31)];
      auto px = [&](int xx, int yy) -> int {
        return (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[yy * src_stride + xx] : 0;
      };
      const int v = px(sx, sy) * w[0] + px(sx + 1, sy) * w[1] + px(sx, sy + 1) * w[2] + px(sx + 1, sy + 1) * w[3];
      const int r = (v + (1 << 14)) >> 15;
      D[x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
  }
}

// cv::createCLAHE(clipLimit, Size(tilesX, tilesY))->apply(src, dst) on 8UC1 (Examples/Stereo-Inertial/
// stereo_inertial_tum_vi.cc:151,190-191; Examples/Stereo/stereo_tum_vi.cc:100,142-143: clip 3.0, 8x8 tiles).  OpenCV
// clahe.cpp (>= 4.0), restated from memory:
//  * an image that does not divide into the tiles is extended at the right / bottom with BORDER_REFLECT_101 by
//    tiles - (size % tiles) pixels for the histograms only;
//  * per tile: 256-bin histogram, bins clipped at max(int(clipLimit * tileArea / 256), 1), the clipped mass handed
//    back as clipped / 256 to every bin plus one count to every (256 / residual)-th bin from bin 0;
//    lut[i] = saturate_cast<uchar>(cumsum * (255.f / tileArea)) (float product, round half even);
//  * per pixel: bilinear blend of the four neighbouring tiles' lut values in float, tile coordinate x / tileW - 0.5,
//    indices clamped AFTER the weights are taken; saturate_cast<uchar> of the float result.
void clahe_u8(const uint8_t* src, int w, int h, ptrdiff_t src_stride, double clip_limit, int tiles_x, int tiles_y,
              uint8_t* dst, ptrdiff_t dst_stride) {
  int ew = w, eh = h;
  if (w % tiles_x != 0 || h % tiles_y != 0) {
    ew = w + (tiles_x - w % tiles_x);
    eh = h + (tiles_y - h % tiles_y);
  }
  auto r101 = [](int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
  };
  const int tw = ew / tiles_x, th = eh / tiles_y, area = tw * th;
  const float lut_scale = (float)255 / area;
  int clip = 0;
  if (clip_limit > 0.0) {
    clip = (int)(clip_limit * area / 256);
    if (clip < 1) clip = 1;
  }
  std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
  for (int ty = 0; ty < tiles_y; ty++)
    for (int tx = 0; tx < tiles_x; tx++) {
      int hist[256] = {0};
      for (int y = ty * th; y < (ty + 1) * th; y++)
        for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[src[r101(y, h) * src_stride + r101(x, w)]]++;
      if (clip > 0) {
        int clipped = 0;
        for (int i = 0; i < 256; i++)
          if (hist[i] > clip) {
            clipped += hist[i] - clip;
            hist[i] = clip;
          }
        const int batch = clipped / 256;
        int residual = clipped - batch * 256;
        for (int i = 0; i < 256; i++) hist[i] += batch;
        if (residual != 0) {
          const int step = std::max(256 / residual, 1);
          for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
        }
      }
      uint8_t* L = &lut[(size_t)(ty * tiles_x + tx) * 256];
      int sum = 0;
      for (int i = 0; i < 256; i++) {
        sum += hist[i];
        const int v = cv_round((float)sum * lut_scale);
        L[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      }
    }
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  for (int y = 0; y < h; y++) {
    const float tyf = y * inv_th - 0.5f;
    int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = std::max(ty1, 0);
    ty2 = std::min(ty2, tiles_y - 1);
    for (int x = 0; x < w; x++) {
      const float txf = x * inv_tw - 0.5f;
      int tx1 = cv_floor(txf), tx2 = tx1 + 1;
      const float xa = txf - tx1, xa1 = 1.0f - xa;
      tx1 = std::max(tx1, 0);
      tx2 = std::min(tx2, tiles_x - 1);
      const int v = src[y * src_stride + x];
      const float l11 = lut[(size_t)(ty1 * tiles_x + tx1) * 256 + v], l12 = lut[(size_t)(ty1 * tiles_x + tx2) * 256 + v];
      const float l21 = lut[(size_t)(ty2 * tiles_x + tx1) * 256 + v], l22 = lut[(size_t)(ty2 * tiles_x + tx2) * 256 + v];
      const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;  // built with -ffp-contract=off
      const int r = cv_round(res);
      dst[y * dst_stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
  }
}

// ======================================================================================= B3 FAST
static void make_ring16(int stride, int pixel[25]) {
  static const int off[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                 {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
  for (int k = 0; k < 16; k++) pixel[k] = off[k][0] + off[k][1] * stride;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
}

// cornerScore<16>: largest threshold for which the pixel is still a corner (= max(t,M) - 1).
int fast_corner_score16(const uint8_t* ptr, const int pixel[25], int threshold) {
  int d[25];
  const int v = ptr[0];
  for (int k = 0; k < 25; k++) d[k] = v - ptr[pixel[k]];
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min(d[k + 1], d[k + 2]);
    a = std::min(a, d[k + 3]);
    if (a <= a0) continue;
    for (int q = 4; q <= 8; q++) a = std::min(a, d[k + q]);
    a0 = std::max(a0, std::min(a, d[k]));
    a0 = std::max(a0, std::min(a, d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max(d[k + 1], d[k + 2]);
    b = std::max(b, d[k + 3]);
    for (int q = 4; q <= 5; q++) b = std::max(b, d[k + q]);
    if (b >= b0) continue;
    for (int q = 6; q <= 8; q++) b = std::max(b, d[k + q]);
    b0 = std::min(b0, std::max(b, d[k]));
    b0 = std::min(b0, std::max(b, d[k + 9]));
  }
  return -b0 - 1;
}

// cv::FAST(img, kps, threshold, nms) TYPE_9_16: three rolling u8 score rows (zero-initialised),
// a pixel is kept iff its score is strictly greater than its 8 neighbours; output in (y, x) order.
void fast9_16(const uint8_t* img, int stride, int cols, int rows, int threshold, bool nms,
              std::vector<FastPt>& out) {
  out.clear();
  if (cols < 7 || rows < 7) return;
  int pixel[25];
  make_ring16(stride, pixel);
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<uint8_t> buf((size_t)cols * 3, 0);
  std::vector<int> cpbuf((size_t)(cols + 1) * 3, 0);
  uint8_t* rowbuf[3] = {buf.data(), buf.data() + cols, buf.data() + 2 * cols};
  int* cprow[3] = {cpbuf.data(), cpbuf.data() + cols + 1, cpbuf.data() + 2 * (cols + 1)};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* p = img + (size_t)i * stride + 3;
    uint8_t* curr = rowbuf[(i - 3) % 3];
    int* cornerpos = cprow[(i - 3) % 3] + 1;
    std::memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, p++) {
        const int v = p[0];
        // 9 contiguous ring pixels all darker than v - t, or all brighter than v + t
        bool corner = false;
        for (int pol = 0; pol < 2 && !corner; pol++) {
          int count = 0;
          for (int k = 0; k < 25; k++) {
            const int x = p[pixel[k]];
            const bool hit = pol == 0 ? (x < v - threshold) : (x > v + threshold);
            if (hit) {
              if (++count > 8) { corner = true; break; }
            } else {
              count = 0;
            }
          }
        }
        if (corner) {
          cornerpos[ncorners++] = j;
          if (nms) curr[j] = (uint8_t)fast_corner_score16(p, pixel, threshold);
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = rowbuf[(i - 4 + 3) % 3];
    const uint8_t* pprev = rowbuf[(i - 5 + 3) % 3];
    const int* cp = cprow[(i - 4 + 3) % 3] + 1;
    const int n = cp[-1];
    for (int k = 0; k < n; k++) {
      const int j = cp[k];
      const int score = prev[j];
      if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                   score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                   score > curr[j] && score > curr[j + 1]))
        out.push_back({j, i - 1, score});
    }
  }
}

// ======================================================================================= B4 blur
const int kBlurTaps451[7] = {18, 34, 48, 56, 48, 34, 18};
const int kBlurTaps440[7] = {18, 34, 49, 55, 49, 34, 18};
static inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}
// GaussianBlur 7x7 sigma=2 CV_8U fixed point: u16 horizontal, u32 vertical, one final rounding.
// simd_vec (0, 16 or 32; meaningful for the 257-sum taps of OpenCV 4.0 .. 4.5.0 only): the model of those releases' VECTORISED
// vertical pass (modules/imgproc/src/smooth.simd.hpp, vlineSmoothONa_yzy_a<uint8_t, ufixedpoint16>, restated from the published
// source -- OpenCV is not in /root/reference and not in this image: parity unpinned).  Its body handles simd_vec columns per step
// for x in [0, (w / simd_vec) * simd_vec): the 8.8 rows are re-biased by -32768 into signed 16-bit lanes, multiplied by the taps
// with v_dotprod, the bias is given back as the CONSTANT 128 << 16 = 32768 * 256, and v_rshr_pack<16> adds the rounding half
// (32768) before the shift.  With taps that sum to 256 this is the exact rounded sum.  With the 257-sum taps the bias taken out is
// 32768 * 257, so the constant leaves the result 32768 short -- exactly the rounding half: the body FLOORS, the scalar loop behind
// it (the last w % simd_vec columns, ufixedpoint32 arithmetic) rounds.  Known answer: a flat image of 100 blurs to
// 257 * 257 * 100 / 65536 = 100.78 -> 100 in the body, 101 in the tail.  simd_vec = 16 for the SSE2 / NEON baseline build, 32 when
// the AVX2 dispatch of smooth.simd.hpp runs (any x86-64 host since 2013 with the default CPU_DISPATCH).
void gaussian_blur7(const Image& src, Image& dst, const int taps[7], int simd_vec) {
  const int w = src.w, h = src.h;
  dst = Image(w, h);
  int sumw = 0;
  for (int k = 0; k < 7; k++) sumw += taps[k];
  const int body_end = (simd_vec > 0 && sumw == 257) ? (w / simd_vec) * simd_vec : 0;
  std::vector<uint32_t> hp((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src.row(y);
    for (int x = 0; x < w; x++) {
      uint32_t acc = 0;
      for (int k = 0; k < 7; k++) acc += (uint32_t)taps[k] * S[reflect101(x + k - 3, w)];
      hp[(size_t)y * w + x] = acc > 65535u ? 65535u : acc;  // ufixedpoint16 saturates (257-sum taps)
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst.row(y);
    for (int x = 0; x < w; x++) {
      uint64_t acc = 0;
      for (int k = 0; k < 7; k++) acc += (uint64_t)taps[k] * hp[(size_t)reflect101(y + k - 3, h) * w + x];
      if (acc > 0xFFFFFFFFull) acc = 0xFFFFFFFFull;  // ufixedpoint32 saturates
      uint32_t v = (uint32_t)((acc + (x < body_end ? 0u : 32768u)) >> 16);
      D[x] = (uint8_t)(v > 255 ? 255 : v);
    }
  }
}

// ======================================================================================= tables
const int8_t kPattern[1024] = {
#include "../orb_slam3_fast_amd/csrc/orb_pattern31.inc"
};

static const int kPatchSize = 31, kHalfPatch = 15, kEdge = 19;

// ORBextractor::ORBextractor, src/ORBextractor.cc:408-469.
Extractor::Extractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniTh_, int minTh_)
    : nfeatures(nfeatures_), nlevels(nlevels_), iniTh(iniTh_), minTh(minTh_), scaleFactor(scaleFactor_) {
  t.scale.resize(nlevels);
  t.sigma2.resize(nlevels);
  t.scale[0] = 1.0f;
  t.sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    t.scale[i] = (float)(t.scale[i - 1] * scaleFactor);  // float * double -> float (:423)
    t.sigma2[i] = t.scale[i] * t.scale[i];
  }
  t.inv_scale.resize(nlevels);
  t.inv_sigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {
    t.inv_scale[i] = 1.0f / t.scale[i];
    t.inv_sigma2[i] = 1.0f / t.sigma2[i];
  }
  pyramid.resize(nlevels);
  blurred.resize(nlevels);
  t.nfeat_level.resize(nlevels);
  const float factor = (float)(1.0f / scaleFactor);  // :437
  float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int l = 0; l < nlevels - 1; l++) {
    t.nfeat_level[l] = cv_round(nDesired);
    sum += t.nfeat_level[l];
    nDesired *= factor;
  }
  t.nfeat_level[nlevels - 1] = std::max(nfeatures - sum, 0);
  // umax, :456-468
  t.umax.assign(kHalfPatch + 1, 0);
  const float s2 = std::sqrt(2.f);
  const int vmax = cv_floor(kHalfPatch * s2 / 2 + 1);
  const int vmin = cv_ceil(kHalfPatch * s2 / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (int v = 0; v <= vmax; ++v) t.umax[v] = cv_round(std::sqrt(hp2 - v * v));
  for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (t.umax[v0] == t.umax[v0 + 1]) ++v0;
    t.umax[v] = v0;
    ++v0;
  }
}

// ORBextractor::ComputePyramid, src/ORBextractor.cc:1108-1145 (borders are written there but never read).
void Extractor::compute_pyramid(const uint8_t* img, int w, int h, ptrdiff_t stride) {
  for (int l = 0; l < nlevels; l++) {
    const float s = t.inv_scale[l];
    const int lw = cv_round((float)w * s), lh = cv_round((float)h * s);
    if (l == 0) {
      pyramid[0] = Image(lw, lh);
      for (int y = 0; y < h; y++) std::memcpy(pyramid[0].row(y), img + y * stride, w);
    } else {
      resize_linear_u8(pyramid[l - 1], pyramid[l], lw, lh);
    }
  }
}

// Cell loop of ComputeKeyPointsOctTree, src/ORBextractor.cc:892-971 (serial variant).
void Extractor::detect_level_candidates(int level, std::vector<KeyPoint>& cand) const {
  cand.clear();
  const Image& im = pyramid[level];
  const int minBX = kEdge - 3, minBY = minBX;
  const int maxBX = im.w - kEdge + 3, maxBY = im.h - kEdge + 3;
  const float W = 35;
  const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
  const int nCols = (int)(width / W), nRows = (int)(height / W);
  const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
  std::vector<FastPt> cell;
  for (int i = 0; i < nRows; i++) {
    const float iniY = (float)(minBY + i * hCell);
    float maxY = iniY + hCell + 6;
    if (iniY >= maxBY - 3) continue;
    if (maxY > maxBY) maxY = (float)maxBY;
    for (int j = 0; j < nCols; j++) {
      const float iniX = (float)(minBX + j * wCell);
      float maxX = iniX + wCell + 6;
      if (iniX >= maxBX - 6) continue;
      if (maxX > maxBX) maxX = (float)maxBX;
      const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
      fast9_16(im.row(y0) + x0, im.w, cw, ch, iniTh, true, cell);
      if (cell.empty()) fast9_16(im.row(y0) + x0, im.w, cw, ch, minTh, true, cell);
      for (const FastPt& p : cell) {
        KeyPoint kp;
        kp.x = (float)p.x + j * wCell;
        kp.y = (float)p.y + i * hCell;
        kp.size = 7.f;
        kp.angle = -1.f;
        kp.response = (float)p.score;
        kp.octave = 0;
        kp.class_id = -1;
        cand.push_back(kp);
      }
    }
  }
}

// ORBextractor::DistributeOctTree + ExtractorNode::DivideNode + compareNodes,
// src/ORBextractor.cc:490-757.  Nodes hold indices into `cand` (order preserved = vKeys order).
namespace {
struct Node {
  int ulx, uly, urx, ury, blx, bly, brx, bry;
  std::vector<int> keys;
  bool no_more = false;
  std::list<Node>::iterator self;
};
struct SizeNode {
  int first;
  Node* second;
};
bool compare_nodes(const SizeNode& a, const SizeNode& b) {  // :542-555
  if (a.first < b.first) return true;
  if (a.first > b.first) return false;
  return a.second->ulx < b.second->ulx;
}
void divide(const Node& p, const std::vector<KeyPoint>& cand, Node n[4]) {  // :490-540
  const int halfX = (int)std::ceil((float)(p.urx - p.ulx) / 2);
  const int halfY = (int)std::ceil((float)(p.bry - p.uly) / 2);
  n[0].ulx = p.ulx; n[0].uly = p.uly; n[0].urx = p.ulx + halfX; n[0].ury = p.uly;
  n[0].blx = p.ulx; n[0].bly = p.uly + halfY; n[0].brx = p.ulx + halfX; n[0].bry = p.uly + halfY;
  n[1].ulx = n[0].urx; n[1].uly = n[0].ury; n[1].urx = p.urx; n[1].ury = p.ury;
  n[1].blx = n[0].brx; n[1].bly = n[0].bry; n[1].brx = p.urx; n[1].bry = p.uly + halfY;
  n[2].ulx = n[0].blx; n[2].uly = n[0].bly; n[2].urx = n[0].brx; n[2].ury = n[0].bry;
  n[2].blx = p.blx; n[2].bly = p.bly; n[2].brx = n[0].brx; n[2].bry = p.bly;
  n[3].ulx = n[2].urx; n[3].uly = n[2].ury; n[3].urx = n[1].brx; n[3].ury = n[1].bry;
  n[3].blx = n[2].brx; n[3].bly = n[2].bry; n[3].brx = p.brx; n[3].bry = p.bry;
  for (int k : p.keys) {
    const KeyPoint& kp = cand[k];
    if (kp.x < n[0].urx) {
      if (kp.y < n[0].bry) n[0].keys.push_back(k); else n[2].keys.push_back(k);
    } else if (kp.y < n[0].bry) {
      n[1].keys.push_back(k);
    } else {
      n[3].keys.push_back(k);
    }
  }
  for (int q = 0; q < 4; q++) n[q].no_more = n[q].keys.size() == 1;
}
}  // namespace

std::vector<KeyPoint> Extractor::distribute_octtree(const std::vector<KeyPoint>& cand, int minX,
                                                    int maxX, int minY, int maxY, int N) const {
  const int nIni = (int)std::round((float)(maxX - minX) / (maxY - minY));  // :566
  const float hX = (float)(maxX - minX) / nIni;
  std::list<Node> nodes;
  std::vector<Node*> ini(nIni);
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.ulx = (int)(hX * (float)i); ni.uly = 0;
    ni.urx = (int)(hX * (float)(i + 1)); ni.ury = 0;
    ni.blx = ni.ulx; ni.bly = maxY - minY;
    ni.brx = ni.urx; ni.bry = maxY - minY;
    nodes.push_back(ni);
    ini[i] = &nodes.back();
  }
  for (int k = 0; k < (int)cand.size(); k++) ini[(size_t)(cand[k].x / hX)]->keys.push_back(k);
  for (auto it = nodes.begin(); it != nodes.end();) {
    if (it->keys.size() == 1) { it->no_more = true; ++it; }
    else if (it->keys.empty()) it = nodes.erase(it);
    else ++it;
  }
  bool finish = false;
  std::vector<SizeNode> expandable;
  auto push_children = [&](Node n[4], int* nToExpand) {
    for (int q = 0; q < 4; q++) {
      if (n[q].keys.empty()) continue;
      nodes.push_front(n[q]);
      if (n[q].keys.size() > 1) {
        if (nToExpand) ++*nToExpand;
        expandable.push_back({(int)n[q].keys.size(), &nodes.front()});
        nodes.front().self = nodes.begin();
      }
    }
  };
  while (!finish) {
    int prevSize = (int)nodes.size();
    int nToExpand = 0;
    expandable.clear();
    for (auto it = nodes.begin(); it != nodes.end();) {
      if (it->no_more) { ++it; continue; }
      Node n[4];
      divide(*it, cand, n);
      push_children(n, &nToExpand);
      it = nodes.erase(it);
    }
    if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
      finish = true;
    } else if ((int)nodes.size() + nToExpand * 3 > N) {
      while (!finish) {
        prevSize = (int)nodes.size();
        std::vector<SizeNode> prev = expandable;
        expandable.clear();
        std::sort(prev.begin(), prev.end(), compare_nodes);  // libstdc++ introsort: tie order matters
        for (int j = (int)prev.size() - 1; j >= 0; j--) {
          Node n[4];
          divide(*prev[j].second, cand, n);
          push_children(n, nullptr);
          nodes.erase(prev[j].second->self);
          if ((int)nodes.size() >= N) break;
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
      }
    }
  }
  std::vector<KeyPoint> result;
  result.reserve(nodes.size());
  for (const Node& nd : nodes) {  // best response per node, first wins (:741-754)
    int best = nd.keys[0];
    float maxResp = cand[best].response;
    for (size_t k = 1; k < nd.keys.size(); k++)
      if (cand[nd.keys[k]].response > maxResp) { best = nd.keys[k]; maxResp = cand[best].response; }
    result.push_back(cand[best]);
  }
  return result;
}

// IC_Angle, src/ORBextractor.cc:75-99.
float ic_angle(const Image& im, int cx, int cy, const std::vector<int>& umax) {
  int m01 = 0, m10 = 0;
  const uint8_t* c = im.row(cy) + cx;
  const int step = im.w;
  for (int u = -kHalfPatch; u <= kHalfPatch; ++u) m10 += u * c[u];
  for (int v = 1; v <= kHalfPatch; ++v) {
    int vsum = 0;
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      const int plus = c[u + v * step], minus = c[u - v * step];
      vsum += plus - minus;
      m10 += u * (plus + minus);
    }
    m01 += v * vsum;
  }
  return fast_atan2((float)m01, (float)m10);
}

// ComputeKeyPointsOctTree, src/ORBextractor.cc:886-999.
void Extractor::compute_keypoints(std::vector<std::vector<KeyPoint>>& all) const {
  all.assign(nlevels, {});
  std::vector<KeyPoint> cand;
  for (int l = 0; l < nlevels; l++) {
    const int minBX = kEdge - 3, minBY = minBX;
    const int maxBX = pyramid[l].w - kEdge + 3, maxBY = pyramid[l].h - kEdge + 3;
    detect_level_candidates(l, cand);
    all[l] = distribute_octtree(cand, minBX, maxBX, minBY, maxBY, t.nfeat_level[l]);
    const int scaledPatch = (int)(kPatchSize * t.scale[l]);
    for (KeyPoint& kp : all[l]) {
      kp.x += minBX;
      kp.y += minBY;
      kp.octave = l;
      kp.size = (float)scaledPatch;
    }
  }
  for (int l = 0; l < nlevels; l++)
    for (KeyPoint& kp : all[l]) kp.angle = ic_angle(pyramid[l], cv_round(kp.x), cv_round(kp.y), t.umax);
}

// computeOrbDescriptor, src/ORBextractor.cc:101-147.
void orb_descriptor(const Image& img, float px, float py, float angle_deg, uint8_t out[32]) {
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  const float angle = angle_deg * factorPI;
  float a, b;
  orb_sincosf(angle, &b, &a);  // a = cos, b = sin
  const uint8_t* center = img.row(cv_round(py)) + cv_round(px);
  const int step = img.w;
  const int8_t* pat = kPattern;
  for (int i = 0; i < 32; i++, pat += 32) {
    int val = 0;
    for (int bit = 0; bit < 8; bit++) {
      const int8_t* p = pat + 4 * bit;
      const int t0 = center[cv_round(p[0] * b + p[1] * a) * step + cv_round(p[0] * a - p[1] * b)];
      const int t1 = center[cv_round(p[2] * b + p[3] * a) * step + cv_round(p[2] * a - p[3] * b)];
      val |= (t0 < t1) << bit;
    }
    out[i] = (uint8_t)val;
  }
}

// ORBextractor::operator(), src/ORBextractor.cc:1015-1106, level loop in serial order (:1067).
int Extractor::extract(const uint8_t* img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                       std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc) {
  if (!img || w <= 0 || h <= 0) return -1;
  compute_pyramid(img, w, h, stride);
  std::vector<std::vector<KeyPoint>> all;
  compute_keypoints(all);
  int n = 0;
  for (int l = 0; l < nlevels; l++) n += (int)all[l].size();
  kps.assign(n, KeyPoint{});
  desc.assign((size_t)n * 32, 0);
  int mono = 0, stereo = n - 1;
  for (int l = 0; l < nlevels; l++) {
    std::vector<KeyPoint>& lk = all[l];
    if (lk.empty()) continue;
    gaussian_blur7(pyramid[l], blurred[l], blur_taps, blur_simd_vec);
    const float scale = t.scale[l];
    for (KeyPoint& kp : lk) {
      uint8_t d[32];
      orb_descriptor(blurred[l], kp.x, kp.y, kp.angle, d);
      if (l != 0) { kp.x *= scale; kp.y *= scale; }
      int slot;
      if (kp.x >= (float)lap0 && kp.x <= (float)lap1) slot = stereo--; else slot = mono++;
      kps[slot] = kp;
      std::memcpy(&desc[(size_t)slot * 32], d, 32);
    }
  }
  return mono;
}

// The same operator() with the THREAD STRUCTURE of the fork (the `cpu_mt` timing baseline): pyramid serial (the
// reference leaves it serial, "TODO" src/ORBextractor.cc:1109), then one task per level for FAST cells + quadtree
// (tbb::parallel_for over levels, :764-846), one per level for the orientations (:876-884), one per level for blur +
// descriptors (:1063-1101); the output assembly runs serially afterwards in level order, so the result is identical to
// extract() (the fork's racy shared counters are not reproduced).
int Extractor::extract_mt(const uint8_t* img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                          std::vector<KeyPoint>& kps, std::vector<uint8_t>& desc) {
  if (!img || w <= 0 || h <= 0) return -1;
  compute_pyramid(img, w, h, stride);
  std::vector<std::vector<KeyPoint>> all(nlevels);
  std::vector<std::vector<uint8_t>> dl(nlevels);
  auto per_level = [&](auto&& body) {
    std::vector<std::thread> th;
    for (int l = 0; l < nlevels; l++) th.emplace_back([&, l]() { body(l); });
    for (auto& t_ : th) t_.join();
  };
  per_level([&](int l) {
    const int minBX = kEdge - 3, minBY = minBX;
    const int maxBX = pyramid[l].w - kEdge + 3, maxBY = pyramid[l].h - kEdge + 3;
    std::vector<KeyPoint> cand;
    detect_level_candidates(l, cand);
    all[l] = distribute_octtree(cand, minBX, maxBX, minBY, maxBY, t.nfeat_level[l]);
    const int scaledPatch = (int)(kPatchSize * t.scale[l]);
    for (KeyPoint& kp : all[l]) {
      kp.x += minBX;
      kp.y += minBY;
      kp.octave = l;
      kp.size = (float)scaledPatch;
    }
  });
  per_level([&](int l) {
    for (KeyPoint& kp : all[l]) kp.angle = ic_angle(pyramid[l], cv_round(kp.x), cv_round(kp.y), t.umax);
  });
  per_level([&](int l) {
    if (all[l].empty()) return;
    gaussian_blur7(pyramid[l], blurred[l], blur_taps, blur_simd_vec);
    dl[l].resize(all[l].size() * 32);
    for (size_t i = 0; i < all[l].size(); i++) orb_descriptor(blurred[l], all[l][i].x, all[l][i].y, all[l][i].angle, &dl[l][i * 32]);
  });
  int n = 0;
  for (int l = 0; l < nlevels; l++) n += (int)all[l].size();
  kps.assign(n, KeyPoint{});
  desc.assign((size_t)n * 32, 0);
  int mono = 0, stereo = n - 1;
  for (int l = 0; l < nlevels; l++) {
    const float scale = t.scale[l];
    for (size_t i = 0; i < all[l].size(); i++) {
      KeyPoint kp = all[l][i];
      if (l != 0) { kp.x *= scale; kp.y *= scale; }
      int slot;
      if (kp.x >= (float)lap0 && kp.x <= (float)lap1) slot = stereo--; else slot = mono++;
      kps[slot] = kp;
      std::memcpy(&desc[(size_t)slot * 32], &dl[l][i * 32], 32);
    }
  }
  return mono;
}

// ======================================================================================= matcher
// ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1959-1973.
int descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4);
    std::memcpy(&y, b + 4 * i, 4);
    uint32_t v = x ^ y;
    v = v - ((v >> 1) & 0x55555555u);
    v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
    dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
  }
  return dist;
}

// Frame::ComputeStereoMatches, src/Frame.cc:921-1084.
void compute_stereo_matches(const std::vector<Image>& pyrL, const std::vector<Image>& pyrR,
                            const std::vector<KeyPoint>& kL, const uint8_t* dL,
                            const std::vector<KeyPoint>& kR, const uint8_t* dR,
                            const std::vector<float>& scale, const std::vector<float>& inv_scale,
                            float bf, float b, std::vector<float>& uRight, std::vector<float>& depth) {
  const int N = (int)kL.size();
  uRight.assign(N, -1.0f);
  depth.assign(N, -1.0f);
  const int thOrbDist = (100 + 50) / 2;
  const int nRows = pyrL[0].h;
  std::vector<std::vector<size_t>> rowIdx(nRows);
  for (int iR = 0; iR < (int)kR.size(); iR++) {
    const KeyPoint& kp = kR[iR];
    if (kp.y == 0.0 && kp.x == 0.0) continue;
    const float r = 2.0f * scale[kp.octave];
    const int maxr = (int)std::ceil(kp.y + r);
    const int minr = (int)std::floor(kp.y - r);
    for (int yi = minr; yi <= maxr; yi++) rowIdx[yi].push_back(iR);
  }
  const float minZ = b, minD = 0, maxD = bf / minZ;
  std::vector<std::pair<int, int>> distIdx;
  for (int iL = 0; iL < N; iL++) {
    const KeyPoint& kpL = kL[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const std::vector<size_t>& cands = rowIdx[(size_t)vL];
    if (cands.empty()) continue;
    const float minU = uL - maxD, maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = 100;
    size_t bestIdxR = 0;
    for (size_t iR : cands) {
      const KeyPoint& kpR = kR[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = descriptor_distance(dL + (size_t)iL * 32, dR + iR * 32);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = kR[bestIdxR].x;
      const float sf = inv_scale[kpL.octave];
      const float scaleduL = std::round(kpL.x * sf);
      const float scaledvL = std::round(kpL.y * sf);
      const float scaleduR0 = std::round(uR0 * sf);
      const int w = 5, L = 5;
      const Image& imL = pyrL[kpL.octave];
      const Image& imR = pyrR[kpL.octave];
      int bestSad = INT_MAX, bestinc = 0;
      float dists[2 * L + 1];
      const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= imR.w) continue;
      const int yl = (int)(scaledvL - w), xl = (int)(scaleduL - w);
      for (int inc = -L; inc <= L; inc++) {
        const int xr = (int)(scaleduR0 + inc - w);
        long sad = 0;
        for (int yy = 0; yy < 2 * w + 1; yy++) {
          const uint8_t* pl = imL.row(yl + yy) + xl;
          const uint8_t* pr = imR.row(yl + yy) + xr;
          for (int xx = 0; xx < 2 * w + 1; xx++) sad += std::abs((int)pl[xx] - (int)pr[xx]);
        }
        const float dist = (float)(double)sad;
        if (dist < bestSad) { bestSad = (int)dist; bestinc = inc; }
        dists[L + inc] = dist;
      }
      if (bestinc == -L || bestinc == L) continue;
      const float d1 = dists[L + bestinc - 1], d2 = dists[L + bestinc], d3 = dists[L + bestinc + 1];
      const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = scale[kpL.octave] * ((float)scaleduR0 + (float)bestinc + deltaR);
      float disparity = uL - bestuR;
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01;
          bestuR = uL - 0.01;  // double arithmetic, narrowed on assignment (:1063)
        }
        depth[iL] = bf / disparity;
        uRight[iL] = bestuR;
        distIdx.push_back({bestSad, iL});
      }
    }
  }
  if (distIdx.empty()) return;  // the reference reads vDistIdx[0] here (UB): guarded
  std::sort(distIdx.begin(), distIdx.end());
  const float median = (float)distIdx[distIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)distIdx.size() - 1; i >= 0; i--) {
    if (distIdx[i].first < thDist) break;
    uRight[distIdx[i].second] = -1;
    depth[distIdx[i].second] = -1;
  }
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(Q, T, k=2): stable w.r.t. train index (B7) + Lowe test (:1302).
void bf_knn2(const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, std::vector<int>& idx2,
             std::vector<int>& dist2, std::vector<uint8_t>& ratio_ok) {
  idx2.assign((size_t)nQ * 2, -1);
  dist2.assign((size_t)nQ * 2, -1);
  ratio_ok.assign(nQ, 0);
  for (int q = 0; q < nQ; q++) {
    int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
    for (int tI = 0; tI < nT; tI++) {
      const int d = descriptor_distance(dQ + (size_t)q * 32, dT + (size_t)tI * 32);
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = tI; }
      else if (d < b1) { b1 = d; i1 = tI; }
    }
    idx2[2 * q] = i0; idx2[2 * q + 1] = i1;
    if (i0 >= 0) dist2[2 * q] = b0;
    if (i1 >= 0) dist2[2 * q + 1] = b1;
    if (i0 >= 0 && i1 >= 0 && (float)b0 < (float)b1 * 0.7) ratio_ok[q] = 1;
  }
}

// ---- KannalaBrandt8 (src/CameraModels/KannalaBrandt8.cpp) -----------------------------------------------------
// Float arithmetic in the reference's expression order.  atan2f / tanf / cosf / sinf come from the host libm and the
// reference build may contract to FMA (-march=native): this part of the path is tolerance parity by construction.
void kb8_project(const KB8& c, const float X[3], float uv[2]) {  // :67-86 (Eigen::Vector3f overload)
  const float x2_plus_y2 = X[0] * X[0] + X[1] * X[1];
  const float theta = atan2f(sqrtf(x2_plus_y2), X[2]);
  const float psi = atan2f(X[1], X[0]);
  const float theta2 = theta * theta;
  const float theta3 = theta * theta2;
  const float theta5 = theta3 * theta2;
  const float theta7 = theta5 * theta2;
  const float theta9 = theta7 * theta2;
  const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
  uv[0] = c.p[0] * r * std::cos(psi) + c.p[2];
  uv[1] = c.p[1] * r * std::sin(psi) + c.p[3];
}

void kb8_unproject(const KB8& c, float u, float v, float ray[3]) {  // :116-147
  const float pwx = (u - c.p[2]) / c.p[0], pwy = (v - c.p[3]) / c.p[1];
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  const double kPi = 3.1415926535897932384626433832795;  // CV_PI
  theta_d = fminf(fmaxf((float)(-kPi / 2.f), theta_d), (float)(kPi / 2.f));
  if (theta_d > 1e-8) {
    float theta = theta_d;
    for (int j = 0; j < 10; j++) {
      const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      const float k0_theta2 = c.p[4] * theta2, k1_theta4 = c.p[5] * theta4;
      const float k2_theta6 = c.p[6] * theta6, k3_theta8 = c.p[7] * theta8;
      const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                              (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
      theta = theta - theta_fix;
      if (fabsf(theta_fix) < c.precision) break;
    }
    scale = std::tan(theta) / theta_d;
  }
  ray[0] = pwx * scale;
  ray[1] = pwy * scale;
  ray[2] = 1.f;
}

void smallest_right_singular_vector(const float A[16], float v[4]) {
  double U[4][4], V[4][4];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      U[i][j] = A[4 * i + j];
      V[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; sweep++) {  // one-sided (Hestenes) Jacobi: orthogonalise the columns of U = A V
    bool rotated = false;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 4; i++) {
          al += U[i][p] * U[i][p];
          be += U[i][q] * U[i][q];
          ga += U[i][p] * U[i][q];
        }
        if (ga == 0.0 || std::fabs(ga) <= 1e-15 * std::sqrt(al * be)) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < 4; i++) {
          const double up = U[i][p], uq = U[i][q];
          U[i][p] = cs * up - sn * uq;
          U[i][q] = sn * up + cs * uq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = cs * vp - sn * vq;
          V[i][q] = sn * vp + cs * vq;
        }
      }
    if (!rotated) break;
  }
  int best = 0;
  double bestn = 0;
  for (int j = 0; j < 4; j++) {
    double n = 0;
    for (int i = 0; i < 4; i++) n += U[i][j] * U[i][j];
    if (j == 0 || n < bestn) {
      bestn = n;
      best = j;
    }
  }
  for (int i = 0; i < 4; i++) v[i] = (float)V[i][best];
}

float kb8_triangulate_matches(const KB8& c1, const KB8& c2, float u1, float v1, float u2, float v2, const float R12[9],
                              const float t12[3], float sigmaLevel, float unc, float p3D[3], float* gate,
                              const float* xh_override, float* A_out) {
  if (gate)
    for (int i = 0; i < 5; i++) gate[i] = NAN;
  float r1[3], r2[3], r21[3];
  kb8_unproject(c1, u1, v1, r1);
  kb8_unproject(c2, u2, v2, r2);
  for (int i = 0; i < 3; i++) r21[i] = R12[3 * i] * r2[0] + R12[3 * i + 1] * r2[1] + R12[3 * i + 2] * r2[2];  // :352
  const float dot = r1[0] * r21[0] + r1[1] * r21[1] + r1[2] * r21[2];
  const float n1 = sqrtf(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  const float n2 = sqrtf(r21[0] * r21[0] + r21[1] * r21[1] + r21[2] * r21[2]);
  const float cosParallaxRays = dot / (n1 * n2);
  if (gate) gate[0] = cosParallaxRays;
  if (cosParallaxRays > 0.9998) return -1;  // :356
  // Tcw1 = [I | 0], Tcw2 = [R21 | -R21 t12]  (:369-376)
  float T1[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}}, T2[3][4];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T2[i][j] = R12[3 * j + i];
    T2[i][3] = (-T2[i][0]) * t12[0] + (-T2[i][1]) * t12[1] + (-T2[i][2]) * t12[2];
  }
  float A[16];  // Triangulate, :420-427
  for (int j = 0; j < 4; j++) {
    A[j] = r1[0] * T1[2][j] - T1[0][j];
    A[4 + j] = r1[1] * T1[2][j] - T1[1][j];
    A[8 + j] = r2[0] * T2[2][j] - T2[0][j];
    A[12 + j] = r2[1] * T2[2][j] - T2[1][j];
  }
  float xh[4];
  if (A_out) std::memcpy(A_out, A, sizeof(A));
  if (xh_override) std::memcpy(xh, xh_override, sizeof(xh));  // study hook: null vector from another SVD (tools/svd_gate_study.py)
  else smallest_right_singular_vector(A, xh);
  const float x3D[3] = {xh[0] / xh[3], xh[1] / xh[3], xh[2] / xh[3]};
  const float z1 = x3D[2];
  if (gate) gate[1] = z1;
  if (!(z1 > 0)) return -2;  // `z1 <= 0`; NaN (xh[3] == 0) rejected as well
  const float z2 = T2[2][0] * x3D[0] + T2[2][1] * x3D[1] + T2[2][2] * x3D[2] + T2[2][3];
  if (gate) gate[2] = z2;
  if (!(z2 > 0)) return -3;
  float uv1[2];
  kb8_project(c1, x3D, uv1);
  const float errX1 = uv1[0] - u1, errY1 = uv1[1] - v1;
  if (gate) gate[3] = (float)((errX1 * errX1 + errY1 * errY1) / (5.991 * sigmaLevel));
  if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -4;
  float x3D2[3];
  for (int i = 0; i < 3; i++) x3D2[i] = T2[i][0] * x3D[0] + T2[i][1] * x3D[1] + T2[i][2] * x3D[2] + T2[i][3];
  float uv2[2];
  kb8_project(c2, x3D2, uv2);
  const float errX2 = uv2[0] - u2, errY2 = uv2[1] - v2;
  if (gate) gate[4] = (float)((errX2 * errX2 + errY2 * errY2) / (5.991 * unc));
  if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -5;
  p3D[0] = x3D[0];
  p3D[1] = x3D[1];
  p3D[2] = x3D[2];
  return z1;
}

int compute_stereo_fisheye_matches(const std::vector<KeyPoint>& kL, const uint8_t* dL, int monoL,
                                   const std::vector<KeyPoint>& kR, const uint8_t* dR, int monoR, const KB8& c1,
                                   const KB8& c2, const float R12[9], const float t12[3],
                                   const std::vector<float>& levelSigma2, std::vector<int>& leftToRight,
                                   std::vector<int>& rightToLeft, std::vector<float>& depth,
                                   std::vector<float>& p3D, int* descMatches, std::vector<float>* gates) {
  const int nL = (int)kL.size(), nR = (int)kR.size();
  leftToRight.assign(nL, -1);
  rightToLeft.assign(nR, -1);
  depth.assign(nL, -1.0f);
  p3D.assign((size_t)nL * 3, 0.0f);
  if (gates) gates->assign((size_t)nL * 6, NAN);
  const int nQ = nL - monoL, nT = nR - monoR;
  std::vector<int> idx2, dist2;
  std::vector<uint8_t> ok;
  bf_knn2(dL + (size_t)monoL * 32, nQ, dR + (size_t)monoR * 32, nT, idx2, dist2, ok);
  int nMatches = 0, nDesc = 0;
  for (int q = 0; q < nQ; q++) {
    if (!ok[q]) continue;
    nDesc++;
    const int iL = q + monoL, iR = idx2[2 * q] + monoR;
    const float sigma1 = levelSigma2[kL[iL].octave], sigma2 = levelSigma2[kR[iR].octave];
    float P[3] = {0, 0, 0};
    float* gt = gates ? gates->data() + (size_t)iL * 6 : nullptr;
    const float d = kb8_triangulate_matches(c1, c2, kL[iL].x, kL[iL].y, kR[iR].x, kR[iR].y, R12, t12, sigma1, sigma2, P, gt);
    if (gt) gt[5] = d;
    if (d > 0.0001f) {
      leftToRight[iL] = iR;
      rightToLeft[iR] = iL;
      p3D[3 * iL] = P[0];
      p3D[3 * iL + 1] = P[1];
      p3D[3 * iL + 2] = P[2];
      depth[iL] = d;
      nMatches++;
    }
  }
  if (descMatches) *descMatches = nDesc;
  return nMatches;
}

// ---- cv::undistortPoints as Frame::UndistortKeyPoints / ComputeImageBounds call it -------------------------------------
void undistort_points(const float* xy_in, int n, const float K[4], const float* dist, int n_dist, float* xy_out) {
  double k[14] = {0};
  for (int i = 0; i < n_dist && i < 14; i++) k[i] = (double)dist[i];
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < n; i++) {
    const double u = xy_in[2 * i], v = xy_in[2 * i + 1];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    if (n_dist > 0) {
      for (int j = 0; j < 5; j++) {  // TermCriteria(MAX_ITER, 5, 0.01): the count is the only active criterion
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {  // "test: undistortPoints.regression_14583"
          x = (u - cx) * ifx;
          y = (v - cy) * ify;
          break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
      }
    }
    // P = K, R = I: xx = fx x + 0 y + cx, ww = 1 / (0 x + 0 y + 1)
    const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
    xy_out[2 * i] = (float)(xx * ww);
    xy_out[2 * i + 1] = (float)(yy * ww);
  }
}

void undistort_keypoints(const std::vector<KeyPoint>& kps, const float K[4], const float* dist, int n_dist,
                         std::vector<KeyPoint>& out) {
  out = kps;
  if (n_dist <= 0 || dist[0] == 0.0f) return;  // src/Frame.cc:854-857
  std::vector<float> xy(2 * kps.size()), uo(2 * kps.size());
  for (size_t i = 0; i < kps.size(); i++) {
    xy[2 * i] = kps[i].x;
    xy[2 * i + 1] = kps[i].y;
  }
  undistort_points(xy.data(), (int)kps.size(), K, dist, n_dist, uo.data());
  for (size_t i = 0; i < kps.size(); i++) {
    out[i].x = uo[2 * i];
    out[i].y = uo[2 * i + 1];
  }
}

void compute_image_bounds(int cols, int rows, const float K[4], const float* dist, int n_dist, float bounds[4]) {
  if (n_dist > 0 && dist[0] != 0.0f) {
    const float c[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
    float o[8];
    undistort_points(c, 4, K, dist, n_dist, o);
    bounds[0] = std::min(o[0], o[4]);  // mnMinX
    bounds[2] = std::max(o[2], o[6]);  // mnMaxX
    bounds[1] = std::min(o[1], o[3]);  // mnMinY
    bounds[3] = std::max(o[5], o[7]);  // mnMaxY
  } else {
    bounds[0] = 0.f;
    bounds[1] = 0.f;
    bounds[2] = (float)cols;
    bounds[3] = (float)rows;
  }
}

// Frame::AssignFeaturesToGrid / PosInGrid, src/Frame.cc:520-547,833-844 (64x48 grid, round-to-cell).
void FrameGrid::build(const std::vector<KeyPoint>& kps, float minX_, float minY_, float maxX_, float maxY_) {
  minX = minX_; minY = minY_; maxX = maxX_; maxY = maxY_;
  invW = 64.f / (maxX - minX);  // static_cast<float>(FRAME_GRID_COLS) / (mnMaxX - mnMinX), Frame.cc:243
  invH = 48.f / (maxY - minY);
  cells.assign(64 * 48, {});
  for (int i = 0; i < (int)kps.size(); i++) {
    const int px = (int)std::round((kps[i].x - minX) * invW);
    const int py = (int)std::round((kps[i].y - minY) * invH);
    if (px < 0 || px >= 64 || py < 0 || py >= 48) continue;
    cells[px * 48 + py].push_back(i);
  }
}

// Frame::GetFeaturesInArea, src/Frame.cc:765-831.
std::vector<int> FrameGrid::features_in_area(const std::vector<KeyPoint>& kps, float x, float y, float r,
                                             int minLevel, int maxLevel) const {
  std::vector<int> out;
  const int cx0 = std::max(0, (int)std::floor((x - minX - r) * invW));
  if (cx0 >= 64) return out;
  const int cx1 = std::min(63, (int)std::ceil((x - minX + r) * invW));
  if (cx1 < 0) return out;
  const int cy0 = std::max(0, (int)std::floor((y - minY - r) * invH));
  if (cy0 >= 48) return out;
  const int cy1 = std::min(47, (int)std::ceil((y - minY + r) * invH));
  if (cy1 < 0) return out;
  const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = cx0; ix <= cx1; ix++)
    for (int iy = cy0; iy <= cy1; iy++)
      for (int idx : cells[ix * 48 + iy]) {
        const KeyPoint& kp = kps[idx];
        if (checkLevels) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float dx = kp.x - x, dy = kp.y - y;
        if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(idx);
      }
  return out;
}

// ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1920-1955.
static void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:618-764, executed in serial i1 order.
int search_for_initialization(const std::vector<KeyPoint>& k1, const uint8_t* d1,
                              const std::vector<KeyPoint>& k2, const uint8_t* d2, const FrameGrid& g2,
                              std::vector<float>& prev, std::vector<int>& matches12, int windowSize,
                              float nnratio, bool checkOri) {
  const int HISTO = 30, TH_LOW = 50;
  int nmatches = 0;
  matches12.assign(k1.size(), -1);
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  std::vector<int> matchedDist(k2.size(), INT_MAX), matches21(k2.size(), -1);
  for (size_t i1 = 0; i1 < k1.size(); i1++) {
    const KeyPoint& kp1 = k1[i1];
    if (kp1.octave > 0) continue;
    std::vector<int> cand = g2.features_in_area(k2, prev[2 * i1], prev[2 * i1 + 1], (float)windowSize,
                                                kp1.octave, kp1.octave);
    if (cand.empty()) continue;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int i2 : cand) {
      const int dist = descriptor_distance(d1 + i1 * 32, d2 + (size_t)i2 * 32);
      if (matchedDist[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW && bestDist < (float)bestDist2 * nnratio) {
      if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; nmatches--; }
      matches12[i1] = bestIdx2;
      matches21[bestIdx2] = (int)i1;
      matchedDist[bestIdx2] = bestDist;
      nmatches++;
      if (checkOri) {
        float rot = k1[i1].angle - k2[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO) bin = 0;
        rotHist[bin].push_back((int)i1);
      }
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int i = 0; i < HISTO; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (size_t i1 = 0; i1 < matches12.size(); i1++)
    if (matches12[i1] >= 0) { prev[2 * i1] = k2[matches12[i1]].x; prev[2 * i1 + 1] = k2[matches12[i1]].y; }
  return nmatches;
}

// ORBmatcher::SearchByProjection (local map points), src/ORBmatcher.cc:41-221, F.Nleft == -1 branch, executed in
// serial iMP order (the fork's tbb::parallel_for around this body is a data race on F.mvpMapPoints / nmatches).
int search_by_projection_map(const std::vector<KeyPoint>& kpsUn, const uint8_t* desc, const float* uRight,
                             const FrameGrid& grid, const std::vector<float>& scaleFactors,
                             const std::vector<MapPointView>& mps, float th, bool bFarPoints, float thFarPoints,
                             float nnratio, std::vector<uint8_t>& occupied, std::vector<int>& match) {
  const int TH_HIGH = 100;
  int nmatches = 0;
  match.assign(kpsUn.size(), -1);
  const bool bFactor = th != 1.0;
  for (size_t iMP = 0; iMP < mps.size(); iMP++) {
    const MapPointView& mp = mps[iMP];
    if (!mp.in_view) continue;
    if (bFarPoints && mp.track_depth > thFarPoints) continue;
    if (mp.bad) continue;
    const int level = mp.predicted_level;
    float r = mp.view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, :223-228
    if (bFactor) r *= th;
    const std::vector<int> cand =
        grid.features_in_area(kpsUn, mp.proj_x, mp.proj_y, r * scaleFactors[level], level - 1, level);
    if (cand.empty()) continue;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : cand) {
      if (occupied[idx]) continue;
      if (uRight && uRight[idx] > 0) {
        const float er = std::fabs(mp.proj_xr - uRight[idx]);
        if (er > r * scaleFactors[level]) continue;
      }
      const int dist = descriptor_distance(mp.desc, desc + (size_t)idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestLevel2 = bestLevel;
        bestLevel = kpsUn[idx].octave;
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = kpsUn[idx].octave;
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
        match[bestIdx] = (int)iMP;            // F.mvpMapPoints[bestIdx] = pMP
        occupied[bestIdx] = mp.has_observations;  // later points skip it only if pMP->Observations() > 0
        nmatches++;
      }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection (CurrentFrame <- LastFrame), src/ORBmatcher.cc:1614-1700 (Nleft == -1) and the
// rotation-consistency cull :1780-1800.
int search_by_projection_frame(const std::vector<KeyPoint>& kpsUn, const uint8_t* desc, const float* uRight,
                               const FrameGrid& grid, const std::vector<ProjectedPoint>& pts, bool checkOri,
                               std::vector<uint8_t>& occupied, std::vector<int>& match) {
  const int HISTO = 30, TH_HIGH = 100;
  int nmatches = 0;
  match.assign(kpsUn.size(), -1);
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  for (size_t i = 0; i < pts.size(); i++) {
    const ProjectedPoint& p = pts[i];
    if (!p.valid) continue;  // no MapPoint, outlier, behind the camera or outside the image (:1615-1636)
    const std::vector<int> cand = grid.features_in_area(kpsUn, p.u, p.v, p.radius, p.min_level, p.max_level);
    if (cand.empty()) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : cand) {
      if (occupied[i2]) continue;
      if (uRight && uRight[i2] > 0) {
        const float er = std::fabs(p.ur - uRight[i2]);
        if (er > p.radius) continue;
      }
      const int dist = descriptor_distance(p.desc, desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= TH_HIGH) {
      match[bestIdx2] = (int)i;
      occupied[bestIdx2] = p.has_observations;
      nmatches++;
      if (checkOri) {
        float rot = p.angle - kpsUn[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int b = 0; b < HISTO; b++) {
      if (b == ind1 || b == ind2 || b == ind3) continue;
      for (int idx : rotHist[b]) {
        match[idx] = -1;  // CurrentFrame.mvpMapPoints[idx] = NULL
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
// (relocalisation), src/ORBmatcher.cc:1808-1918: the loop over pKF's map points from the window search on (:1855-1916).
int search_by_projection_keyframe(const std::vector<KeyPoint>& kpsUn, const uint8_t* desc, const FrameGrid& grid,
                                  const std::vector<ProjectedPoint>& pts, int ORBdist, bool checkOri,
                                  std::vector<uint8_t>& occupied, std::vector<int>& match) {
  const int HISTO = 30;
  int nmatches = 0;
  // CurrentFrame.mvpMapPoints as seen by this routine: -1 = NULL, -2 = a pointer that was there before, i >= 0 = pKF's point i
  std::vector<int> mvpMapPoints(kpsUn.size());
  for (size_t i = 0; i < kpsUn.size(); i++) mvpMapPoints[i] = occupied[i] ? -2 : -1;
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  for (size_t i = 0; i < pts.size(); i++) {
    const ProjectedPoint& p = pts[i];
    if (!p.valid) continue;  // :1823-1845 on the caller's side
    const std::vector<int> vIndices2 = grid.features_in_area(kpsUn, p.u, p.v, p.radius, p.min_level, p.max_level);  // :1855
    if (vIndices2.empty()) continue;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (mvpMapPoints[i2] != -1) continue;  // :1871
      const int dist = descriptor_distance(p.desc, desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= ORBdist) {  // :1886
      mvpMapPoints[bestIdx2] = (int)i;
      nmatches++;
      if (checkOri) {
        float rot = p.angle - kpsUn[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int b = 0; b < HISTO; b++) {
      if (b == ind1 || b == ind2 || b == ind3) continue;
      for (int idx : rotHist[b]) {
        mvpMapPoints[idx] = -1;  // :1911
        nmatches--;
      }
    }
  }
  match.assign(kpsUn.size(), -1);
  for (size_t i = 0; i < kpsUn.size(); i++) {
    occupied[i] = mvpMapPoints[i] != -1;
    if (mvpMapPoints[i] >= 0) match[i] = mvpMapPoints[i];
  }
  return nmatches;
}

// Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:122-149) with F12 already formed (:130-133 is the caller's
// Eigen product); built without contraction like the reference's x86-64 baseline build.
static bool pinhole_epipolar_constrain(const KeyPoint& kp1, const KeyPoint& kp2, const float* F12, float unc) {
  const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
  const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
  const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
  const float num = a * kp2.x + b * kp2.y + c;
  const float den = a * a + b * b;
  if (den == 0) return false;
  const float dsqr = num * num / den;
  return dsqr < 3.84 * unc;
}

// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:886-1106) for single-camera key frames (mpCamera2 == NULL in both).
int search_for_triangulation(const std::vector<uint32_t>& nodes1, const std::vector<int>& start1, const std::vector<uint32_t>& feat1,
                             const std::vector<KeyPoint>& k1, const uint8_t* d1, const uint8_t* hasMP1, const float* uRight1,
                             const std::vector<uint32_t>& nodes2, const std::vector<int>& start2, const std::vector<uint32_t>& feat2,
                             const std::vector<KeyPoint>& k2, const uint8_t* d2, const uint8_t* hasMP2, const float* uRight2,
                             const std::vector<float>& scaleFactors2, const std::vector<float>& levelSigma2_2, const float ep[2],
                             const float F12[9], bool bOnlyStereo, bool bCoarse, bool checkOri, std::vector<int>& vMatches12) {
  const int HISTO = 30, TH_LOW = 50;
  int nmatches = 0;
  std::vector<bool> vbMatched2(k2.size(), false);  // never set by the reference either (:933): a keypoint of pKF2 can be taken twice
  vMatches12.assign(k1.size(), -1);
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  size_t f1 = 0, f2 = 0;
  while (f1 < nodes1.size() && f2 < nodes2.size()) {
    if (nodes1[f1] == nodes2[f2]) {
      for (int i1 = start1[f1]; i1 < start1[f1 + 1]; i1++) {
        const size_t idx1 = feat1[i1];
        if (hasMP1[idx1]) continue;  // :953
        const bool bStereo1 = uRight1 && uRight1[idx1] >= 0;
        if (bOnlyStereo && !bStereo1) continue;
        const KeyPoint& kp1 = k1[idx1];
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int i2 = start2[f2]; i2 < start2[f2 + 1]; i2++) {
          const size_t idx2 = feat2[i2];
          if (vbMatched2[idx2] || hasMP2[idx2]) continue;  // :976
          const bool bStereo2 = uRight2 && uRight2[idx2] >= 0;
          if (bOnlyStereo)
            if (!bStereo2) continue;
          const int dist = descriptor_distance(d1 + idx1 * 32, d2 + idx2 * 32);
          if (dist > TH_LOW || dist > bestDist) continue;  // ties replace: the LAST of the best candidates wins
          const KeyPoint& kp2 = k2[idx2];
          if (!bStereo1 && !bStereo2) {  // && !pKF1->mpCamera2
            const float distex = ep[0] - kp2.x;
            const float distey = ep[1] - kp2.y;
            if (distex * distex + distey * distey < 100 * scaleFactors2[kp2.octave]) continue;
          }
          if (bCoarse || pinhole_epipolar_constrain(kp1, kp2, F12, levelSigma2_2[kp2.octave])) {
            bestIdx2 = (int)idx2;
            bestDist = dist;
          }
        }
        if (bestIdx2 >= 0) {
          const KeyPoint& kp2 = k2[bestIdx2];
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (checkOri) {
            float rot = kp1.angle - kp2.angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO) bin = 0;
            rotHist[bin].push_back((int)idx1);
          }
        }
      }
      f1++;
      f2++;
    } else if (nodes1[f1] < nodes2[f2]) {
      f1 = std::lower_bound(nodes1.begin(), nodes1.end(), nodes2[f2]) - nodes1.begin();
    } else {
      f2 = std::lower_bound(nodes2.begin(), nodes2.end(), nodes1[f1]) - nodes2.begin();
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int b = 0; b < HISTO; b++) {
      if (b == ind1 || b == ind2 || b == ind3) continue;
      for (int idx : rotHist[b]) {
        vMatches12[idx] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th, bRight) (src/ORBmatcher.cc:1108-1277): the search part
// of the loop body (:1195-1256) -- KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:705-749: the frame grid without a level filter),
// the level window, the chi-square gate on the reprojection error and the first strict minimum of the descriptor distance.
int fuse_search(const std::vector<KeyPoint>& kps, const uint8_t* desc, const float* uRight, const FrameGrid& grid,
                const std::vector<float>& invLevelSigma2, const std::vector<FusePoint>& pts, int maxDist,
                std::vector<int>& bestIdxOut, std::vector<int>& bestDistOut) {
  int nFused = 0;
  bestIdxOut.assign(pts.size(), -1);
  bestDistOut.assign(pts.size(), 256);
  for (size_t i = 0; i < pts.size(); i++) {
    const FusePoint& p = pts[i];
    if (!p.valid) continue;  // :1141-1192 on the caller's side
    const int nPredictedLevel = p.predicted_level;
    const std::vector<int> vIndices = grid.features_in_area(kps, p.u, p.v, p.radius, -1, -1);
    if (vIndices.empty()) continue;
    int bestDist = 256, bestIdx = -1;
    for (int idx : vIndices) {
      const KeyPoint& kp = kps[idx];
      const int kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (uRight && uRight[idx] >= 0) {
        const float ex = p.u - kp.x;
        const float ey = p.v - kp.y;
        const float er = p.ur - uRight[idx];
        const float e2 = ex * ex + ey * ey + er * er;
        if (e2 * invLevelSigma2[kpLevel] > 7.8) continue;
      } else {
        const float ex = p.u - kp.x;
        const float ey = p.v - kp.y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
      }
      const int dist = descriptor_distance(p.desc, desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    bestDistOut[i] = bestDist;
    if (bestDist <= maxDist) {  // TH_LOW (Fuse, :1258,1357), TH_HIGH (SearchBySim3, :1494,1564)
      bestIdxOut[i] = bestIdx;
      nFused++;
    }
  }
  return nFused;
}

// ORBmatcher::SearchForTriangulation, the two-camera-rig branch (src/ORBmatcher.cc:906-923, 1007-1064).
int search_for_triangulation_rig(const std::vector<uint32_t>& nodes1, const std::vector<int>& start1, const std::vector<uint32_t>& feat1,
                                 const std::vector<KeyPoint>& k1, const uint8_t* d1, const uint8_t* hasMP1, int nLeft1,
                                 const std::vector<uint32_t>& nodes2, const std::vector<int>& start2, const std::vector<uint32_t>& feat2,
                                 const std::vector<KeyPoint>& k2, const uint8_t* d2, const uint8_t* hasMP2, int nLeft2,
                                 const std::vector<float>& levelSigma2_1, const std::vector<float>& levelSigma2_2, const TriRig& rig,
                                 bool bOnlyStereo, bool bCoarse, bool checkOri, std::vector<int>& vMatches12,
                                 std::vector<uint8_t>* borderline) {
  const int HISTO = 30, TH_LOW = 50;
  int nmatches = 0;
  vMatches12.assign(k1.size(), -1);
  if (borderline) borderline->assign(k1.size(), 0);
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  KB8 cam[4];
  for (int c = 0; c < 4; c++) {
    for (int i = 0; i < 8; i++) cam[c].p[i] = rig.cam[c][i];
    cam[c].precision = rig.precision;
  }
  size_t f1 = 0, f2 = 0;
  while (f1 < nodes1.size() && f2 < nodes2.size()) {
    if (nodes1[f1] == nodes2[f2]) {
      for (int i1 = start1[f1]; i1 < start1[f1 + 1]; i1++) {
        const size_t idx1 = feat1[i1];
        if (hasMP1[idx1]) continue;
        // bStereo1 = (!pKF1->mpCamera2 && ...) is false for a rig: bOnlyStereo skips every feature (:957-959)
        if (bOnlyStereo) continue;
        const KeyPoint& kp1 = k1[idx1];
        const bool bRight1 = (int)idx1 >= nLeft1;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int i2 = start2[f2]; i2 < start2[f2 + 1]; i2++) {
          const size_t idx2 = feat2[i2];
          if (hasMP2[idx2]) continue;
          const int dist = descriptor_distance(d1 + idx1 * 32, d2 + idx2 * 32);
          if (dist > TH_LOW || dist > bestDist) continue;
          const KeyPoint& kp2 = k2[idx2];
          const bool bRight2 = (int)idx2 >= nLeft2;
          // (no epipole gate: `!pKF1->mpCamera2` is false, :997)
          const int sel = (bRight1 ? 2 : 0) + (bRight2 ? 1 : 0);  // ll, lr, rl, rr (:1008-1041)
          bool ok = bCoarse;
          if (!ok) {
            float P[3], gate[5];
            const float z = kb8_triangulate_matches(cam[bRight1 ? 1 : 0], cam[bRight2 ? 3 : 2], kp1.x, kp1.y, kp2.x, kp2.y, rig.R[sel],
                                                    rig.t[sel], levelSigma2_1[kp1.octave], levelSigma2_2[kp2.octave], P, gate);
            ok = z > 0.0001f;
            if (borderline) {
              bool near = std::fabs(gate[0] - 0.9998f) < 1e-5f;
              if (!std::isnan(gate[1])) near = near || std::fabs(gate[1]) < 1e-3f;
              if (!std::isnan(gate[2])) near = near || std::fabs(gate[2]) < 1e-3f;
              if (!std::isnan(gate[3])) near = near || std::fabs(gate[3] - 1.0f) < 1e-2f;
              if (!std::isnan(gate[4])) near = near || std::fabs(gate[4] - 1.0f) < 1e-2f;
              if (near) (*borderline)[idx1] = 1;
            }
          }
          if (ok) {
            bestIdx2 = (int)idx2;
            bestDist = dist;
          }
        }
        if (bestIdx2 >= 0) {
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (checkOri) {
            float rot = kp1.angle - k2[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO) bin = 0;
            rotHist[bin].push_back((int)idx1);
          }
        }
      }
      f1++;
      f2++;
    } else if (nodes1[f1] < nodes2[f2]) {
      f1 = std::lower_bound(nodes1.begin(), nodes1.end(), nodes2[f2]) - nodes1.begin();
    } else {
      f2 = std::lower_bound(nodes2.begin(), nodes2.end(), nodes1[f1]) - nodes2.begin();
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int b = 0; b < HISTO; b++) {
      if (b == ind1 || b == ind2 || b == ind3) continue;
      for (int idx : rotHist[b]) {
        vMatches12[idx] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), src/ORBmatcher.cc:766-884.
int search_by_bow_keyframes(const std::vector<uint32_t>& nodes1, const std::vector<int>& start1, const std::vector<uint32_t>& feat1,
                            const uint8_t* d1, const float* angle1, const uint8_t* valid1, int n1, const std::vector<uint32_t>& nodes2,
                            const std::vector<int>& start2, const std::vector<uint32_t>& feat2, const uint8_t* d2, const float* angle2,
                            const uint8_t* valid2, int n2, float nnratio, bool checkOri, std::vector<int>& vMatches12) {
  const int HISTO = 30, TH_LOW = 50;
  vMatches12.assign(n1, -1);
  std::vector<bool> vbMatched2(n2, false);
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  int nmatches = 0;
  size_t f1 = 0, f2 = 0;
  while (f1 < nodes1.size() && f2 < nodes2.size()) {
    if (nodes1[f1] == nodes2[f2]) {
      for (int i1 = start1[f1]; i1 < start1[f1 + 1]; i1++) {
        const size_t idx1 = feat1[i1];
        if (!valid1[idx1]) continue;  // :799-806
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int i2 = start2[f2]; i2 < start2[f2 + 1]; i2++) {
          const size_t idx2 = feat2[i2];
          if (vbMatched2[idx2] || !valid2[idx2]) continue;  // :817-825
          const int dist = descriptor_distance(d1 + idx1 * 32, d2 + idx2 * 32);
          if (dist < bestDist1) {
            bestDist2 = bestDist1;
            bestDist1 = dist;
            bestIdx2 = (int)idx2;
          } else if (dist < bestDist2) {
            bestDist2 = dist;
          }
        }
        if (bestDist1 < TH_LOW) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            vMatches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = true;
            if (checkOri) {
              float rot = angle1[idx1] - angle2[bestIdx2];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)std::round(rot * factor);
              if (bin == HISTO) bin = 0;
              rotHist[bin].push_back((int)idx1);
            }
            nmatches++;
          }
        }
      }
      f1++;
      f2++;
    } else if (nodes1[f1] < nodes2[f2]) {
      f1 = std::lower_bound(nodes1.begin(), nodes1.end(), nodes2[f2]) - nodes1.begin();
    } else {
      f2 = std::lower_bound(nodes2.begin(), nodes2.end(), nodes1[f1]) - nodes2.begin();
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int b = 0; b < HISTO; b++) {
      if (b == ind1 || b == ind2 || b == ind3) continue;
      for (int idx : rotHist[b]) {
        vMatches12[idx] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ---- stereo-fisheye branches ------------------------------------------------------------------------------------------------
int search_by_projection_map_fisheye(const std::vector<KeyPoint>& kps, const uint8_t* desc, int nLeft, const FrameGrid& gridL,
                                     const FrameGrid& gridR, const std::vector<float>& scaleFactors,
                                     const std::vector<MapPointView>& mps, const std::vector<MapPointRight>& mpsR, float th,
                                     bool bFarPoints, float thFarPoints, float nnratio, const std::vector<int>& leftToRight,
                                     const std::vector<int>& rightToLeft, std::vector<uint8_t>& occupied,
                                     std::vector<int>& match) {
  const int TH_HIGH = 100;
  int nmatches = 0;
  match.assign(kps.size(), -1);
  const std::vector<KeyPoint> kL(kps.begin(), kps.begin() + nLeft), kR(kps.begin() + nLeft, kps.end());
  const bool bFactor = th != 1.0;
  auto assign = [&](int slot, size_t iMP) {  // F.mvpMapPoints[slot] = pMP
    match[slot] = (int)iMP;
    occupied[slot] = mps[iMP].has_observations;
  };
  for (size_t iMP = 0; iMP < mps.size(); iMP++) {
    const MapPointView& mp = mps[iMP];
    const MapPointRight& mr = mpsR[iMP];
    if (!mp.in_view && !mr.in_view_r) continue;               // :54
    if (bFarPoints && mp.track_depth > thFarPoints) continue;  // :56
    if (mp.bad) continue;                                      // :58
    if (mp.in_view) {                                          // :60-138
      const int level = mp.predicted_level;
      float r = mp.view_cos > 0.998 ? 2.5f : 4.0f;
      if (bFactor) r *= th;
      const std::vector<int> cand = gridL.features_in_area(kL, mp.proj_x, mp.proj_y, r * scaleFactors[level], level - 1, level);
      if (!cand.empty()) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : cand) {
          if (occupied[idx]) continue;  // (no mvuRight gate: F.Nleft != -1, :90)
          const int dist = descriptor_distance(mp.desc, desc + (size_t)idx * 32);
          if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestLevel2 = bestLevel;
            bestLevel = kL[idx].octave;
            bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = kL[idx].octave;
            bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;  // skips the right camera too, :120
          if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
            assign(bestIdx, iMP);
            if (leftToRight[bestIdx] != -1) {  // also the stereo observation in the right camera, :126-132
              assign(leftToRight[bestIdx] + nLeft, iMP);
              nmatches++;
            }
            nmatches++;
          }
        }
      }
    }
    if (mr.in_view_r) {  // :141-213
      const int level = mr.predicted_level_r;
      if (level != -1) {
        const float r = mr.view_cos_r > 0.998 ? 2.5f : 4.0f;  // not scaled by th, :144
        const std::vector<int> cand = gridR.features_in_area(kR, mp.proj_xr, mr.proj_yr, r * scaleFactors[level], level - 1, level);
        if (cand.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : cand) {
          if (occupied[idx + nLeft]) continue;
          const int dist = descriptor_distance(mp.desc, desc + (size_t)(idx + nLeft) * 32);
          if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestLevel2 = bestLevel;
            bestLevel = kR[idx].octave;
            bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = kR[idx].octave;
            bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
          if (rightToLeft[bestIdx] != -1) {  // :199-204
            assign(rightToLeft[bestIdx], iMP);
            nmatches++;
          }
          assign(bestIdx + nLeft, iMP);
          nmatches++;
        }
      }
    }
  }
  return nmatches;
}

int search_by_projection_frame_fisheye(const std::vector<KeyPoint>& kps, const uint8_t* desc, int nLeft, const FrameGrid& gridL,
                                       const FrameGrid& gridR, const std::vector<ProjectedPoint>& pts, const float* uvRight,
                                       bool checkOri, std::vector<uint8_t>& occupied, std::vector<int>& match) {
  const int HISTO = 30, TH_HIGH = 100;
  int nmatches = 0;
  match.assign(kps.size(), -1);
  const std::vector<KeyPoint> kL(kps.begin(), kps.begin() + nLeft), kR(kps.begin() + nLeft, kps.end());
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  auto vote = [&](float angleLF, float angleCF, int slot) {
    float rot = angleLF - angleCF;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO) bin = 0;
    rotHist[bin].push_back(slot);
  };
  for (size_t i = 0; i < pts.size(); i++) {
    const ProjectedPoint& p = pts[i];
    if (!p.valid) continue;
    {
      const std::vector<int> cand = gridL.features_in_area(kL, p.u, p.v, p.radius, p.min_level, p.max_level);
      if (cand.empty()) continue;  // skips the right camera as well, :1651
      int bestDist = 256, bestIdx2 = -1;
      for (int i2 : cand) {
        if (occupied[i2]) continue;  // (no mvuRight gate: CurrentFrame.Nleft != -1, :1667)
        const int dist = descriptor_distance(p.desc, desc + (size_t)i2 * 32);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
      if (bestDist <= TH_HIGH) {
        match[bestIdx2] = (int)i;
        occupied[bestIdx2] = p.has_observations;
        nmatches++;
        if (checkOri) vote(p.angle, kL[bestIdx2].angle, bestIdx2);
      }
    }
    {  // :1703-1775
      const std::vector<int> cand = gridR.features_in_area(kR, uvRight[2 * i], uvRight[2 * i + 1], p.radius, p.min_level, p.max_level);
      int bestDist = 256, bestIdx2 = -1;
      for (int i2 : cand) {
        if (occupied[i2 + nLeft]) continue;
        const int dist = descriptor_distance(p.desc, desc + (size_t)(i2 + nLeft) * 32);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
      }
      if (bestDist <= TH_HIGH) {
        match[bestIdx2 + nLeft] = (int)i;
        occupied[bestIdx2 + nLeft] = p.has_observations;
        nmatches++;
        if (checkOri) vote(p.angle, kR[bestIdx2].angle, bestIdx2 + nLeft);
      }
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int b = 0; b < HISTO; b++) {
      if (b == ind1 || b == ind2 || b == ind3) continue;
      for (int idx : rotHist[b]) {
        match[idx] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ======================================================================================= f4: bag of words (DBoW2)
void Vocabulary::build(int k_, int L_, int scoring_, int weighting_, int n, const int* parent_, const uint8_t* isLeaf,
                       const uint8_t* desc_, const double* weight_) {
  k = k_; L = L_; scoring = scoring_; weighting = weighting_;
  parent.assign(parent_, parent_ + n);
  desc.assign(desc_, desc_ + (size_t)n * 32);
  weight.assign(weight_, weight_ + n);
  wordId.assign(n, -1);
  std::vector<int> cnt(n + 1, 0);
  for (int i = 1; i < n; i++) cnt[parent[i] + 1]++;
  childStart.assign(n + 1, 0);
  for (int i = 0; i < n; i++) childStart[i + 1] = childStart[i] + cnt[i + 1];
  children.assign(n > 0 ? n - 1 : 0, 0);
  std::vector<int> fill(childStart.begin(), childStart.end() - 1);
  nWords = 0;
  for (int i = 1; i < n; i++) {  // file order: children.push_back(nid), words numbered as their leaves appear (:1389-1414)
    children[fill[parent[i]]++] = i;
    if (isLeaf[i]) wordId[i] = nWords++;
  }
}

bool Vocabulary::load_text(const char* path) {
  FILE* f = std::fopen(path, "r");
  if (!f) return false;
  int k_, L_, n1, n2;
  if (std::fscanf(f, "%d %d %d %d", &k_, &L_, &n1, &n2) != 4 || k_ < 0 || k_ > 20 || L_ < 1 || L_ > 10 || n1 < 0 || n1 > 5 ||
      n2 < 0 || n2 > 3) {
    std::fclose(f);
    return false;
  }
  std::vector<int> par(1, 0);
  std::vector<uint8_t> leaf(1, 0), d(32, 0);
  std::vector<double> w(1, 0.0);
  for (;;) {
    int pid, isLeaf;
    if (std::fscanf(f, "%d %d", &pid, &isLeaf) != 2) break;
    uint8_t row[32];
    bool ok = true;
    for (int i = 0; i < 32 && ok; i++) {
      int b;
      ok = std::fscanf(f, "%d", &b) == 1;
      row[i] = (uint8_t)b;
    }
    double wt;
    if (!ok || std::fscanf(f, "%lf", &wt) != 1) break;
    par.push_back(pid);
    leaf.push_back(isLeaf > 0);
    d.insert(d.end(), row, row + 32);
    w.push_back(wt);
  }
  std::fclose(f);
  build(k_, L_, n1, n2, (int)par.size(), par.data(), leaf.data(), d.data(), w.data());
  return true;
}

bool Vocabulary::save_text(const char* path) const {
  FILE* f = std::fopen(path, "w");
  if (!f) return false;
  std::fprintf(f, "%d %d  %d %d\n", k, L, scoring, weighting);
  for (size_t i = 1; i < parent.size(); i++) {
    std::fprintf(f, "%d %d ", parent[i], is_leaf((int)i) ? 1 : 0);
    for (int b = 0; b < 32; b++) std::fprintf(f, "%d ", desc[i * 32 + b]);
    std::fprintf(f, "%.17g\n", weight[i]);
  }
  std::fclose(f);
  return true;
}

void bow_transform_one(const Vocabulary& v, const uint8_t* d, int levelsup, int& wordId, double& weight, int& nodeId) {
  const int nidLevel = v.L - levelsup;
  nodeId = 0;
  bool nidSet = nidLevel <= 0;
  int cur = 0, level = 0;
  do {
    ++level;
    const int c0 = v.childStart[cur], c1 = v.childStart[cur + 1];
    int best = v.children[c0];
    int bestD = descriptor_distance(d, &v.desc[(size_t)best * 32]);
    for (int c = c0 + 1; c < c1; c++) {
      const int id = v.children[c];
      const int dist = descriptor_distance(d, &v.desc[(size_t)id * 32]);
      if (dist < bestD) { bestD = dist; best = id; }
    }
    cur = best;
    if (level == nidLevel) { nodeId = cur; nidSet = true; }
  } while (!v.is_leaf(cur));
  if (!nidSet) nodeId = cur;
  wordId = v.wordId[cur];
  weight = v.weight[cur];
}

void bow_transform(const Vocabulary& v, const uint8_t* desc, int n, int levelsup, std::vector<uint32_t>& words,
                   std::vector<double>& values, std::vector<uint32_t>& nodes, std::vector<int>& nodeStart,
                   std::vector<uint32_t>& features) {
  words.clear(); values.clear(); nodes.clear(); nodeStart.assign(1, 0); features.clear();
  if (v.parent.size() <= 1) return;  // empty()
  // mustNormalize: every scoring but DOT_PRODUCT; L2 norm only for L2_NORM (ScoringObject.h:73-90)
  const bool must = v.scoring != 5;
  const bool l2 = v.scoring == 1;
  std::map<uint32_t, double> bow;
  std::map<uint32_t, std::vector<uint32_t>> fv;
  const bool additive = v.weighting == 0 || v.weighting == 1;  // TF_IDF, TF
  for (int i = 0; i < n; i++) {
    int wid, nid;
    double w;
    bow_transform_one(v, desc + (size_t)i * 32, levelsup, wid, w, nid);
    if (w > 0) {
      auto it = bow.lower_bound((uint32_t)wid);
      if (it != bow.end() && it->first == (uint32_t)wid) {
        if (additive) it->second += w;  // addWeight; addIfNotExist keeps the first
      } else {
        bow.insert(it, std::make_pair((uint32_t)wid, w));
      }
      fv[(uint32_t)nid].push_back((uint32_t)i);
    }
  }
  if (additive && !bow.empty() && !must) {
    const double nd = (double)bow.size();
    for (auto& e : bow) e.second /= nd;
  }
  if (must) {
    double norm = 0.0;
    if (!l2) {
      for (auto& e : bow) norm += std::fabs(e.second);
    } else {
      for (auto& e : bow) norm += e.second * e.second;
      norm = std::sqrt(norm);
    }
    if (norm > 0.0)
      for (auto& e : bow) e.second /= norm;
  }
  for (auto& e : bow) { words.push_back(e.first); values.push_back(e.second); }
  for (auto& e : fv) {
    nodes.push_back(e.first);
    features.insert(features.end(), e.second.begin(), e.second.end());
    nodeStart.push_back((int)features.size());
  }
}

int search_by_bow(const std::vector<uint32_t>& kfNodes, const std::vector<int>& kfStart, const std::vector<uint32_t>& kfFeat,
                  const uint8_t* kfDesc, const float* kfAngle, const uint8_t* kfValid, const std::vector<uint32_t>& fNodes,
                  const std::vector<int>& fStart, const std::vector<uint32_t>& fFeat, const uint8_t* fDesc, const float* fAngle,
                  int nF, int nLeftF, float nnratio, bool checkOri, std::vector<int>& match) {
  const int HISTO = 30, TH_LOW = 50;
  match.assign(nF, -1);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO];
  const float factor = 1.0f / HISTO;
  auto vote = [&](int iKF, int iF) {
    float rot = kfAngle[iKF] - fAngle[iF];
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * factor);
    if (bin == HISTO) bin = 0;
    rotHist[bin].push_back(iF);
  };
  size_t a = 0, b = 0;
  while (a < kfNodes.size() && b < fNodes.size()) {
    if (kfNodes[a] == fNodes[b]) {
      for (int p = kfStart[a]; p < kfStart[a + 1]; p++) {
        const int iKF = (int)kfFeat[p];
        if (!kfValid[iKF]) continue;  // !pMP || pMP->isBad()
        const uint8_t* dKF = kfDesc + (size_t)iKF * 32;
        int best1 = 256, bestIdx = -1, best2 = 256, best1R = 256, bestIdxR = -1, best2R = 256;
        for (int q = fStart[b]; q < fStart[b + 1]; q++) {
          const int iF = (int)fFeat[q];
          if (match[iF] >= 0) continue;
          const int dist = descriptor_distance(dKF, fDesc + (size_t)iF * 32);
          if (nLeftF == -1 || iF < nLeftF) {
            if (dist < best1) { best2 = best1; best1 = dist; bestIdx = iF; }
            else if (dist < best2) { best2 = dist; }
          } else {
            if (dist < best1R) { best2R = best1R; best1R = dist; bestIdxR = iF; }
            else if (dist < best2R) { best2R = dist; }
          }
        }
        if (best1 <= TH_LOW) {
          if ((float)best1 < nnratio * (float)best2) {
            match[bestIdx] = iKF;
            if (checkOri) vote(iKF, bestIdx);
            nmatches++;
          }
          if (best1R <= TH_LOW) {  // ":363-365": the ratio test of the right eye is short-circuited by "|| true"
            match[bestIdxR] = iKF;
            if (checkOri) vote(iKF, bestIdxR);
            nmatches++;
          }
        }
      }
      a++;
      b++;
    } else if (kfNodes[a] < fNodes[b]) {
      a = std::lower_bound(kfNodes.begin(), kfNodes.end(), fNodes[b]) - kfNodes.begin();
    } else {
      b = std::lower_bound(fNodes.begin(), fNodes.end(), kfNodes[a]) - fNodes.begin();
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO, ind1, ind2, ind3);
    for (int i = 0; i < HISTO; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx : rotHist[i]) { match[idx] = -1; nmatches--; }
    }
  }
  return nmatches;
}


// ---- Frame::isInFrustum + MapPoint::PredictScale (test infrastructure for orbx_project_map_points_batch) ----------------------
MapPointView is_in_frustum_kb8(const FramePoseKB8& T, const float P[3], const float Pn[3], float minDistance, float maxDistance,
                               float minX, float minY, float maxX, float maxY, float viewingCosLimit, float logScaleFactor,
                               int nlevels, double margin[2]) {
  MapPointView v{};
  double gate = 1e30, frac = 1e30;
  auto near = [&](float value, float threshold) {
    const double d = std::fabs((double)value - (double)threshold) / std::max(1.0, std::fabs((double)threshold));
    gate = std::min(gate, d);
  };
  auto done = [&]() {
    if (margin) { margin[0] = gate; margin[1] = frac; }
    return v;
  };
  // Pc = mR * P + mt   (src/Frame.cc:1354)
  const float X = ((T.R[0] * P[0] + T.R[1] * P[1]) + T.R[2] * P[2]) + T.t[0];
  const float Y = ((T.R[3] * P[0] + T.R[4] * P[1]) + T.R[5] * P[2]) + T.t[1];
  const float Z = ((T.R[6] * P[0] + T.R[7] * P[1]) + T.R[8] * P[2]) + T.t[2];
  const float pcDist = std::sqrt((X * X + Y * Y) + Z * Z);
  near(Z, 0.f);
  if (Z < 0.0f) return done();   // :1359
  KB8 cam{};
  for (int i = 0; i < 8; i++) cam.p[i] = T.kb8[i];
  const float Pc[3] = {X, Y, Z};
  float uv[2];
  kb8_project(cam, Pc, uv);      // :1363-1366
  near(uv[0], minX); near(uv[0], maxX);
  if (uv[0] < minX || uv[0] > maxX) return done();
  near(uv[1], minY); near(uv[1], maxY);
  if (uv[1] < minY || uv[1] > maxY) return done();
  const float maxD = 1.2f * maxDistance, minD = 0.8f * minDistance;
  const float ox = P[0] - T.Ow[0], oy = P[1] - T.Ow[1], oz = P[2] - T.Ow[2];   // PO = P - twc
  const float dist = std::sqrt((ox * ox + oy * oy) + oz * oz);
  near(dist, minD); near(dist, maxD);
  if (dist < minD || dist > maxD) return done();
  const float viewCos = ((ox * Pn[0] + oy * Pn[1]) + oz * Pn[2]) / dist;
  near(viewCos, viewingCosLimit);
  if (viewCos < viewingCosLimit) return done();
  const float ratio = maxDistance / dist;
  const float q = std::log(ratio) / logScaleFactor;
  frac = std::fabs((double)q - std::nearbyint((double)q));
  int nScale = (int)std::ceil(q);
  if (nScale < 0) nScale = 0;
  else if (nScale >= nlevels) nScale = nlevels - 1;
  v.in_view = 1;
  v.proj_x = uv[0];
  v.proj_y = uv[1];
  v.track_depth = pcDist;
  v.view_cos = viewCos;
  v.predicted_level = nScale;
  return done();
}

ProjectedPoint project_last_frame_point(const FramePoseQ& T, const float Pw[3], int lastOctave, float lastAngle, float th,
                                        const std::vector<float>& scaleFactors, float minX, float minY, float maxX, float maxY,
                                        double* margin) {
  ProjectedPoint o{};
  double gate = 1e30;
  auto near = [&](float value, float threshold) {
    const double d = std::fabs((double)value - (double)threshold) / std::max(1.0, std::fabs((double)threshold));
    gate = std::min(gate, d);
  };
  auto done = [&]() {
    if (margin) *margin = gate;
    return o;
  };
  o.angle = lastAngle;
  const float qx = T.q[0], qy = T.q[1], qz = T.q[2], qw = T.q[3];
  const float px = Pw[0], py = Pw[1], pz = Pw[2];
  // Thirdparty/Sophus/sophus/so3.hpp:363-365
  float ux = qy * pz - qz * py, uy = qz * px - qx * pz, uz = qx * py - qy * px;   // uv = q.vec().cross(p)
  ux += ux; uy += uy; uz += uz;                                                    // uv += uv
  const float cxv = qy * uz - qz * uy, cyv = qz * ux - qx * uz, czv = qx * uy - qy * ux;   // q.vec().cross(uv)
  const float rx = (px + qw * ux) + cxv, ry = (py + qw * uy) + cyv, rz = (pz + qw * uz) + czv;
  const float xc = rx + T.t[0], yc = ry + T.t[1], zc = rz + T.t[2];               // se3.hpp:323: so3() * p + translation()
  const float invzc = (float)(1.0 / (double)zc);                                  // src/ORBmatcher.cc:1626
  near(zc, 0.f);
  if (invzc < 0) return done();                                                   // :1628
  const float u = T.fx * xc / zc + T.cx, v = T.fy * yc / zc + T.cy;               // src/CameraModels/Pinhole.cpp:46-52
  if (!(u == u) || !(v == v)) return done();   // (0 / 0: the reference's comparisons all fail and it goes on with NaN; not modelled)
  near(u, minX); near(u, maxX);
  if (u < minX || u > maxX) return done();                                        // :1632-1633
  near(v, minY); near(v, maxY);
  if (v < minY || v > maxY) return done();                                        // :1634-1635
  const int nLevels = (int)scaleFactors.size();
  const int oct = std::min(std::max(lastOctave, 0), nLevels - 1);
  o.radius = th * scaleFactors[oct];                                              // :1643
  if (T.direction == 1) { o.min_level = oct; o.max_level = -1; }                  // :1647-1649 GetFeaturesInArea(u, v, r, nLastOctave)
  else if (T.direction == 2) { o.min_level = 0; o.max_level = oct; }              // :1650-1652
  else { o.min_level = oct - 1; o.max_level = oct + 1; }                          // :1653-1655
  o.u = u;
  o.v = v;
  o.ur = u - T.bf * invzc;                                                        // :1669
  o.valid = 1;
  return done();
}

MapPointView is_in_frustum(const FramePose& T, const float P[3], const float Pn[3], float minDistance, float maxDistance,
                           float minX, float minY, float maxX, float maxY, float viewingCosLimit, float logScaleFactor,
                           int nlevels, double margin[2]) {
  MapPointView v{};
  v.proj_x = -1.f;   // src/Frame.cc:635-636
  v.proj_y = -1.f;
  double gate = 1e30, frac = 1e30;
  auto near = [&](float value, float threshold) {
    const double d = std::fabs((double)value - (double)threshold) / std::max(1.0, std::fabs((double)threshold));
    gate = std::min(gate, d);
  };
  auto done = [&]() {
    if (margin) { margin[0] = gate; margin[1] = frac; }
    return v;
  };
  // Pc = mRcw * P + mtcw   (:642)
  const float X = ((T.Rcw[0] * P[0] + T.Rcw[1] * P[1]) + T.Rcw[2] * P[2]) + T.tcw[0];
  const float Y = ((T.Rcw[3] * P[0] + T.Rcw[4] * P[1]) + T.Rcw[5] * P[2]) + T.tcw[1];
  const float Z = ((T.Rcw[6] * P[0] + T.Rcw[7] * P[1]) + T.Rcw[8] * P[2]) + T.tcw[2];
  const float pcDist = std::sqrt((X * X + Y * Y) + Z * Z);   // Pc.norm()
  const float invz = 1.0f / Z;
  near(Z, 0.f);
  if (Z < 0.0f) return done();   // :648
  const float u = T.fx * X / Z + T.cx, vv = T.fy * Y / Z + T.cy;   // Pinhole::project (src/CameraModels/Pinhole.cpp:46-52)
  near(u, minX); near(u, maxX);
  if (u < minX || u > maxX) return done();
  near(vv, minY); near(vv, maxY);
  if (vv < minY || vv > maxY) return done();
  const float maxD = 1.2f * maxDistance, minD = 0.8f * minDistance;   // Get{Max,Min}DistanceInvariance (src/MapPoint.cc:531-541)
  const float ox = P[0] - T.Ow[0], oy = P[1] - T.Ow[1], oz = P[2] - T.Ow[2];
  const float dist = std::sqrt((ox * ox + oy * oy) + oz * oz);
  near(dist, minD); near(dist, maxD);
  if (dist < minD || dist > maxD) return done();
  const float viewCos = ((ox * Pn[0] + oy * Pn[1]) + oz * Pn[2]) / dist;
  near(viewCos, viewingCosLimit);
  if (viewCos < viewingCosLimit) return done();
  const float ratio = maxDistance / dist;   // PredictScale: mfMaxDistance / currentDist
  const float q = std::log(ratio) / logScaleFactor;
  frac = std::fabs((double)q - std::nearbyint((double)q));
  int nScale = (int)std::ceil(q);
  if (nScale < 0) nScale = 0;
  else if (nScale >= nlevels) nScale = nlevels - 1;
  v.in_view = 1;
  v.proj_x = u;
  v.proj_xr = u - T.bf * invz;
  v.track_depth = pcDist;
  v.proj_y = vv;
  v.predicted_level = nScale;
  v.view_cos = viewCos;
  return done();
}

}  // namespace orbo
