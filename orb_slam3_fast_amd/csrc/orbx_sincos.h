// orbx_sincos.h — glibc's sinf / cosf, bit for bit, for the descriptor steering of computeOrbDescriptor.
//
// The reference evaluates `(float)cos(angle), (float)sin(angle)` on a float under `using namespace std`
// (src/ORBextractor.cc:66-67,106-107), i.e. libm cosf / sinf.  glibc >= 2.28 computes both in double
// (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, s_sincosf_data.c): for |y| < pi/4 the polynomial directly,
// otherwise n = round(y * 2/pi) through a scaled double -> int32 conversion, r = y - n * pi/2, then the degree-7 sine or
// degree-8 cosine polynomial picked by the quadrant, rounded once to float.  x86-64 glibc carries two ifunc variants of
// that one C source: __sinf_fma / __cosf_fma (every a + b*c contracted; chosen on CPUs with FMA + AVX2) and
// __sinf_sse2 / __cosf_sse2 (separate multiply and add).  Both are restated here (FUSED = true / false) in the operation
// order of the glibc 2.35 objects (libm-2.35.a: s_sinf-fma.o, s_sinf-sse2.o, s_cosf-*.o, s_sincosf-fma.o).  Exhaustive
// sweep (tools/sincos_sweep.cpp): over EVERY float in [0, 2 pi] — 1.09e9 arguments, all the path can produce — both
// variants return the same floats, and both equal the host libm's sinf / cosf; the double results differ in their
// last bits but never across a float rounding boundary on this domain.  So the descriptors equal those of an x86-64
// glibc build of the reference whichever variant its loader picked; the kernels run the FMA form.
//
// Valid for |y| < 120 (the path produces y = fastAtan2(..) * (pi/180) in [0, 2 pi]); glibc's reduce_large is not restated.
// Host + device: one source; the tests run both variants on the device against the host libm (orbx_debug_sincos).
#ifndef ORBX_SINCOS_H
#define ORBX_SINCOS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbx {

template <bool FUSED>
__host__ __device__ __forceinline__ double sc_ma(double a, double b, double c) {  // a * b + c
#pragma clang fp contract(off)
  if (FUSED) return __builtin_fma(a, b, c);
  const double p = a * b;
  return p + c;
}

// n even: sine polynomial of x (x2 = x^2), n odd: cosine polynomial; `neg` selects the negated cosine row.
template <bool FUSED>
__host__ __device__ __forceinline__ float sc_poly(double x, double x2, int n, bool neg) {
#pragma clang fp contract(off)
  if ((n & 1) == 0) {
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const double x3 = x * x2;
    const double s1 = sc_ma<FUSED>(x2, S3, S2);
    const double x7 = x3 * x2;
    const double s = sc_ma<FUSED>(x3, S1, x);
    return (float)sc_ma<FUSED>(x7, s1, s);
  }
  const double sg = neg ? -1.0 : 1.0;  // table[1] holds the negated cosine coefficients (exact sign flips)
  const double C0 = sg * 0x1p0, C1 = sg * -0x1.ffffffd0c621cp-2, C2 = sg * 0x1.55553e1068f19p-5,
               C3 = sg * -0x1.6c087e89a359dp-10, C4 = sg * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2;
  const double c2 = sc_ma<FUSED>(x2, C4, C3);
  const double c1 = sc_ma<FUSED>(x2, C1, C0);
  const double x6 = x4 * x2;
  const double c = sc_ma<FUSED>(x4, C2, c1);
  return (float)sc_ma<FUSED>(x6, c2, c);
}

template <bool FUSED>
__host__ __device__ __forceinline__ void glibc_sincosf(float y, float& s_out, float& c_out) {
#pragma clang fp contract(off)
  const uint32_t top = (__builtin_bit_cast(uint32_t, y) >> 20) & 0x7ffu;  // abstop12
  double x = (double)y;
  if (top < 0x3f4u) {      // |y| < pi/4
    if (top < 0x398u) {    // |y| < 2^-12
      s_out = y;
      c_out = 1.0f;
      return;
    }
    const double x2 = x * x;
    s_out = sc_poly<FUSED>(x, x2, 0, false);
    c_out = sc_poly<FUSED>(x, x2, 1, false);
    return;
  }
  // reduce_fast: hpi_inv is 2/pi * 2^24, so the rounded quotient sits in bits 24..31 of the truncated product
  const double r = x * 0x1.45F306DC9C883p+23;
  const int n = ((int)r + 0x800000) >> 24;
  const double hpi = 0x1.921FB54442D18p0;
  if (FUSED) {
    x = __builtin_fma(-(double)n, hpi, x);
  } else {
    const double nh = (double)n * hpi;
    x = x - nh;
  }
  const double sgn = ((n + 1) & 2) ? -1.0 : 1.0;  // sign[n & 3] = {1, -1, -1, 1}
  const bool neg = (n & 2) != 0;
  const double xs = x * sgn, x2 = x * x;
  s_out = sc_poly<FUSED>(xs, x2, n, neg);        // sinf: sinf_poly(x * s, x * x, p, n)
  c_out = sc_poly<FUSED>(xs, x2, n ^ 1, neg);    // cosf: sinf_poly(x * s, x * x, p, n ^ 1)
}

}  // namespace orbx
#endif
