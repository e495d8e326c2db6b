// orbx_preproc.hip — image pre-processing in front of the extractor (gray, resize, cv::remap, CLAHE) and
// Frame::UndistortKeyPoints (src/Frame.cc:853-919).
#include "orbx_device.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace orbx {

// ================================================================================================ pre-processing
// cvtColor(..., COLOR_*2GRAY) for 8U (OpenCV >= 3.4.2 / 4.x: 15-bit coefficients, one rounding).  Thread per pixel.
__global__ __launch_bounds__(256) void k_cvt_gray(const uint8_t* __restrict__ src, int w, int h, long long sp, long long sip,
                                                  int cn, int rgb, uint8_t* __restrict__ dst, long long dp, long long dip) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= w || y >= h) return;
  const uint8_t* S = src + blockIdx.z * sip + y * sp + (long long)x * cn;
  const int c0 = S[0], g = S[1], c2 = S[2];
  const int r = rgb ? c0 : c2, b = rgb ? c2 : c0;
  dst[blockIdx.z * dip + y * dp + x] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15);
}

// cv::resize INTER_LINEAR 8U on interleaved channels with host-built coefficient tables (the B2 arithmetic of k_resize):
// thread per destination pixel, all channels.  A once-per-frame convenience kernel, not tiled.
__global__ __launch_bounds__(256) void k_resize_generic(const uint8_t* __restrict__ src, int sw, int sh, long long sp,
                                                        long long sip, int cn, uint8_t* __restrict__ dst, int dw, int dh,
                                                        long long dp, long long dip, const int* __restrict__ xofs,
                                                        const short* __restrict__ xab, const int* __restrict__ yofs,
                                                        const short* __restrict__ yab) {
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw || dy >= dh) return;
  src += blockIdx.z * sip;
  dst += blockIdx.z * dip;
  const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1), a0 = xab[2 * dx], a1 = xab[2 * dx + 1];
  const int sy = yofs[dy], b0 = yab[2 * dy], b1 = yab[2 * dy + 1];
  const uint8_t* R0 = src + (long long)min(max(sy, 0), sh - 1) * sp;
  const uint8_t* R1 = src + (long long)min(max(sy + 1, 0), sh - 1) * sp;
  for (int c = 0; c < cn; c++) {
    const int t0 = R0[sx * cn + c] * a0 + R0[sx1 * cn + c] * a1;
    const int t1 = R1[sx * cn + c] * a0 + R1[sx1 * cn + c] * a1;
    dst[dy * dp + (long long)dx * cn + c] = (uint8_t)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
  }
}

// The same conversion as a streaming kernel (round 6): a thread owns 16 consecutive pixels of a row -- three (four) 16-byte loads,
// one 16-byte store, eight vector instructions per pixel (byte extract x 3, 24-bit multiply-add x 3, shift, pack) -- for rows
// whose pixels start on 16-byte boundaries (base and pitches multiples of 16; x is a multiple of 16, so x * cn is too).  The
// last w % 16 pixels of a row and unaligned frames go through k_cvt_gray.  1 B written + cn B read per pixel: HBM-bound.
template <int CN>
__global__ __launch_bounds__(256) void k_cvt_gray16(const uint8_t* __restrict__ src, int segs, int h, long long sp, long long sip, int rgb,
                                                    uint8_t* __restrict__ dst, long long dp, long long dip) {
  const int item = blockIdx.x * 256 + threadIdx.x;   // (row, 16-pixel segment), row-major
  if (item >= segs * h) return;
  const int y = item / segs, sg = item - y * segs;
  const uint4* S = reinterpret_cast<const uint4*>(src + blockIdx.y * sip + y * sp + (long long)sg * 16 * CN);
  uint32_t d[4 * CN];
#pragma unroll
  for (int k = 0; k < CN; k++) {
    const uint4 q = S[k];
    d[4 * k] = q.x; d[4 * k + 1] = q.y; d[4 * k + 2] = q.z; d[4 * k + 3] = q.w;
  }
  const uint32_t w0 = rgb ? 9798u : 3735u, w2 = rgb ? 3735u : 9798u;   // weight of channel 0 / channel 2 (R or B first)
  uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
  for (int px = 0; px < 16; px++) {
    const int b0 = px * CN, b1 = b0 + 1, b2 = b0 + 2;   // byte positions of the pixel's three colour channels (compile time)
    const uint32_t c0 = (d[b0 >> 2] >> (8 * (b0 & 3))) & 255u, c1 = (d[b1 >> 2] >> (8 * (b1 & 3))) & 255u,
                   c2 = (d[b2 >> 2] >> (8 * (b2 & 3))) & 255u;
    const uint32_t g = (__umul24(c0, w0) + __umul24(c1, 19235u) + __umul24(c2, w2) + 16384u) >> 15;
    out[px >> 2] |= g << (8 * (px & 3));
  }
  *reinterpret_cast<uint4*>(dst + blockIdx.y * dip + y * dp + (long long)sg * 16) = make_uint4(out[0], out[1], out[2], out[3]);
}

hipError_t launch_cvt_gray(const uint8_t* src, int w, int h, long long sp, long long sip, int cn, int rgb, uint8_t* dst,
                           long long dp, long long dip, int nimg, hipStream_t s) {
  const bool aligned = !(((uintptr_t)src | (uintptr_t)dst | (uintptr_t)sp | (uintptr_t)sip | (uintptr_t)dp | (uintptr_t)dip) & 15);
  static const bool naive = getenv("ORBX_GRAY_NAIVE") && atoi(getenv("ORBX_GRAY_NAIVE")) != 0;   // A / B switch: the per-pixel kernel everywhere
  const int segs = aligned && !naive && (cn == 3 || cn == 4) ? w / 16 : 0;
  if (segs > 0) {
    const dim3 grid((unsigned)(((long long)segs * h + 255) / 256), nimg);
    if (cn == 3) hipLaunchKernelGGL(k_cvt_gray16<3>, grid, dim3(256), 0, s, src, segs, h, sp, sip, rgb, dst, dp, dip);
    else hipLaunchKernelGGL(k_cvt_gray16<4>, grid, dim3(256), 0, s, src, segs, h, sp, sip, rgb, dst, dp, dip);
  }
  const int x0 = 16 * segs;   // the rest of every row, pixel by pixel
  if (x0 < w)
    hipLaunchKernelGGL(k_cvt_gray, dim3((w - x0 + 255) / 256, h, nimg), dim3(256), 0, s, src + (long long)x0 * cn, w - x0, h, sp, sip, cn, rgb,
                       dst + x0, dp, dip);
  return hipGetLastError();
}
hipError_t launch_resize_generic(const uint8_t* src, int sw, int sh, long long sp, long long sip, int cn, uint8_t* dst, int dw,
                                 int dh, long long dp, long long dip, const int* xofs, const short* xab, const int* yofs,
                                 const short* yab, int nimg, hipStream_t s) {
  hipLaunchKernelGGL(k_resize_generic, dim3((dw + 255) / 256, dh, nimg), dim3(256), 0, s, src, sw, sh, sp, sip, cn, dst, dw, dh,
                     dp, dip, xofs, xab, yofs, yab);
  return hipGetLastError();
}

// cv::remap(src, dst, mapx, mapy, INTER_LINEAR, BORDER_CONSTANT 0) with CV_32FC1 maps on 8UC1/3/4 (src/System.cc:294-295).
// Fixed point exactly as OpenCV's RemapInvoker / remapBilinear: position = cvRound(map * 32), 5 fraction bits per axis,
// weights (32-fx)(32-fy)*32 ... (= BilinearTab_i, exact products) except fraction (0,0) whose 32768 saturates to 32767
// and is repaired on the last tap: {32767, 0, 0, 1}; out = (sum + 2^14) >> 15; taps outside the source are 0.
// A thread produces 4 consecutive destination pixels: two 16-byte map loads, 4 x 4 byte gathers (the maps are smooth, so a
// wave's gathers fall into a few cache lines), one dword store for single-channel images.  HBM-bound: 8 B of map per
// pixel against 1 B read + 1 B written.  Image i of a batch uses map i % nMaps (left / right eye).
__device__ __forceinline__ int cv_round_sse(float t) {  // cvtss2si: out-of-range and NaN give INT_MIN
  return fabsf(t) < 2147483648.f ? __float2int_rn(t) : (int)0x80000000;
}
__global__ __launch_bounds__(256) void k_remap(RemapArgs a) {
  const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + (threadIdx.x >> 6), img = blockIdx.z;
  if (x0 >= a.dw || y >= a.dh) return;
  const int m = img % a.nMaps;
  const float* MX = a.mapx + (long long)m * a.mapImgPitch + (long long)y * a.mapPitch + x0;
  const float* MY = a.mapy + (long long)m * a.mapImgPitch + (long long)y * a.mapPitch + x0;
  const uint8_t* S = a.src + (long long)img * a.srcImgPitch;
  uint8_t* D = a.dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + (long long)x0 * a.cn;
  float mx[4], my[4];
  const bool full = x0 + 3 < a.dw;
  if (full && a.mapVec4) {
    const float4 vx = *reinterpret_cast<const float4*>(MX), vy = *reinterpret_cast<const float4*>(MY);
    mx[0] = vx.x; mx[1] = vx.y; mx[2] = vx.z; mx[3] = vx.w;
    my[0] = vy.x; my[1] = vy.y; my[2] = vy.z; my[3] = vy.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool in = x0 + k < a.dw;
      mx[k] = in ? MX[k] : 0.f;
      my[k] = in ? MY[k] : 0.f;
    }
  }
  const int cn = a.cn;
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int fsx = cv_round_sse(mx[k] * 32.f), fsy = cv_round_sse(my[k] * 32.f);
    const int sx = min(max(fsx >> 5, -32768), 32767), sy = min(max(fsy >> 5, -32768), 32767);
    const int fx = fsx & 31, fy = fsy & 31;
    int w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
    if ((fx | fy) == 0) { w0 = 32767; w3 = 1; }
    const bool x0in = (unsigned)sx < (unsigned)a.sw, x1in = (unsigned)(sx + 1) < (unsigned)a.sw;
    const bool y0in = (unsigned)sy < (unsigned)a.sh, y1in = (unsigned)(sy + 1) < (unsigned)a.sh;
    const uint8_t* R0 = S + (long long)sy * a.srcPitch + (long long)sx * cn;
    const uint8_t* R1 = R0 + a.srcPitch;
    if (cn == 1) {
      const int p00 = (x0in && y0in) ? R0[0] : 0, p01 = (x1in && y0in) ? R0[1] : 0;
      const int p10 = (x0in && y1in) ? R1[0] : 0, p11 = (x1in && y1in) ? R1[1] : 0;
      const uint32_t r = (uint32_t)(p00 * w0 + p01 * w1 + p10 * w2 + p11 * w3 + 16384) >> 15;
      packed |= r << (8 * k);
    } else if (x0 + k < a.dw) {
      for (int c = 0; c < cn; c++) {
        const int p00 = (x0in && y0in) ? R0[c] : 0, p01 = (x1in && y0in) ? R0[cn + c] : 0;
        const int p10 = (x0in && y1in) ? R1[c] : 0, p11 = (x1in && y1in) ? R1[cn + c] : 0;
        D[k * cn + c] = (uint8_t)((uint32_t)(p00 * w0 + p01 * w1 + p10 * w2 + p11 * w3 + 16384) >> 15);
      }
    }
  }
  if (cn == 1) {
    if (full && a.dstVec4) {
      *reinterpret_cast<uint32_t*>(D) = packed;
    } else {
      for (int k = 0; k < 4 && x0 + k < a.dw; k++) D[k] = (uint8_t)(packed >> (8 * k));
    }
  }
}
// Single-channel batches: the images that share a map (image % nMaps) are processed in groups of kRemapGroup by the same
// thread, so the 8 B / pixel of map data and the fixed-point weights are fetched / built once per group instead of once per
// image -- the maps, not the pixels, are the kernel's HBM traffic.
constexpr int kRemapGroup = 8;
__global__ __launch_bounds__(256, 8) void k_remap1(RemapArgs a, int nimg, int xb, int yb, int nblk) {
  // Tile order (round 6): workgroups go to the 8 XCDs round-robin by flat id, so the 4-row tiles of one column strip -- which
  // share two of the three source rows a window touches -- used to land in eight different L2s.  XCD c now walks the c-th
  // eighth of the tiles in (map / group, column strip, row) order: vertical neighbours meet in one L2.
  const int chunk = (nblk + 7) >> 3, T = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
  if (T >= nblk) return;
  const int tz = T / (xb * yb), txy = T - tz * (xb * yb), tx = txy / yb, ty = txy - tx * yb;
  const int x0 = 4 * (tx * 64 + (threadIdx.x & 63)), y = ty * 4 + (threadIdx.x >> 6);
  if (x0 >= a.dw || y >= a.dh) return;
  const int m = tz % a.nMaps, grp = tz / a.nMaps;
  const float* MX = a.mapx + (long long)m * a.mapImgPitch + (long long)y * a.mapPitch + x0;
  const float* MY = a.mapy + (long long)m * a.mapImgPitch + (long long)y * a.mapPitch + x0;
  float mx[4], my[4];
  const bool full = x0 + 3 < a.dw;
  if (full && a.mapVec4) {
    const float4 vx = *reinterpret_cast<const float4*>(MX), vy = *reinterpret_cast<const float4*>(MY);
    mx[0] = vx.x; mx[1] = vx.y; mx[2] = vx.z; mx[3] = vx.w;
    my[0] = vy.x; my[1] = vy.y; my[2] = vy.z; my[3] = vy.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool in = x0 + k < a.dw;
      mx[k] = in ? MX[k] : -8.f;  // outside the source: no loads for the padding lanes
      my[k] = in ? MY[k] : -8.f;
    }
  }
  // Taps.  Per pixel: the horizontal pair (sx, sx + 1) of source rows sy and sy + 1, addresses clamped into the image;
  // taps outside the source get weight 0 (BORDER_CONSTANT 0), and when the clamp moved the pair by one column (sx == -1
  // or sx == sw - 1) the surviving weight moves to the other half of the pair.
  // The gathers are the cost of this kernel (scattered sub-dword loads run at a few lanes per clock), so the four pixels of
  // a thread share them: rectification maps are smooth, their 4 x 2 x 2 taps fall into an 8-byte window of three
  // consecutive source rows, which is fetched with three (unaligned) 8-byte loads; the pairs come out of the window with one
  // v_perm_b32 each (selector precomputed per pixel).  A thread whose taps do not fit (strong magnification, a seam of the
  // map) takes the per-pixel path: eight 16-bit loads.  Needs sw >= 8.
  int off0[4], off1[4];  // byte offsets of the pairs in source rows sy and sy + 1
  uint32_t wlo[4];       // w0 | w1 << 16
  uint32_t whi[4];       // w2 | w3 << 16
  int sxc[4], syc0[4], syc1[4];
  const int pitch = (int)a.srcPitch;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int fsx = cv_round_sse(mx[k] * 32.f), fsy = cv_round_sse(my[k] * 32.f);
    const int sx = min(max(fsx >> 5, -32768), 32767), sy = min(max(fsy >> 5, -32768), 32767);
    const int fx = fsx & 31, fy = fsy & 31;
    uint32_t w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
    if ((fx | fy) == 0) { w0 = 32767; w3 = 1; }
    const bool xin0 = (unsigned)sx < (unsigned)a.sw, xin1 = (unsigned)(sx + 1) < (unsigned)a.sw;
    const bool yin0 = (unsigned)sy < (unsigned)a.sh, yin1 = (unsigned)(sy + 1) < (unsigned)a.sh;
    if (!xin0) w0 = w2 = 0;
    if (!xin1) w1 = w3 = 0;
    if (!yin0) w0 = w1 = 0;
    if (!yin1) w2 = w3 = 0;
    sxc[k] = min(max(sx, 0), a.sw - 2);
    if (sxc[k] > sx) { w0 = w1; w2 = w3; w1 = w3 = 0; }       // sx == -1 (or further left, all weights already 0)
    else if (sxc[k] < sx) { w1 = w0; w3 = w2; w0 = w2 = 0; }  // sx == sw - 1 (or further right)
    wlo[k] = w0 | (w1 << 16);
    whi[k] = w2 | (w3 << 16);
    syc0[k] = min(max(sy, 0), a.sh - 1);
    syc1[k] = min(max(sy + 1, 0), a.sh - 1);
    off0[k] = syc0[k] * pitch + sxc[k];
    off1[k] = syc1[k] * pitch + sxc[k];
  }
  const int bx = min(min(min(sxc[0], sxc[1]), min(sxc[2], sxc[3])), a.sw - 8);
  const int by = min(min(syc0[0], syc0[1]), min(syc0[2], syc0[3]));
  bool fast = true;
  uint32_t selTop[4];             // v_perm selector: byte d -> bits 0..7, byte d + 1 -> bits 16..23 of the 8-byte window (both rows)
  uint32_t wrow[4][3];            // the pixel's weight pairs by WINDOW ROW: (w0, w1) on the row of its top pair, (w2, w3) on the row of its
                                  // bottom pair, 0 on the third -- a pixel is three perm + dot2 over the rows, no row selects
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int d = sxc[k] - bx, e0 = syc0[k] - by, e1 = syc1[k] - by;
    fast = fast && d <= 6 && e0 <= 1 && e1 <= 2;
    selTop[k] = (uint32_t)d | 0x0c000c00u | ((uint32_t)(d + 1) << 16);
#pragma unroll
    for (int j = 0; j < 3; j++) wrow[k][j] = (e0 == j ? wlo[k] : 0u) + (e1 == j ? whi[k] : 0u);   // (both on one row only when a clamped row carries weight 0)
  }
  // Round 6 (PMC: the texture addresser was busy 70 % of the launch at 45 cycles per UNALIGNED 8-byte load, and the per-pixel row
  // selects had been compiled into exec-mask branches: 1188 scalar against 901 vector instructions per wave): a row of the window
  // is three ALIGNED dwords from bx & ~3, shifted into place by two v_alignbyte, and the row selects are gone: the weights are laid out per window row.
  const int bs = bx & 3, bxa = bx - bs;
  fast = fast && bxa + 12 <= a.sw;   // (the three dwords stay inside the row)
  const int wo0 = by * pitch + bxa, wo1 = min(by + 1, a.sh - 1) * pitch + bxa, wo2 = min(by + 2, a.sh - 1) * pitch + bxa;
  const uint8_t* __restrict__ src = a.src;
  uint8_t* __restrict__ dst = a.dst;
  const int first = grp * kRemapGroup, perMap = (nimg - m + a.nMaps - 1) / a.nMaps;
  const int count = min(kRemapGroup, perMap - first);
  struct Row3 { uint32_t w[3]; };
  auto shift_row = [&](const Row3& r) { return make_uint2(__builtin_amdgcn_alignbyte(r.w[1], r.w[0], bs), __builtin_amdgcn_alignbyte(r.w[2], r.w[1], bs)); };
  auto blend_fast = [&](const Row3& q0, const Row3& q1, const Row3& q2) {
    const uint2 r0 = shift_row(q0), r1 = shift_row(q1), r2 = shift_row(q2);
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // three v_dot2_u32_u16 over the window rows: (p00, p01) . (w0, w1) + (p10, p11) . (w2, w3) + 0; every weight is below 2^15
      uint32_t acc = udot2_u16(__builtin_amdgcn_perm(r0.y, r0.x, selTop[k]), wrow[k][0], 16384u);
      acc = udot2_u16(__builtin_amdgcn_perm(r1.y, r1.x, selTop[k]), wrow[k][1], acc);
      acc = udot2_u16(__builtin_amdgcn_perm(r2.y, r2.x, selTop[k]), wrow[k][2], acc);
      packed |= (acc >> 15) << (8 * k);
    }
    return packed;
  };
  auto put = [&](int img, uint32_t packed) {
    uint8_t* D = dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + x0;
    if (full && a.dstVec4) {
      *reinterpret_cast<uint32_t*>(D) = packed;
    } else {
      for (int k = 0; k < 4 && x0 + k < a.dw; k++) D[k] = (uint8_t)(packed >> (8 * k));
    }
  };
  // The loop over the group's images is a chain load -> blend -> store per image: two images per trip, all six row loads in
  // flight before the first blend.
  // (wave-uniform choice: in the fast branch the per-pixel offsets of the slow path are dead, which keeps the kernel at eight
  // waves per SIMD; a wave with one seam thread takes the per-pixel loads for all of its lanes)
  if (__builtin_amdgcn_ballot_w64(!fast) == 0) {
    int g = 0;
    for (; g + 1 < count; g += 2) {
      const int imgA = m + a.nMaps * (first + g), imgB = imgA + a.nMaps;
      const uint8_t* SA = src + (long long)imgA * a.srcImgPitch;
      const uint8_t* SB = src + (long long)imgB * a.srcImgPitch;
      Row3 a0, a1, a2, b0, b1, b2;
      __builtin_memcpy(&a0, SA + wo0, 12);
      __builtin_memcpy(&a1, SA + wo1, 12);
      __builtin_memcpy(&a2, SA + wo2, 12);
      __builtin_memcpy(&b0, SB + wo0, 12);
      __builtin_memcpy(&b1, SB + wo1, 12);
      __builtin_memcpy(&b2, SB + wo2, 12);
      put(imgA, blend_fast(a0, a1, a2));
      put(imgB, blend_fast(b0, b1, b2));
    }
    if (g < count) {
      const int img = m + a.nMaps * (first + g);
      const uint8_t* S = src + (long long)img * a.srcImgPitch;
      Row3 r0, r1, r2;
      __builtin_memcpy(&r0, S + wo0, 12);
      __builtin_memcpy(&r1, S + wo1, 12);
      __builtin_memcpy(&r2, S + wo2, 12);
      put(img, blend_fast(r0, r1, r2));
    }
  } else {
    for (int g = 0; g < count; g++) {
      const int img = m + a.nMaps * (first + g);
      const uint8_t* S = src + (long long)img * a.srcImgPitch;
      uint32_t packed = 0;
      uint16_t t[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        __builtin_memcpy(&t[k], S + off0[k], 2);
        __builtin_memcpy(&b[k], S + off1[k], 2);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t acc = udot2_u16(__builtin_amdgcn_perm(0u, (uint32_t)t[k], 0x0c010c00u), wlo[k], 16384u);
        acc = udot2_u16(__builtin_amdgcn_perm(0u, (uint32_t)b[k], 0x0c010c00u), whi[k], acc);
        packed |= (acc >> 15) << (8 * k);
      }
      put(img, packed);
    }
  }
}
static int g_remap_lds = [] { const char* e = getenv("ORBX_REMAP_LDS"); return e ? atoi(e) : 1; }();   // test / A-B hook
                                        // (orbx_debug_set_remap_lds, ORBX_REMAP_LDS): 0 = the plans keep k_remap1
void debug_set_remap_lds(int on) { g_remap_lds = on; }
// ---------------------------------------------------------------------------------------------------------------- round 6b
// k_remap_lds: the same arithmetic with the source pixels staged through LDS.  k_remap1's gathers go through the texture
// addresser three 12-byte loads per thread and image (28 cycles each: the unit is busy 70 % of the launch) although
// the 1024 pixels of a workgroup read one compact footprint of the source.  Here the HOST works out that footprint once per
// plan (remap_tile_table below: per 128 x 8 output tile the bounding box of its taps, dword-aligned, with the margin the
// 12-byte windows and the third window row need), the workgroup copies it with coalesced dword loads -- for two images of the
// group per trip, the next trip's loads in flight under the current blends -- and the windows come from LDS.
// Used by the pre-processing plans whose every tile fits kRemapLdsDwords (rectification maps do; a plan with a wider tile, an
// unaligned source or cn != 1 keeps k_remap1 / k_remap).  Results are bit-identical: the taps, weights and blends are k_remap1's.
#ifndef REMAP_ABLATE
#define REMAP_ABLATE 0
#endif
#ifndef REMAP_ROWMAJOR
#define REMAP_ROWMAJOR 0
#endif
constexpr int kRemapTileW = 128, kRemapTileH = 8;
#ifndef REMAP_AHEAD
#define REMAP_AHEAD 1
#endif
constexpr int kRemapAhead = REMAP_AHEAD;            // trips of staging loads in flight (1: measured best -- 2 / 4 trips ahead ran 68 / 85 us against 44.7)
constexpr int kRemapLdsDwords = 1024;                 // per staged image: 4 KB, four dwords per thread
__global__ __launch_bounds__(256) void k_remap_lds(RemapArgs a, int nimg, int nblk, int gsz) {
  __shared__ uint4 stage[2][2][kRemapLdsDwords / 4];   // [trip parity][image of the trip][footprint]: 16 KB
  const int chunk = (nblk + 7) >> 3, T = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);   // XCD-contiguous tile order (k_remap1)
  if (T >= nblk) return;
  const int xb = a.tilesX, yb = a.tilesY;
#if REMAP_ROWMAJOR
  const int tz = T / (xb * yb), tq = T - tz * (xb * yb), ty = tq / xb, tx = tq - ty * xb, txy = tx * yb + ty;
#else
  const int tz = T / (xb * yb), txy = T - tz * (xb * yb), tx = txy / yb, ty = txy - tx * yb;
#endif
  const int m = tz % a.nMaps, grp = tz / a.nMaps;
  const int4 fp = *reinterpret_cast<const int4*>(a.tileTab + 8ll * ((long long)m * xb * yb + txy));        // x0a, y0, dwords per row, rows
  const int fmagic = a.tileTab[8ll * ((long long)m * xb * yb + txy) + 4];                                   // ceil(2^32 / dwords per row)
  const int x0a = fp.x, fy0 = fp.y, wQ = fp.z, total = fp.z * fp.w, P = 16 * fp.z;   // (16-byte units: one slot per thread)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int y = ty * kRemapTileH + 2 * w + (lane >> 5), x0 = tx * kRemapTileW + 4 * (lane & 31);
  const bool rowIn = y < a.dh;
  // the thread's four map entries; positions behind the right / bottom edge repeat the edge's entry (their taps stay inside the
  // footprint, their results are not stored)
  const int yr = min(y, a.dh - 1);
  const float* MX = a.mapx + (long long)m * a.mapImgPitch + (long long)yr * a.mapPitch;
  const float* MY = a.mapy + (long long)m * a.mapImgPitch + (long long)yr * a.mapPitch;
  float mx[4], my[4];
  const bool full = x0 + 3 < a.dw;
  if (full && a.mapVec4) {
    const float4 vx = *reinterpret_cast<const float4*>(MX + x0), vy = *reinterpret_cast<const float4*>(MY + x0);
    mx[0] = vx.x; mx[1] = vx.y; mx[2] = vx.z; mx[3] = vx.w;
    my[0] = vy.x; my[1] = vy.y; my[2] = vy.z; my[3] = vy.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int xr = min(x0 + k, a.dw - 1);
      mx[k] = MX[xr];
      my[k] = MY[xr];
    }
  }
  // staging slot: 16-byte piece tid of the footprint = (row tid / wQ, piece tid % wQ); threads behind the footprint's end idle
  const int pitch = (int)a.srcPitch;
  const bool sIn = tid < total;
  int sOff;
  {
    const int i = min(tid, total - 1);
    const int r = (int)__umulhi((unsigned)i, (unsigned)fmagic), c = i - r * wQ;
    sOff = (fy0 + r) * pitch + x0a + 16 * c;
  }
  // taps (k_remap1)
  auto taps = [&](float fmx, float fmy, int& sxk, int& sy0k, int& sy1k, uint32_t& wl, uint32_t& wh) {
    const int fsx = cv_round_sse(fmx * 32.f), fsy = cv_round_sse(fmy * 32.f);
    const int sx = min(max(fsx >> 5, -32768), 32767), sy = min(max(fsy >> 5, -32768), 32767);
    const int fx = fsx & 31, fy = fsy & 31;
    uint32_t w0 = (32 - fx) * (32 - fy) * 32, w1 = fx * (32 - fy) * 32, w2 = (32 - fx) * fy * 32, w3 = fx * fy * 32;
    if ((fx | fy) == 0) { w0 = 32767; w3 = 1; }
    const bool xin0 = (unsigned)sx < (unsigned)a.sw, xin1 = (unsigned)(sx + 1) < (unsigned)a.sw;
    const bool yin0 = (unsigned)sy < (unsigned)a.sh, yin1 = (unsigned)(sy + 1) < (unsigned)a.sh;
    if (!xin0) w0 = w2 = 0;
    if (!xin1) w1 = w3 = 0;
    if (!yin0) w0 = w1 = 0;
    if (!yin1) w2 = w3 = 0;
    sxk = min(max(sx, 0), a.sw - 2);
    if (sxk > sx) { w0 = w1; w2 = w3; w1 = w3 = 0; }
    else if (sxk < sx) { w1 = w0; w3 = w2; w0 = w2 = 0; }
    wl = w0 | (w1 << 16);
    wh = w2 | (w3 << 16);
    sy0k = min(max(sy, 0), a.sh - 1);
    sy1k = min(max(sy + 1, 0), a.sh - 1);
  };
  uint32_t wlo[4], whi[4];
  int sxc[4], syc0[4], syc1[4];
#pragma unroll
  for (int k = 0; k < 4; k++) taps(mx[k], my[k], sxc[k], syc0[k], syc1[k], wlo[k], whi[k]);
  const int bx = min(min(sxc[0], sxc[1]), min(sxc[2], sxc[3]));
  const int by = min(min(syc0[0], syc0[1]), min(syc0[2], syc0[3]));
  bool fast = true;
  uint32_t selTop[4], wrow[4][3];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int d = sxc[k] - bx, e0 = syc0[k] - by, e1 = syc1[k] - by;
    fast = fast && d <= 6 && e0 <= 1 && e1 <= 2;
    selTop[k] = (uint32_t)d | 0x0c000c00u | ((uint32_t)(d + 1) << 16);
#pragma unroll
    for (int j = 0; j < 3; j++) wrow[k][j] = (e0 == j ? wlo[k] : 0u) + (e1 == j ? whi[k] : 0u);
  }
  const int bs = bx & 3;
  fast = fast && bx - bs + 12 <= a.sw;   // (the footprint ends at the row end: remap_tile_table)
  // LDS byte offsets of the window rows (the footprint holds 12 bytes from bx & ~3 on rows by .. min(by + 2, sh - 1): remap_tile_table)
  const int lw0 = (by - fy0) * P + (bx - bs - x0a);
  const int lw1 = (min(by + 1, a.sh - 1) - fy0) * P + (bx - bs - x0a), lw2 = (min(by + 2, a.sh - 1) - fy0) * P + (bx - bs - x0a);
  const uint8_t* __restrict__ src = a.src;
  uint8_t* __restrict__ dst = a.dst;
  const int first = grp * gsz, perMap = (nimg - m + a.nMaps - 1) / a.nMaps;
  const int count = min(gsz, perMap - first);
  const bool allFast = __builtin_amdgcn_ballot_w64(!fast) == 0;
  auto blend = [&](const uint32_t* st) -> uint32_t {
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(st);
    uint32_t packed = 0;
#if REMAP_ABLATE == 2   // timing only: no blend
    return *reinterpret_cast<const uint32_t*>(sb + lw0);
#endif
    if (allFast) {
      const uint32_t* q0 = reinterpret_cast<const uint32_t*>(sb + lw0);
      const uint32_t* q1 = reinterpret_cast<const uint32_t*>(sb + lw1);
      const uint32_t* q2 = reinterpret_cast<const uint32_t*>(sb + lw2);
      const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], b0 = q1[0], b1 = q1[1], b2 = q1[2], c0 = q2[0], c1 = q2[1], c2 = q2[2];
      const uint2 r0 = make_uint2(__builtin_amdgcn_alignbyte(a1, a0, bs), __builtin_amdgcn_alignbyte(a2, a1, bs));
      const uint2 r1 = make_uint2(__builtin_amdgcn_alignbyte(b1, b0, bs), __builtin_amdgcn_alignbyte(b2, b1, bs));
      const uint2 r2 = make_uint2(__builtin_amdgcn_alignbyte(c1, c0, bs), __builtin_amdgcn_alignbyte(c2, c1, bs));
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t acc = udot2_u16(__builtin_amdgcn_perm(r0.y, r0.x, selTop[k]), wrow[k][0], 16384u);
        acc = udot2_u16(__builtin_amdgcn_perm(r1.y, r1.x, selTop[k]), wrow[k][1], acc);
        acc = udot2_u16(__builtin_amdgcn_perm(r2.y, r2.x, selTop[k]), wrow[k][2], acc);
        packed |= (acc >> 15) << (8 * k);
      }
    } else {   // a wave with a thread whose taps spread further than the window (a seam, strong magnification): byte reads.
               // The per-pixel taps are rebuilt from the map entries here (rare path) instead of living in 20 registers.
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int sxk, sy0k, sy1k;
        uint32_t wl, wh;
        taps(mx[k], my[k], sxk, sy0k, sy1k, wl, wh);
        const uint8_t* t = sb + (sy0k - fy0) * P + (sxk - x0a);
        const uint8_t* b = sb + (sy1k - fy0) * P + (sxk - x0a);
        uint32_t acc = udot2_u16((uint32_t)t[0] | ((uint32_t)t[1] << 16), wl, 16384u);
        acc = udot2_u16((uint32_t)b[0] | ((uint32_t)b[1] << 16), wh, acc);
        packed |= (acc >> 15) << (8 * k);
      }
    }
    return packed;
  };
  auto put = [&](int img, uint32_t packed) {
#if REMAP_ABLATE == 1   // timing only: no stores
    if (packed != 0x12345678u) return;
#endif
    if (!rowIn || x0 >= a.dw) return;
    uint8_t* D = dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + x0;
    if (full && a.dstVec4) {
      *reinterpret_cast<uint32_t*>(D) = packed;
    } else {
      for (int k = 0; k < 4 && x0 + k < a.dw; k++) D[k] = (uint8_t)(packed >> (8 * k));
    }
  };
  // Staging: one 16-byte load per thread and image (global_load_dwordx4 + ds_write_b128), two images per trip, the next trip's
  // loads in flight under the current blends.  Measured at 64 x 1280x720, 32 images per group (HISTORY.md, round 6): dword pieces
  // in registers one trip ahead 47.0 us, the same by LDS-DMA up to six trips ahead 47.0, these 16-byte pieces 44.7 -- and 68 / 85 us
  // with two / four trips of them in flight.  The timing ablation puts loads + stores alone at 29 us (185 MB over the memory side
  // at the measured copy rate), the blends alone at 19, the whole at their sum minus 4: more loads in flight do not buy overlap
  // here, they cost it.
  if (count <= 0) return;   // (uniform: map m has no image in this batch -- an odd batch's last map)
  const int ntrips = (count + 1) >> 1;
  uint4 ra[kRemapAhead], rb[kRemapAhead];   // trips t + 1 .. t + kRemapAhead in flight while trip t is blended
  auto fetch = [&](int trip, uint4& xa, uint4& xb) {
    const int tc = min(trip, ntrips - 1);   // (behind the last trip: a repeat of it, never staged)
    const int imgA = m + a.nMaps * (first + 2 * tc), imgB = m + a.nMaps * (first + min(2 * tc + 1, count - 1));
    if (sIn && REMAP_ABLATE != 3) {
      xa = *reinterpret_cast<const uint4*>(src + (long long)imgA * a.srcImgPitch + sOff);
      xb = *reinterpret_cast<const uint4*>(src + (long long)imgB * a.srcImgPitch + sOff);
    }
  };
#pragma unroll
  for (int q = 0; q < kRemapAhead; q++) fetch(q, ra[q], rb[q]);
  for (int tb = 0; tb < ntrips; tb += kRemapAhead) {
#pragma unroll
    for (int q = 0; q < kRemapAhead; q++) {
      const int t = tb + q;
      if (t >= ntrips) break;
      uint4* sA = stage[t & 1][0];
      uint4* sB = stage[t & 1][1];
      if (sIn) {
        sA[tid] = ra[q];
        sB[tid] = rb[q];
      }
      __syncthreads();   // (one barrier per trip: a buffer is written again two trips later, behind the barrier that follows its reads)
      fetch(t + kRemapAhead, ra[q], rb[q]);
      const int imgA = m + a.nMaps * (first + 2 * t);
      put(imgA, blend(reinterpret_cast<const uint32_t*>(sA)));
      if (2 * t + 1 < count) put(imgA + a.nMaps, blend(reinterpret_cast<const uint32_t*>(sB)));
    }
  }
}

// Footprints of k_remap_lds (host, once per plan).  For map m and output tile (tx, ty) -- kRemapTileW x kRemapTileH pixels --
// the taps of the tile's pixels are evaluated exactly as the kernel does (cvRound(32 x) >> 5, clamped into the source) and their
// bounding box is widened to what the kernel reads: columns from (min sx) & ~15 to past the 12-byte window of the right-most
// thread in 16-byte pieces, rows down to the third window row.  Entry = {x0a, y0, pieces per row, rows, ceil(2^32 / pieces per row), 0, 0, 0}.
// Returns false when a tile needs more than kRemapLdsDwords (the plan keeps k_remap1) or the footprint would leave the rows.
bool remap_tile_table(const float* mapx, const float* mapy, long long mapStride, int dw, int dh, int sw, int sh, int nMaps,
                      std::vector<int>& tab, int& tilesX, int& tilesY) {
  tilesX = (dw + kRemapTileW - 1) / kRemapTileW;
  tilesY = (dh + kRemapTileH - 1) / kRemapTileH;
  tab.assign((size_t)8 * tilesX * tilesY * nMaps, 0);
  if (sw < 32 || (sw & 15)) return false;   // (16-byte staging pieces inside the rows; narrower / unaligned sources keep k_remap1)
  auto cvr = [](float t) { return std::fabs(t) < 2147483648.f ? (int)std::nearbyintf(t) : (int)0x80000000; };
  for (int m = 0; m < nMaps; m++)
    for (int tx = 0; tx < tilesX; tx++)
      for (int ty = 0; ty < tilesY; ty++) {
        int minX = sw, maxX = 0, minY = sh, maxY = 0;
        for (int y = ty * kRemapTileH; y < std::min(dh, (ty + 1) * kRemapTileH); y++) {
          const float* MX = mapx + ((long long)m * dh + y) * mapStride;
          const float* MY = mapy + ((long long)m * dh + y) * mapStride;
          for (int x = tx * kRemapTileW; x < std::min(dw, (tx + 1) * kRemapTileW); x++) {
            const int fsx = cvr(MX[x] * 32.f), fsy = cvr(MY[x] * 32.f);
            const int sx = std::min(std::max(fsx >> 5, -32768), 32767), sy = std::min(std::max(fsy >> 5, -32768), 32767);
            const int sxc = std::min(std::max(sx, 0), sw - 2);
            const int y0c = std::min(std::max(sy, 0), sh - 1), y1c = std::min(std::max(sy + 1, 0), sh - 1);
            minX = std::min(minX, sxc); maxX = std::max(maxX, sxc);
            minY = std::min(minY, y0c); maxY = std::max(maxY, y1c);
          }
        }
        int x0a = minX & ~15;
        int endX = std::max(maxX + 2, (maxX & ~3) + 12);
        endX = std::min((endX + 15) & ~15, sw);           // (a window that would cross the row end is not taken by the kernel's fast path)
        if (endX - x0a < 32) x0a = endX - 32;             // (two pieces at least: the row / column split of a slot multiplies by 2^32 / pieces)
        const int wD = (endX - x0a) / 16, rows = std::min(maxY + 2, sh) - minY;
        if (wD <= 0 || rows <= 0 || (long long)wD * rows > kRemapLdsDwords / 4) return false;
        int* e = &tab[(size_t)8 * (((size_t)m * tilesX + tx) * tilesY + ty)];
        e[0] = x0a; e[1] = minY; e[2] = wD; e[3] = rows;
        e[4] = (int)(unsigned)((0x100000000ull + (unsigned)wD - 1) / (unsigned)wD);
      }
  return true;
}

hipError_t launch_remap(const RemapArgs& a, int nimg, hipStream_t s) {
  if (a.cn == 1 && a.tileTab && g_remap_lds) {
    // images per workgroup: the taps and weights of a tile are built once per group, so larger groups halve that share of the
    // vector work -- as long as the launch keeps a few workgroups per CU slot
    static const int gforce = [] { const char* e = getenv("ORBX_REMAP_GROUP"); return e ? atoi(e) : 0; }();
    const int perMap = (nimg + a.nMaps - 1) / a.nMaps, tiles = a.tilesX * a.tilesY * a.nMaps;
    int gsz = 8;
    while (gsz < 32 && gsz < perMap && (long long)tiles * ((perMap + 2 * gsz - 1) / (2 * gsz)) >= 1500) gsz *= 2;
    if (gforce > 0) gsz = gforce;
    const int groups = (perMap + gsz - 1) / gsz, nblk = tiles * groups;
    hipLaunchKernelGGL(k_remap_lds, dim3(8 * ((nblk + 7) / 8)), dim3(256), 0, s, a, nimg, nblk, gsz);
  } else if (a.cn == 1 && a.sw >= 8) {
    const int perMap = (nimg + a.nMaps - 1) / a.nMaps, groups = (perMap + kRemapGroup - 1) / kRemapGroup;
    const int xb = (a.dw + 255) / 256, yb = (a.dh + 3) / 4, nblk = xb * yb * a.nMaps * groups;
    hipLaunchKernelGGL(k_remap1, dim3(8 * ((nblk + 7) / 8)), dim3(256), 0, s, a, nimg, xb, yb, nblk);
  } else {
    hipLaunchKernelGGL(k_remap, dim3((a.dw + 255) / 256, (a.dh + 3) / 4, nimg), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// cv::CLAHE::apply on 8UC1 (Examples/Stereo/stereo_tum_vi.cc:100,142-143; OpenCV clahe.cpp).  Two kernels:
//  k_clahe_lut   block per (tile, image): per-wave LDS histograms of the tile (BORDER_REFLECT_101 extension at the right /
//                bottom when the image does not divide into tiles), clip + redistribution (clipped / 256 to every bin,
//                one extra count to every (256 / residual)-th bin), block prefix sum, lut = rne(cumsum * 255.f / area);
//  k_clahe_apply thread per 4 pixels: the float bilinear blend of the four neighbouring tiles' lut entries in
//                OpenCV's expression order (the TU is built with -ffp-contract=off), rne + saturate.
// The lut of an image (tilesX * tilesY * 256 B = 16 KB for 8x8) stays in L1 / L2 for the apply pass.
__global__ __launch_bounds__(256) void k_clahe_lut(ClaheArgs a) {
  // 16 histogram copies, copy = lane & 15, stride 257 words: neighbouring pixels carry (nearly) the same grey value, and LDS
  // atomics on one address retire one lane per clock -- a single copy per wave made flat image regions run 16x slower.
  __shared__ int hist[16 * 257];
  __shared__ int wsum[4];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = blockIdx.x, img = blockIdx.y;
  const int ty = tile / a.tilesX, tx = tile - ty * a.tilesX;
  const uint8_t* S = a.src + (long long)img * a.srcImgPitch;
  // (round 6) a 64 x 64 tile is one 16-byte load per thread: requested BEFORE the histograms are cleared, so that the clear and
  // its barrier run under the load's latency
  const bool wideTile = (a.tw & 15) == 0 && (tx + 1) * a.tw <= a.w && (ty + 1) * a.th <= a.h && (a.srcPitch & 15) == 0 &&
                        (a.srcImgPitch & 15) == 0 && ((uintptr_t)a.src & 15) == 0;
  const int spr = a.tw >> 4, nseg = spr * a.th;
  const uint8_t* T0 = S + (long long)(ty * a.th) * a.srcPitch + tx * a.tw;
  uint4 q0 = make_uint4(0, 0, 0, 0);
  if (wideTile && tid < nseg) {
    const int yy = tid / spr, xsg = tid - yy * spr;
    q0 = *reinterpret_cast<const uint4*>(T0 + (long long)yy * a.srcPitch + 16 * xsg);
  }
  for (int k = tid; k < 16 * 257; k += 256) hist[k] = 0;
  __syncthreads();
  // thread per 4 pixels of a tile row (sub-dword loads run at a fraction of the dword rate); quads that touch the tile's right
  // edge or the reflected extension go pixel by pixel
  const int qpr = (a.tw + 3) >> 2, nquads = qpr * a.th;
  const float inv_qpr = 1.0f / (float)qpr;
  const int hcopy = (lane & 15) * 257;
  if (wideTile) {
    for (int i = tid; i < nseg; i += 256) {
      uint4 q = q0;
      if (i != tid) {
        const int yy = i / spr, xsg = i - yy * spr;
        q = *reinterpret_cast<const uint4*>(T0 + (long long)yy * a.srcPitch + 16 * xsg);
      }
      const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        atomicAdd(&hist[hcopy + (qw[k] & 255)], 1);
        atomicAdd(&hist[hcopy + ((qw[k] >> 8) & 255)], 1);
        atomicAdd(&hist[hcopy + ((qw[k] >> 16) & 255)], 1);
        atomicAdd(&hist[hcopy + (qw[k] >> 24)], 1);
      }
    }
  } else
  for (int i = tid; i < nquads; i += 256) {
    int yy = (int)((float)i * inv_qpr);
    int xq = i - yy * qpr;
    if (xq < 0) { yy--; xq += qpr; }
    if (xq >= qpr) { yy++; xq -= qpr; }
    const int y = ty * a.th + yy, xx = 4 * xq, x = tx * a.tw + xx;
    if (y < a.h && xx + 3 < a.tw && x + 3 < a.w) {
      uint32_t q;
      __builtin_memcpy(&q, S + (long long)y * a.srcPitch + x, 4);
      atomicAdd(&hist[hcopy + (q & 255)], 1);
      atomicAdd(&hist[hcopy + ((q >> 8) & 255)], 1);
      atomicAdd(&hist[hcopy + ((q >> 16) & 255)], 1);
      atomicAdd(&hist[hcopy + (q >> 24)], 1);
    } else {
      int yr = y;
      while (yr >= a.h || yr < 0) yr = yr < 0 ? -yr : 2 * a.h - 2 - yr;  // reflect 101 (the extension is shorter than the image)
      for (int k = 0; k < 4 && xx + k < a.tw; k++) {
        int xr = x + k;
        while (xr >= a.w || xr < 0) xr = xr < 0 ? -xr : 2 * a.w - 2 - xr;
        atomicAdd(&hist[hcopy + S[(long long)yr * a.srcPitch + xr]], 1);
      }
    }
  }
  __syncthreads();
  int v = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) v += hist[k * 257 + tid];
  if (a.clip > 0) {
    int ex = max(v - a.clip, 0);
    v -= ex;
    ex = wave_sum_dpp(ex);   // (DPP row shifts: a __shfl_xor step is a ds_bpermute round trip)
    if (lane == 0) wsum[wave] = ex;
    __syncthreads();
    const int clipped = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const int batch = clipped >> 8, residual = clipped & 255;
    v += batch;
    if (residual) {
      const int step = max(256 / residual, 1);
      if (tid % step == 0 && tid / step < residual) v++;
    }
  }
  int sum = wave_scan_dpp(v);  // inclusive prefix sum over the 256 bins
  if (lane == 63) wsum[wave] = sum;
  __syncthreads();
  for (int k = 0; k < wave; k++) sum += wsum[k];
  const int r = __float2int_rn((float)sum * a.lutScale);
  a.lut[((long long)img * a.tilesX * a.tilesY + tile) * 256 + tid] = (uint8_t)min(max(r, 0), 255);
}

__global__ __launch_bounds__(256) void k_clahe_apply(ClaheArgs a) {
  const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + (threadIdx.x >> 6), img = blockIdx.z;
  if (x0 >= a.w || y >= a.h) return;
  const float tyf = (float)y * a.invTh - 0.5f;
  int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
  const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
  ty1 = max(ty1, 0);
  ty2 = min(ty2, a.tilesY - 1);
  const uint8_t* L = a.lut + (long long)img * a.tilesX * a.tilesY * 256;
  const uint8_t* L1 = L + (long long)ty1 * a.tilesX * 256;
  const uint8_t* L2 = L + (long long)ty2 * a.tilesX * 256;
  const uint8_t* S = a.src + (long long)img * a.srcImgPitch + (long long)y * a.srcPitch + x0;
  uint8_t* D = a.dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + x0;
  const bool full = x0 + 3 < a.w;
  uint32_t in4;
  if (full && a.srcVec4) {
    in4 = *reinterpret_cast<const uint32_t*>(S);
  } else {
    in4 = 0;
    for (int k = 0; k < 4 && x0 + k < a.w; k++) in4 |= (uint32_t)S[k] << (8 * k);
  }
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float txf = (float)(x0 + k) * a.invTw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0);
    tx2 = min(tx2, a.tilesX - 1);
    const int v = (in4 >> (8 * k)) & 255;
    const float l11 = (float)L1[tx1 * 256 + v], l12 = (float)L1[tx2 * 256 + v];
    const float l21 = (float)L2[tx1 * 256 + v], l22 = (float)L2[tx2 * 256 + v];
    const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    const int r = __float2int_rn(res);
    packed |= (uint32_t)min(max(r, 0), 255) << (8 * k);
  }
  if (full && a.dstVec4) {
    *reinterpret_cast<uint32_t*>(D) = packed;
  } else {
    for (int k = 0; k < 4 && x0 + k < a.w; k++) D[k] = (uint8_t)(packed >> (8 * k));
  }
}
// Fast apply pass.  Between the centres of four neighbouring tiles (an "interpolation cell": fixed tx1, tx2, ty1, ty2) the
// four lut bytes of a grey value can be packed into one dword, so a pixel costs ONE gather instead of four:
//  k_clahe_pack    cell tables [img][tilesY + 1][tilesX + 1][256] = l11 | l12 << 8 | l21 << 16 | l22 << 24 (81 KB per image
//                  for 8 x 8 tiles, L2-resident);
//  k_clahe_apply4  thread per 4 pixels: dword load, 4 x (dword gather, 4 v_cvt_f32_ubyte, blend), dword store.
// (A variant that staged the cell tables of a row band in LDS was slower: the tables are as large as the slab they serve.)
__global__ __launch_bounds__(256) void k_clahe_pack(ClaheArgs a, uint32_t* __restrict__ cells) {
  const int cx = blockIdx.x % (a.tilesX + 1), cy = blockIdx.x / (a.tilesX + 1), img = blockIdx.y, v = threadIdx.x;
  const int tx1 = max(cx - 1, 0), tx2 = min(cx, a.tilesX - 1), ty1 = max(cy - 1, 0), ty2 = min(cy, a.tilesY - 1);
  const uint8_t* L = a.lut + (long long)img * a.tilesX * a.tilesY * 256;
  const uint32_t l11 = L[(ty1 * a.tilesX + tx1) * 256 + v], l12 = L[(ty1 * a.tilesX + tx2) * 256 + v];
  const uint32_t l21 = L[(ty2 * a.tilesX + tx1) * 256 + v], l22 = L[(ty2 * a.tilesX + tx2) * 256 + v];
  cells[((long long)img * (a.tilesY + 1) * (a.tilesX + 1) + blockIdx.x) * 256 + v] = l11 | (l12 << 8) | (l21 << 16) | (l22 << 24);
}
__global__ __launch_bounds__(256) void k_clahe_apply4(ClaheArgs a, const uint32_t* __restrict__ cells) {
  const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + (threadIdx.x >> 6), img = blockIdx.z;
  if (x0 >= a.w || y >= a.h) return;
  const float tyf = (float)y * a.invTh - 0.5f;
  const int ty1 = (int)floorf(tyf);
  const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
  const int ncx = a.tilesX + 1;
  const uint32_t* C = cells + ((long long)img * (a.tilesY + 1) + min(max(ty1 + 1, 0), a.tilesY)) * ncx * 256;
  const uint8_t* S = a.src + (long long)img * a.srcImgPitch + (long long)y * a.srcPitch + x0;
  uint8_t* D = a.dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + x0;
  const bool full = x0 + 3 < a.w;
  uint32_t in4;
  if (full && a.srcVec4) {
    in4 = *reinterpret_cast<const uint32_t*>(S);
  } else {
    in4 = 0;
    for (int k = 0; k < 4 && x0 + k < a.w; k++) in4 |= (uint32_t)S[k] << (8 * k);
  }
  uint32_t e[4];
  float xa[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float txf = (float)(x0 + k) * a.invTw - 0.5f;
    const int tx1 = (int)floorf(txf);
    xa[k] = txf - (float)tx1;
    e[k] = C[min(max(tx1 + 1, 0), a.tilesX) * 256 + ((in4 >> (8 * k)) & 255)];
  }
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float xa1 = 1.0f - xa[k];
    const float l11 = (float)(e[k] & 255), l12 = (float)((e[k] >> 8) & 255), l21 = (float)((e[k] >> 16) & 255), l22 = (float)(e[k] >> 24);
    const float res = (l11 * xa1 + l12 * xa[k]) * ya1 + (l21 * xa1 + l22 * xa[k]) * ya;
    packed |= (uint32_t)min(max(__float2int_rn(res), 0), 255) << (8 * k);
  }
  if (full && a.dstVec4) {
    *reinterpret_cast<uint32_t*>(D) = packed;
  } else {
    for (int k = 0; k < 4 && x0 + k < a.w; k++) D[k] = (uint8_t)(packed >> (8 * k));
  }
}
// Round 6: 16 pixels per thread (one 16-byte load, sixteen table gathers in flight, one 16-byte store) when the rows allow it
// (width a multiple of 16, 16-byte aligned rows): a 4-pixel thread was one dependent load -> gather -> store chain per wave and
// the launch ran at 1.7 TB/s of its 6 -- latency, not bandwidth.  Same arithmetic, same order.
__global__ __launch_bounds__(256) void k_clahe_apply16(ClaheArgs a, const uint32_t* __restrict__ cells, int tpr) {
  const int t = blockIdx.x * 256 + threadIdx.x, img = blockIdx.y;
  const int y = t / tpr, x0 = 16 * (t - y * tpr);
  if (y >= a.h) return;
  const float tyf = (float)y * a.invTh - 0.5f;
  const int ty1 = (int)floorf(tyf);
  const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
  const int ncx = a.tilesX + 1;
  const uint32_t* C = cells + ((long long)img * (a.tilesY + 1) + min(max(ty1 + 1, 0), a.tilesY)) * ncx * 256;
  const uint4 in = *reinterpret_cast<const uint4*>(a.src + (long long)img * a.srcImgPitch + (long long)y * a.srcPitch + x0);
  const uint32_t inw[4] = {in.x, in.y, in.z, in.w};
  uint32_t e[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const float txf = (float)(x0 + k) * a.invTw - 0.5f;
    const int tx1 = (int)floorf(txf);
    e[k] = C[min(max(tx1 + 1, 0), a.tilesX) * 256 + ((inw[k >> 2] >> (8 * (k & 3))) & 255)];
  }
  uint32_t outw[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const float txf = (float)(x0 + k) * a.invTw - 0.5f;
    const int tx1 = (int)floorf(txf);
    const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
    const float l11 = (float)(e[k] & 255), l12 = (float)((e[k] >> 8) & 255), l21 = (float)((e[k] >> 16) & 255), l22 = (float)(e[k] >> 24);
    const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    outw[k >> 2] |= (uint32_t)min(max(__float2int_rn(res), 0), 255) << (8 * (k & 3));
  }
  *reinterpret_cast<uint4*>(a.dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + x0) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
}
// Round 6, the product path for the usual geometry (tile width a multiple of 32, 16-byte aligned rows): ONE workgroup per
// interpolation cell.  PMC of the per-pixel table gathers above (profiles/r6a_preproc_pmc.txt): the texture addresser was busy 76 %
// of k_clahe_apply16 -- a wave's 64 scattered dword gathers cost it 37 cycles -- so the cell's table (256 dwords: the four
// neighbouring tiles' lut bytes of every grey value, packed) is built in LDS by the workgroup itself from the four luts (no
// k_clahe_pack launch, no cell-table buffer) and the 16 look-ups of a thread are LDS reads.  A cell is tw x th pixels (half that at
// the image border) = tw / 16 segments per row; thread = (segment, row phase).  Same float blend in OpenCV's expression order.
__global__ __launch_bounds__(256) void k_clahe_apply_cell(ClaheArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t tab[256];
  const int ncx = a.tilesX + 1, cx = blockIdx.x % ncx, cy = blockIdx.x / ncx, img = blockIdx.y, tid = threadIdx.x;
  if (tid < 64) {   // the four luts as DWORDS by one wave (16 byte loads per workgroup before: the texture addresser takes a wave's
                    // byte load lane by lane), transposed to one table entry per grey value by four v_perm
    const int tx1 = max(cx - 1, 0), tx2 = min(cx, a.tilesX - 1), ty1 = max(cy - 1, 0), ty2 = min(cy, a.tilesY - 1);
    const uint32_t* L = reinterpret_cast<const uint32_t*>(a.lut + (long long)img * a.tilesX * a.tilesY * 256);   // (256-byte luts: aligned)
    const uint32_t l11 = L[(ty1 * a.tilesX + tx1) * 64 + tid], l12 = L[(ty1 * a.tilesX + tx2) * 64 + tid];
    const uint32_t l21 = L[(ty2 * a.tilesX + tx1) * 64 + tid], l22 = L[(ty2 * a.tilesX + tx2) * 64 + tid];
    // entry of grey value 4 tid + j = byte j of (l11, l12, l21, l22)
    const uint32_t lo01 = __builtin_amdgcn_perm(l12, l11, 0x05010400u), hi01 = __builtin_amdgcn_perm(l12, l11, 0x07030602u);   // [a0 b0 a1 b1], [a2 b2 a3 b3]
    const uint32_t lo23 = __builtin_amdgcn_perm(l22, l21, 0x05010400u), hi23 = __builtin_amdgcn_perm(l22, l21, 0x07030602u);
    uint4 e;
    e.x = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);   // [a0 b0 c0 d0]
    e.y = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);   // [a1 b1 c1 d1]
    e.z = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
    e.w = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
    reinterpret_cast<uint4*>(tab)[tid] = e;
  }
  // pixel (x, y) belongs to cell (floor(x / tw - 0.5) + 1, floor(y / th - 0.5) + 1): columns [cx tw - tw / 2, cx tw + tw / 2)
  const int xb = max(cx * a.tw - (a.tw >> 1), 0), xe = min(cx * a.tw + (a.tw >> 1), a.w);
  const int yb = max(cy * a.th - (a.th >> 1), 0), ye = min(cy * a.th + (a.th >> 1), a.h);
  const int spr = (xe - xb) >> 4;               // 16-pixel segments per row of the cell (tw % 32 == 0: xb, xe are multiples of 16)
  __syncthreads();
  if (spr <= 0) return;
  const int rpp = 256 / spr;                    // rows per pass (spr <= 16 is checked on the host)
  const int sg = tid % spr, r0 = tid / spr;
  if (r0 >= rpp) return;
  const int x0 = xb + 16 * sg;
  float xa[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const float txf = (float)(x0 + k) * a.invTw - 0.5f;
    xa[k] = txf - (float)(int)floorf(txf);
  }
  for (int y = yb + r0; y < ye; y += rpp) {
    const float tyf = (float)y * a.invTh - 0.5f;
    const float ya = tyf - (float)(int)floorf(tyf), ya1 = 1.0f - ya;
    const uint4 in = *reinterpret_cast<const uint4*>(a.src + (long long)img * a.srcImgPitch + (long long)y * a.srcPitch + x0);
    const uint32_t inw[4] = {in.x, in.y, in.z, in.w};
    uint32_t outw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t e = tab[(inw[k >> 2] >> (8 * (k & 3))) & 255];
      const float xa1 = 1.0f - xa[k];
      const float l11 = (float)(e & 255), l12 = (float)((e >> 8) & 255), l21 = (float)((e >> 16) & 255), l22 = (float)(e >> 24);
      const float res = (l11 * xa1 + l12 * xa[k]) * ya1 + (l21 * xa1 + l22 * xa[k]) * ya;
      outw[k >> 2] |= (uint32_t)min(max(__float2int_rn(res), 0), 255) << (8 * (k & 3));
    }
    *reinterpret_cast<uint4*>(a.dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + x0) = make_uint4(outw[0], outw[1], outw[2], outw[3]);
  }
}
size_t clahe_cells_bytes(const ClaheArgs& a, int nimg) {
  return (size_t)nimg * (a.tilesY + 1) * (a.tilesX + 1) * 256 * sizeof(uint32_t);
}
// The per-cell kernel cuts the image into cells with integer arithmetic; the reference's rule is floor((float)x * invTw - 0.5f)
// evaluated per pixel in float.  They agree when the float expression changes exactly at x = k tw + tw / 2 (always for power-of-two
// tiles); checked here on both sides of every boundary with the kernel's own float operations, otherwise the gather kernels run.
static bool clahe_cells_exact(int n, int t, int tiles, float inv) {
  for (int c = 0; c <= tiles; c++) {
    const int b = max(c * t - (t >> 1), 0), e = min(c * t + (t >> 1), n);
    if (b >= e) continue;
    for (int x : {b, e - 1}) {
      const float f = (float)x * inv - 0.5f;
      if (min(max((int)floorf(f) + 1, 0), tiles) != c) return false;
    }
  }
  return true;
}
static int g_clahe_cell_kernel = 1;   // test hook: 0 = the gather kernels also where the per-cell kernel applies
void debug_set_clahe_cell_kernel(int on) { g_clahe_cell_kernel = on; }
hipError_t launch_clahe(const ClaheArgs& a, int nimg, uint32_t* cells, hipStream_t s) {
  hipLaunchKernelGGL(k_clahe_lut, dim3(a.tilesX * a.tilesY, nimg), dim3(256), 0, s, a);
  const dim3 grid((a.w + 255) / 256, (a.h + 3) / 4, nimg);
  // (the reflect-101 extension of k_clahe_lut keeps tw * tilesX >= w: a cell never reaches past the image by more than it is clipped)
  const bool wide = a.w % 16 == 0 && a.srcPitch % 16 == 0 && a.dstPitch % 16 == 0 && a.srcImgPitch % 16 == 0 && a.dstImgPitch % 16 == 0 &&
                    ((uintptr_t)a.src & 15) == 0 && ((uintptr_t)a.dst & 15) == 0;
  if (wide && a.tw % 32 == 0 && a.tw <= 256 && g_clahe_cell_kernel && clahe_cells_exact(a.w, a.tw, a.tilesX, a.invTw) &&
      clahe_cells_exact(a.h, a.th, a.tilesY, a.invTh)) {
    hipLaunchKernelGGL(k_clahe_apply_cell, dim3((a.tilesX + 1) * (a.tilesY + 1), nimg), dim3(256), 0, s, a);
  } else if (cells) {
    hipLaunchKernelGGL(k_clahe_pack, dim3((a.tilesX + 1) * (a.tilesY + 1), nimg), dim3(256), 0, s, a, cells);
    if (wide) {
      const int tpr = a.w / 16;
      hipLaunchKernelGGL(k_clahe_apply16, dim3((tpr * a.h + 255) / 256, nimg), dim3(256), 0, s, a, cells, tpr);
    } else
    hipLaunchKernelGGL(k_clahe_apply4, grid, dim3(256), 0, s, a, cells);
  } else {
    hipLaunchKernelGGL(k_clahe_apply, grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// ================================================================================================ undistort
// cv::undistortPoints as Frame::UndistortKeyPoints / ComputeImageBounds call it (src/Frame.cc:853-919): double
// arithmetic in OpenCV's expression order, no contraction (TU flag) -- identical to the oracle's.
__global__ __launch_bounds__(256) void k_undistort(UndistortArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  double k[12];
#pragma unroll
  for (int j = 0; j < 12; j++) k[j] = (double)a.k[j];
  const double fx = a.K[0], fy = a.K[1], cx = a.K[2], cy = a.K[3];
  const double ifx = 1. / fx, ify = 1. / fy;
  const double u = a.in[(long long)i * a.stride], v = a.in[(long long)i * a.stride + 1];
  double x = (u - cx) * ifx, y = (v - cy) * ify;
  const double x0 = x, y0 = y;
  if (a.hasDist) {
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) {
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
  const double xx = fx * x + 0. * y + cx, yy = 0. * x + fy * y + cy, ww = 1. / (0. * x + 0. * y + 1.);
  a.out[(long long)i * a.stride] = (float)(xx * ww);
  a.out[(long long)i * a.stride + 1] = (float)(yy * ww);
}

hipError_t launch_undistort(const UndistortArgs& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_undistort, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace orbx
