// orbx_device.h — small device helpers shared by the kernel translation units of liborbx.
#ifndef ORBX_DEVICE_H
#define ORBX_DEVICE_H
#include "orbx_internal.h"

namespace orbx {

__device__ __forceinline__ int rne_f(float v) { return __float2int_rn(v); }  // cvRound
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }
// number of set bits of a wave mask below this lane: v_mbcnt_lo + v_mbcnt_hi (no 64-bit vector shifts)
__device__ __forceinline__ int prefix_count(uint64_t m) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

typedef unsigned short orbx_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t c) {  // a.lo*b.lo + a.hi*b.hi + c
  return __builtin_amdgcn_udot2(__builtin_bit_cast(orbx_us2, a), __builtin_bit_cast(orbx_us2, b), c, false);
}

// ---- cross-lane sums / scans / minima by DPP (gfx9 row shifts + row broadcasts): one VALU instruction per step, where
// __shfl_up / __shfl_xor cost a ds_bpermute round trip through the LDS pipe plus its address arithmetic.
// Inclusive prefix sum over the 64 lanes: prefix sums inside each row of 16 lanes (row_shr 1, 2, 4, 8), then the row
// totals travel down with row_bcast:15 (into rows 1 and 3) and row_bcast:31 (into rows 2 and 3).
__device__ __forceinline__ int wave_scan_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}
__device__ __forceinline__ int wave_sum_dpp(int v) {  // sum over the wave, wave-uniform
  return __builtin_amdgcn_readlane(wave_scan_dpp(v), 63);
}
__device__ __forceinline__ uint64_t wave_scan_dpp_u64(uint64_t v) {
#define ORBX_DPP_STEP64(CTRL, ROWMASK, BOUND)                                                              \
  {                                                                                                         \
    const uint32_t tl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWMASK, 0xf, BOUND);        \
    const uint32_t th = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWMASK, 0xf, BOUND); \
    v += ((uint64_t)th << 32) | tl;                                                                         \
  }
  ORBX_DPP_STEP64(0x111, 0xf, true)
  ORBX_DPP_STEP64(0x112, 0xf, true)
  ORBX_DPP_STEP64(0x114, 0xf, true)
  ORBX_DPP_STEP64(0x118, 0xf, true)
  ORBX_DPP_STEP64(0x142, 0xa, false)
  ORBX_DPP_STEP64(0x143, 0xc, false)
#undef ORBX_DPP_STEP64
  return v;
}
__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v) {  // minimum over the wave, wave-uniform
  // (lanes without a source keep their own value: `old` = v)
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// XCD-aware block -> tile mapping.  Workgroups go to the 8 XCDs round-robin by flat workgroup id, so neighbouring
// tiles land in 8 different L2s and every shared halo line is fetched from HBM once per XCD.  This permutation keeps
// the dispatch order balanced (every aligned chunk of 8*K flat ids still covers the same 8*K tiles) but hands each
// XCD a run of K consecutive tiles.  `by` is the slower grid index (image); chunks cut by an image boundary keep
// the identity order.
__device__ __forceinline__ int xcd_run_remap_rt(int bx, int nbx, int by, int K) {  // run length chosen at run time
  if (K <= 1) return bx;
  const int o = (int)(((unsigned)by * (unsigned)nbx) & 7u), xs = bx + o, ch = 8 * K;
  const int c0 = (xs / ch) * ch;
  if (c0 < o || c0 + ch > nbx + o) return bx;
  const int r = xs - c0;
  return c0 + (r & 7) * K + (r >> 3) - o;
}
template <int K>
__device__ __forceinline__ int xcd_run_remap(int bx, int nbx, int by) {
  if (K <= 1) return bx;
  constexpr int ch = 8 * K;  // (K a power of two: the chunk arithmetic is shifts and masks)
  const int o = (int)(((unsigned)by * (unsigned)nbx) & 7u), xs = bx + o;
  const int c0 = xs & ~(ch - 1);
  if (c0 < o || c0 + ch > nbx + o) return bx;
  const int r = xs - c0;
  return c0 + (r & 7) * K + (r >> 3) - o;
}

__device__ __forceinline__ int hamming256(const uint32_t* a, const uint32_t* b) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
  return d;
}

}  // namespace orbx
#endif
