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

__device__ __forceinline__ int hamming256(const uint32_t* a, const uint32_t* b) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
  return d;
}

}  // namespace orbx
#endif
