// Micro-benchmark: issue cost of one wave64 VALU instruction on gfx950, per instruction class.
//
// Each kernel runs ITER x 128 copies of ONE instruction (inline asm, 16 independent destination registers, so no
// result is needed before 15 other instructions have issued) and brackets the loop with s_memtime (shader cycles)
// and s_memrealtime (100 MHz).  Three occupancies:
//   w1  one wave on the whole GPU                     -> cycles / instruction a single wave can reach
//   w2  two waves on ONE SIMD (a 512-thread block puts waves 0 and 4 on SIMD 0; only those two run the loop)
//   w8  8 waves on every SIMD of every CU (32 x CUs blocks of 256 threads, i.e. 4 rounds of 8 resident blocks) -> cycles / instruction / SIMD at saturation,
//       from the kernel's HIP-event duration and the clock measured inside it
// Output: one line per class; the w8 column is the number the VALU roofline of DESIGN.md 5 uses.
// Build: hipcc --offload-arch=gfx950 -O2 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

struct Rec {
  unsigned long long cyc, rt;
  uint32_t sink, pad;
};

#define KERNEL(NAME, ASM)                                                                                  \
  __global__ __launch_bounds__(1024) void k_##NAME(Rec* out, int iters, uint32_t seed, int active_mask) { \
    uint32_t a[16];                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = seed * (i + 3) + threadIdx.x * 17;               \
    uint32_t b = seed ^ (threadIdx.x * 0x9E3779B9u), c = (seed >> 3) + threadIdx.x;                        \
    const int wave = threadIdx.x >> 6;                                                                     \
    const uint32_t sk = __builtin_amdgcn_readfirstlane(seed | 0xff);                                       \
    asm volatile("v_cmp_gt_u32 vcc, 5, %0" ::"v"(threadIdx.x) : "vcc");                                   \
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;                                                     \
    if ((active_mask >> (wave & 15)) & 1) {                                                                \
      r0 = __builtin_amdgcn_s_memrealtime();                                                               \
      t0 = __builtin_amdgcn_s_memtime();                                                                   \
      for (int it = 0; it < iters; it++) {                                                                 \
        _Pragma("unroll") for (int r = 0; r < 8; r++) {                                                    \
          _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(sk)); \
        }                                                                                                  \
      }                                                                                                    \
      t1 = __builtin_amdgcn_s_memtime();                                                                   \
      r1 = __builtin_amdgcn_s_memrealtime();                                                               \
    }                                                                                                      \
    uint32_t s = 0;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= a[i];                                              \
    if ((threadIdx.x & 63) == 0) {                                                                         \
      Rec rec;                                                                                             \
      rec.cyc = t1 - t0;                                                                                   \
      rec.rt = r1 - r0;                                                                                    \
      rec.sink = s;                                                                                        \
      rec.pad = 0;                                                                                         \
      out[blockIdx.x * (blockDim.x >> 6) + wave] = rec;                                                    \
    }                                                                                                      \
  }

// 64-bit destinations (packed f32, f64, mad_u64): separate macro with register pairs
#define KERNEL64(NAME, ASM)                                                                                \
  __global__ __launch_bounds__(1024) void k_##NAME(Rec* out, int iters, uint32_t seed, int active_mask) { \
    double a[16];                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 16; i++) a[i] = __hiloint2double(0x3f800000 + i, 0x3f800000 + threadIdx.x); \
    double b = __hiloint2double(0x3f810000, 0x3f800100), c = __hiloint2double(0x3a810000, 0x3a800100);     \
    const int wave = threadIdx.x >> 6;                                                                     \
    const uint32_t sk = __builtin_amdgcn_readfirstlane(seed | 0xff);                                       \
    asm volatile("v_cmp_gt_u32 vcc, 5, %0" ::"v"(threadIdx.x) : "vcc");                                   \
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;                                                     \
    if ((active_mask >> (wave & 15)) & 1) {                                                                \
      r0 = __builtin_amdgcn_s_memrealtime();                                                               \
      t0 = __builtin_amdgcn_s_memtime();                                                                   \
      for (int it = 0; it < iters; it++) {                                                                 \
        _Pragma("unroll") for (int r = 0; r < 8; r++) {                                                    \
          _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c), "s"(sk)); \
        }                                                                                                  \
      }                                                                                                    \
      t1 = __builtin_amdgcn_s_memtime();                                                                   \
      r1 = __builtin_amdgcn_s_memrealtime();                                                               \
    }                                                                                                      \
    uint32_t s = 0;                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; i++) s ^= (uint32_t)__double2loint(a[i]) ^ (uint32_t)__double2hiint(a[i]); \
    if ((threadIdx.x & 63) == 0) {                                                                         \
      Rec rec;                                                                                             \
      rec.cyc = t1 - t0;                                                                                   \
      rec.rt = r1 - r0;                                                                                    \
      rec.sink = s;                                                                                        \
      rec.pad = 0;                                                                                         \
      out[blockIdx.x * (blockDim.x >> 6) + wave] = rec;                                                    \
    }                                                                                                      \
  }

// ---- 32-bit integer, plain VOP2 / VOP3
KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(and_b32, "v_and_b32 %0, %0, %1")
KERNEL(lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
KERNEL(add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL(lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(bfe_u32, "v_bfe_u32 %0, %0, 3, 9")
KERNEL(perm_b32, "v_perm_b32 %0, %0, %1, %2")
KERNEL(alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, 1")
KERNEL(alignbit_b32, "v_alignbit_b32 %0, %0, %1, 7")
KERNEL(min_u32, "v_min_u32 %0, %0, %1")
KERNEL(min3_u32, "v_min3_u32 %0, %0, %1, %2")
KERNEL(max3_i32, "v_max3_i32 %0, %0, %1, %2")
KERNEL(med3_i32, "v_med3_i32 %0, %0, %1, %2")
KERNEL(bcnt_u32, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, -1, %0")
KERNEL(cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(cmp_gt_u32, "v_cmp_gt_u32 vcc, %0, %1")
KERNEL(cmp_gt_u32_e64, "v_cmp_gt_u32 s[20:21], %0, %1")
KERNEL(mov_b32, "v_mov_b32 %0, %1")
// ---- multiplies / dot / sad
KERNEL(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL(mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL(dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
KERNEL(dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
KERNEL(dot8_u32_u4, "v_dot8_u32_u4 %0, %0, %1, %2")
KERNEL(sad_u8, "v_sad_u8 %0, %0, %1, %2")
KERNEL(msad_u8, "v_msad_u8 %0, %0, %1, %2")
KERNEL(sad_u16, "v_sad_u16 %0, %0, %1, %2")
KERNEL(sad_u32, "v_sad_u32 %0, %0, %1, %2")
KERNEL(lerp_u8, "v_lerp_u8 %0, %0, %1, %2")
// ---- SDWA / DPP
KERNEL(add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(max_u16_sdwa, "v_max_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
KERNEL(cmp_gt_u16_sdwa, "v_cmp_gt_u16_sdwa vcc, %0, %1 src0_sel:BYTE_1 src1_sel:BYTE_2")
KERNEL(add_u32_dpp_shr, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(mov_b32_dpp_bcast, "v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf")
KERNEL(mov_b32_dpp_quad, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
// ---- 16-bit and packed 16-bit
KERNEL(add_u16, "v_add_u16 %0, %0, %1")
KERNEL(min_u16, "v_min_u16 %0, %0, %1")
KERNEL(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
KERNEL(pk_min_u16, "v_pk_min_u16 %0, %0, %1")
KERNEL(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL(pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 4, %0")
KERNEL(pk_add_f16, "v_pk_add_f16 %0, %0, %1")
KERNEL(pk_min_f16, "v_pk_min_f16 %0, %0, %1")
KERNEL(pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL(pk_minimum3_f16, "v_pk_minimum3_f16 %0, %0, %1, %2")
KERNEL(pk_maximum3_f16, "v_pk_maximum3_f16 %0, %0, %1, %2")
KERNEL(min3_f16, "v_min3_f16 %0, %0, %1, %2")
// ---- f32
KERNEL(add_f32, "v_add_f32 %0, %0, %1")
KERNEL(mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL(fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL(max_f32, "v_max_f32 %0, %0, %1")
KERNEL(min3_f32, "v_min3_f32 %0, %0, %1, %2")
KERNEL(cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %0, %0")
KERNEL(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
KERNEL(cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %0, 1, %1")
KERNEL(rcp_f32, "v_rcp_f32 %0, %0")
KERNEL(exp_f32, "v_exp_f32 %0, %0")

// ---- round 3 additions: which VOP1/VOP2 ops are double rate, and which operand kinds keep them there
KERNEL(sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(subrev_u32, "v_subrev_u32 %0, %0, %1")
KERNEL(or_b32, "v_or_b32 %0, %0, %1")
KERNEL(xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL(xnor_b32, "v_xnor_b32 %0, %0, %1")
KERNEL(not_b32, "v_not_b32 %0, %0")
KERNEL(lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
KERNEL(ashrrev_i32, "v_ashrrev_i32 %0, 3, %0")
KERNEL(max_u32, "v_max_u32 %0, %0, %1")
KERNEL(min_i32, "v_min_i32 %0, %0, %1")
KERNEL(max_i32, "v_max_i32 %0, %0, %1")
KERNEL(add_co_u32, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL(addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL(add_u32_sgpr, "v_add_u32 %0, %3, %0")
KERNEL(and_b32_sgpr, "v_and_b32 %0, %3, %0")
KERNEL(add_u32_inl, "v_add_u32 %0, 7, %0")
KERNEL(and_b32_lit, "v_and_b32 %0, 0xff00ff, %0")
KERNEL(add_u32_lit, "v_add_u32 %0, 0x12345, %0")
KERNEL(add_u32_e64, "v_add_u32_e64 %0, %0, %1")
KERNEL(add_u32_e64_clamp, "v_add_u32_e64 %0, %0, %1 clamp")
KERNEL(sub_u16, "v_sub_u16 %0, %0, %1")
KERNEL(max_u16, "v_max_u16 %0, %0, %1")
KERNEL(min_i16, "v_min_i16 %0, %0, %1")
KERNEL(max_i16, "v_max_i16 %0, %0, %1")
KERNEL(mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
KERNEL(lshlrev_b16, "v_lshlrev_b16 %0, 3, %0")
KERNEL(lshrrev_b16, "v_lshrrev_b16 %0, 3, %0")
KERNEL(mad_u16, "v_mad_u16 %0, %0, %1, %2")
KERNEL(add_f16, "v_add_f16 %0, %0, %1")
KERNEL(sub_f16, "v_sub_f16 %0, %0, %1")
KERNEL(mul_f16, "v_mul_f16 %0, %0, %1")
KERNEL(max_f16, "v_max_f16 %0, %0, %1")
KERNEL(min_f16, "v_min_f16 %0, %0, %1")
KERNEL(fma_f16, "v_fma_f16 %0, %0, %1, %2")
KERNEL(sub_f32, "v_sub_f32 %0, %0, %1")
KERNEL(subrev_f32, "v_subrev_f32 %0, %0, %1")
KERNEL(min_f32, "v_min_f32 %0, %0, %1")
KERNEL(mac_f32_sgpr, "v_fmac_f32 %0, %3, %1")
KERNEL(fma_f32_sgpr, "v_fma_f32 %0, %0, %3, %2")
KERNEL(mad_mix, "v_fma_mix_f32 %0, %0, %1, %2")
KERNEL(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
KERNEL(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
KERNEL(cvt_f16_f32, "v_cvt_f16_f32 %0, %0")
KERNEL(cvt_pkrtz_f16_f32, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
KERNEL(cmp_gt_f32, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL(cmp_gt_u16, "v_cmp_gt_u16 vcc, %0, %1")
KERNEL(cmp_lt_i32_e64, "v_cmp_lt_i32 s[20:21], %0, %1")
KERNEL(cmp_class_f32, "v_cmp_class_f32 vcc, %0, %1")
KERNEL(readlane_like_mov, "v_mov_b32 %0, %3")
KERNEL(sad_u8_sgpr, "v_sad_u8 %0, %0, %3, %2")
KERNEL(add_f32_dpp, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(add_f32_sdwa, "v_add_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD")
KERNEL(mov_b32_sdwa, "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2")
KERNEL(and_b32_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(min_u16_sdwa, "v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
KERNEL(min_u16_sdwa_w, "v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0")
// ---- 64-bit destinations
KERNEL64(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL64(pk_mov_b32, "v_pk_mov_b32 %0, %1, %2")
KERNEL64(add_f64, "v_add_f64 %0, %0, %1")
KERNEL64(fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")

typedef void (*KFn)(Rec*, int, uint32_t, int);
struct Entry {
  const char* name;
  KFn fn;
};
#define E(NAME) {#NAME, k_##NAME}
static Entry entries[] = {
    E(add_u32), E(and_b32), E(lshlrev_b32), E(add3_u32), E(lshl_add_u32), E(and_or_b32), E(bfe_u32), E(perm_b32),
    E(alignbyte_b32), E(alignbit_b32), E(min_u32), E(min3_u32), E(max3_i32), E(med3_i32), E(bcnt_u32), E(mbcnt_lo),
    E(cndmask_b32), E(cmp_gt_u32), E(cmp_gt_u32_e64), E(mov_b32), E(mul_u32_u24), E(mad_u32_u24), E(mul_lo_u32),
    E(mul_hi_u32), E(dot4_u32_u8), E(dot2_u32_u16), E(dot8_u32_u4), E(sad_u8), E(msad_u8), E(sad_u16), E(sad_u32),
    E(lerp_u8), E(add_u32_sdwa), E(max_u16_sdwa), E(cmp_gt_u16_sdwa), E(add_u32_dpp_shr), E(mov_b32_dpp_bcast),
    E(mov_b32_dpp_quad), E(add_u16), E(min_u16), E(pk_add_u16), E(pk_sub_i16), E(pk_min_u16), E(pk_max_i16),
    E(pk_mul_lo_u16), E(pk_mad_u16), E(pk_lshrrev_b16), E(pk_add_f16), E(pk_min_f16), E(pk_fma_f16),
    E(pk_minimum3_f16), E(pk_maximum3_f16), E(min3_f16), E(add_f32), E(mul_f32), E(fma_f32), E(fmac_f32), E(max_f32),
    E(min3_f32), E(cvt_f32_ubyte1), E(cvt_f32_u32), E(cvt_pk_u8_f32), E(rcp_f32), E(exp_f32), E(pk_fma_f32),
    E(pk_add_f32), E(pk_mul_f32), E(pk_mov_b32), E(add_f64), E(fma_f64), E(mul_f64), E(lshlrev_b64), E(sub_u32), E(subrev_u32), E(or_b32), E(xor_b32), E(xnor_b32), E(not_b32), E(lshrrev_b32), E(ashrrev_i32), E(max_u32), E(min_i32), E(max_i32), E(add_co_u32), E(addc_co_u32), E(add_u32_sgpr), E(and_b32_sgpr), E(add_u32_inl), E(and_b32_lit), E(add_u32_lit), E(add_u32_e64), E(add_u32_e64_clamp), E(sub_u16), E(max_u16), E(min_i16), E(max_i16), E(mul_lo_u16), E(lshlrev_b16), E(lshrrev_b16), E(mad_u16), E(add_f16), E(sub_f16), E(mul_f16), E(max_f16), E(min_f16), E(fma_f16), E(sub_f32), E(subrev_f32), E(min_f32), E(mac_f32_sgpr), E(fma_f32_sgpr), E(mad_mix), E(cvt_f32_i32), E(cvt_u32_f32), E(cvt_f32_ubyte0), E(cvt_f16_f32), E(cvt_pkrtz_f16_f32), E(cmp_gt_f32), E(cmp_gt_u16), E(cmp_lt_i32_e64), E(cmp_class_f32), E(readlane_like_mov), E(sad_u8_sgpr), E(add_f32_dpp), E(add_f32_sdwa), E(mov_b32_sdwa), E(and_b32_sdwa), E(min_u16_sdwa), E(min_u16_sdwa_w),
};

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  bool json = false, sat_only = false;
  const char* only = nullptr;  // comma-separated list of instruction names
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--json")) json = true;
    else if (!strcmp(argv[i], "--sat")) sat_only = true;  // only the saturated launch (for rocprofv3 --pmc passes)
    else if (!strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  Rec* d;
  const int maxWaves = 4096 * 16;
  CK(hipMalloc(&d, sizeof(Rec) * maxWaves));
  std::vector<Rec> h(maxWaves);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters1 = 2000, iters8 = 1000;
  const double per_iter = 128.0;
  if (!json)
    printf("# %s, %d CUs, clockRate %d kHz\n# %-20s %10s %10s %12s %10s %10s\n", prop.gcnArchName, cus, prop.clockRate,
           "instruction", "w1 cyc", "w2 cyc/ins", "w8 cyc/SIMD", "w8 GHz", "w8 Ginst/s");
  else
    printf("{\"device\": \"%s\", \"cus\": %d, \"rows\": [\n", prop.gcnArchName, cus);
  bool first = true;
  for (const Entry& en : entries) {
    if (only) {
      const size_t L = strlen(en.name);
      bool hit = false;
      for (const char* q = only; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr)
        if (!strncmp(q, en.name, L) && (q[L] == ',' || q[L] == 0)) hit = true;
      if (!hit) continue;
    }
    // warm
    hipLaunchKernelGGL(en.fn, dim3(cus), dim3(256), 0, 0, d, 20, 1u, 0xffff);
    CK(hipDeviceSynchronize());
    double w1 = 0, w2 = 0;
    if (!sat_only) {
      // w1: one wave
      hipLaunchKernelGGL(en.fn, dim3(1), dim3(64), 0, 0, d, iters1, 1u, 0xffff);
      CK(hipMemcpy(h.data(), d, sizeof(Rec), hipMemcpyDeviceToHost));
      w1 = (double)h[0].cyc / (iters1 * per_iter);
      // w2: two waves on one SIMD (waves 0 and 4 of a 512-thread block)
      hipLaunchKernelGGL(en.fn, dim3(1), dim3(512), 0, 0, d, iters1, 1u, 0x11);
      CK(hipMemcpy(h.data(), d, sizeof(Rec) * 8, hipMemcpyDeviceToHost));
      w2 = (double)(h[0].cyc > h[4].cyc ? h[0].cyc : h[4].cyc) / (2.0 * iters1 * per_iter);
    }
    // w8: 8 waves per SIMD on every CU
    const int blocks = cus * 32;  // 256-thread blocks: one wave per SIMD each, 8 resident per CU, 4 rounds
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, d, iters8, 1u, 0xffff);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d, sizeof(Rec) * blocks * 4, hipMemcpyDeviceToHost));
    double clk = 0;
    unsigned long long maxcyc = 0;
    for (int i = 0; i < blocks * 4; i++) {
      clk += (double)h[i].cyc / ((double)h[i].rt * 10.0);  // cycles per ns (s_memrealtime = 100 MHz)
      if (h[i].cyc > maxcyc) maxcyc = h[i].cyc;
    }
    clk /= blocks * 4;
    const double insts = (double)blocks * 4 * iters8 * per_iter;
    const double ginst = insts / (ms * 1e6);
    // cycles one SIMD spends per instruction: in-kernel cycles of the slowest wave / instructions issued on its SIMD
    const double w8 = (double)maxcyc / (8.0 * iters8 * per_iter);
    const double w8_wall = (ms * 1e6 * clk) / (insts / (cus * 4.0));
    if (!json)
      printf("  %-20s %10.2f %10.2f %12.2f %10.3f %10.1f   (wall-clock based: %.2f)\n", en.name, w1, w2, w8, clk, ginst,
             w8_wall);
    else
      printf("%s {\"inst\": \"%s\", \"w1\": %.3f, \"w2\": %.3f, \"w8\": %.3f, \"w8_wall\": %.3f, \"ghz\": %.3f, \"ginst_s\": %.1f}",
             first ? "" : ",\n", en.name, w1, w2, w8, w8_wall, clk, ginst);
    first = false;
    fflush(stdout);
  }
  if (json) printf("\n]}\n");
  return 0;
}
