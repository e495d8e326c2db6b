// orbx_detect.hip — FAST-9-16 detection kernels of the ORB front-end (gfx950 / CDNA4, wave64).
//
// ORBextractor::ComputeKeyPointsOctTree's cell loop (src/ORBextractor.cc:892-971 of the reference): per-cell cv::FAST with
// non-max suppression at iniThFAST, minThFAST for cells that come out empty.  SURVEY A3 / B3.
#include "orbx_device.h"

namespace orbx {

// ================================================================================================ detect
// FAST-9-16 on an LDS tile, in two stages (k_detect).
//
// Layout: the cell ROI (cell + 3 px FAST halo each side) sits in LDS with ROI column 0 on a dword boundary
// (the loader funnel-shifts the unaligned global row).  Stage 1: a lane owns a "quad" of 4 detectable pixels and reads
// the three rows that hold ring pixels 0 / 4 / 8 / 12 of the quad (7 dwords, kept in registers) for the compass pre-test
// on two packed pixel pairs (compass_pair); survivors go to an LDS list.
// Stage 2: dense lanes, two survivors per lane in packed f16: the exact contrast M from the 16 ring pixels (8-windows of
// minima / maxima shared by the two arcs that contain them, 36 packed operations per polarity); corner iff M > t, score
// M - 1.  Then list-based 3x3 NMS.
constexpr int kRingDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
constexpr int kRingDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
#ifndef ORBX_ROW_ROUNDS
#define ORBX_ROW_ROUNDS 1
#endif
constexpr int kListTotal = 704;  // u16 entries of k_detect's one LDS list: corners [0, nList), then compass survivors [nList, sEnd)
constexpr int kDetectXcdRun = 8;  // cells per XCD run of k_detect's block order (1 = plain order fetched 2.7x the bytes)
constexpr int kLoadRows = 12;   // rows per lane the tile loader of k_detect keeps in flight
#ifndef ORBX_DETECT_WIDE_LOAD
#define ORBX_DETECT_WIDE_LOAD 1
#endif
constexpr bool kDetectWideLoad = ORBX_DETECT_WIDE_LOAD != 0;   // unaligned 16 / 8 / 4-byte tile loads for the compile-time pitches
struct __attribute__((packed, aligned(1))) U128Unaligned { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 load_u128_unaligned(const uint8_t* p) {  // one global_load_dwordx4 at any byte address (HSA runs
  const U128Unaligned t = *reinterpret_cast<const U128Unaligned*>(p);     // the memory pipeline in unaligned-access mode)
  return make_uint4(t.x, t.y, t.z, t.w);
}
struct __attribute__((packed, aligned(1))) U64Unaligned { uint32_t x, y; };
struct __attribute__((packed, aligned(1))) U32Unaligned { uint32_t x; };
__device__ __forceinline__ uint2 load_u64_unaligned(const uint8_t* p) {
  const U64Unaligned t = *reinterpret_cast<const U64Unaligned*>(p);
  return make_uint2(t.x, t.y);
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) { return reinterpret_cast<const U32Unaligned*>(p)->x; }

// FAST contrast of a pixel: M = max over the 16 nine-pixel arcs of the arc's minimum one-signed contrast
// = max( max_s min(arc_s) - c , c - min_s max(arc_s) ).  Sliding 9-windows are built from 3-windows: 16 + 16 + 8
// three-input operations per polarity.  The pixel is a corner at threshold t iff M > t, and its cornerScore is M - 1
//  -- one pass gives both the decision and the score.
// Two pixels per lane in packed half precision.  A pixel value v (0..255) is used
// as the f16 BIT PATTERN v, i.e. the subnormal v * 2^-24 (kernels run with f16 denormals preserved,
// .amdhsa_float_denorm_mode_16_64 3): order preserving, and sums / differences of such values (|d| <= 255) are exact
// multiples of 2^-24, so v_pk_minimum3_f16 / v_pk_maximum3_f16 / v_pk_add_f16 give bit-exact integer results for two
// pixels at the cost of one, and the two bytes are packed by a single v_perm (no bias to OR in).  Returns M for
// pixel A in .x and pixel B in .y, each as the bit pattern of |M| with the f16 sign bit for M < 0.
typedef _Float16 orbx_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ orbx_h2 pk_min3(orbx_h2 a, orbx_h2 b, orbx_h2 c) {
  return __builtin_elementwise_minimum(__builtin_elementwise_minimum(a, b), c);
}
__device__ __forceinline__ orbx_h2 pk_max3(orbx_h2 a, orbx_h2 b, orbx_h2 c) {
  return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}
// Necessary condition for a 9-arc: two cyclically adjacent compass pixels (ring 0, 4, 8, 12) of the same polarity, i.e. one of
// {0, 8} and one of {4, 12} (in a 4-cycle every even position is adjacent to every odd one): min(max(v0, v8), max(v4, v12)) > c + t,
// likewise below c - t.  Evaluated on TWO horizontally adjacent pixels (P, P + 1) in packed f16 (round 4; rounds 1-3 ran it per pixel
// on SDWA byte operands, 20 instructions per pair): the five operands of
// the pair -- centre, ring 0 / 8 (rows 6 / 0, same columns), ring 4 / 12 (row 3, columns +3 / -3) -- are spread from the row
// dwords by one v_perm each ([b, 0, b', 0] = two f16 subnormal patterns), then 6 packed min / max, two packed subtractions, one
// packed max and two compares: 16 instructions per pair instead of 20.  M' = max(hiMin - c, c - loMax) > t, exact like stage 2.
template <int P>
__device__ __forceinline__ void compass_pair(const uint32_t (&r)[7][3], orbx_h2 th2, uint64_t& m0, uint64_t& m1) {
  auto spread = [&](int row, int b) {  // bytes b, b + 1 of the 12-byte row -> (f16 pattern, f16 pattern)
    const int d = (b + 1 <= 7) ? 0 : 1, i = b - 4 * d;
    const uint32_t sel = 0x0c000c00u | (uint32_t)i | ((uint32_t)(i + 1) << 16);
    return __builtin_bit_cast(orbx_h2, __builtin_amdgcn_perm(r[row][d + 1], r[row][d], sel));
  };
  const orbx_h2 c = spread(3, 3 + P), v0 = spread(6, 3 + P), v8 = spread(0, 3 + P), v4 = spread(3, 6 + P), v12 = spread(3, P);
  const orbx_h2 hiMin = __builtin_elementwise_minimum(__builtin_elementwise_maximum(v0, v8), __builtin_elementwise_maximum(v4, v12));
  const orbx_h2 loMax = __builtin_elementwise_maximum(__builtin_elementwise_minimum(v0, v8), __builtin_elementwise_minimum(v4, v12));
  const orbx_h2 M = __builtin_elementwise_maximum(hiMin - c, c - loMax);
  m0 = __ballot(M.x > th2.x);
  m1 = __ballot(M.y > th2.y);
}

// EXPERIMENT (round 5, -DORBX_EVEN_FILTER=1; measured and rejected): a second necessary test on the dense survivor
// list, in front of the 16-pixel contrast pass.  A 9-arc covers at least four CONSECUTIVE even ring positions (0, 2, .., 14), so
// a corner has four cyclically consecutive even-ring pixels all > c + t or all < c - t: with e[q] = ring pixel 2q,
// hi = max_q min(e[q], e[q+1], e[q+2], e[q+3]), lo = min_q max(...), the pixel can only be a corner if max(hi - c, c - lo) > t.
// Two survivors per lane in packed f16 like the contrast pass: 9 byte reads + 9 v_perm per pair, 8 + 8 + 3 packed operations per
// polarity, 3 + 2 to decide.
#ifndef ORBX_D16_PAIRS
#define ORBX_D16_PAIRS 0   // measured: bit-exact, 17 v_perm -> 17 v_or per pass, and NO faster (216.5 vs 218.4 us: one wait for all 34 loads loses the overlap the compiler's staged waits had); profiles/r5_d16_pairs_ab.txt
#endif
#ifndef ORBX_EVEN_FILTER
#define ORBX_EVEN_FILTER 0
#endif
// TIMING-ONLY ablations (wrong results; tools/build_variant.sh): 1 no tile load, 2 no stage 1 (no survivors), 4 no stage 2 (survivors
// dropped), 8 no NMS / emit, 16 return behind the scalar front
#ifndef DET_ABLATE
#define DET_ABLATE 0
#endif
__device__ __forceinline__ orbx_h2 even_ring_contrast2_lds(const uint8_t* a8, const uint8_t* b8, int TP) {
  orbx_h2 e[8];
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int off = (kRingDY[2 * q] + 3) * TP + kRingDX[2 * q] + 3;
    orbx_us2 v;
    v.x = a8[off];
    v.y = b8[off];
    e[q] = __builtin_bit_cast(orbx_h2, v);
  }
  orbx_h2 hi, lo;
  {
    orbx_h2 p2[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; q++) p2[q] = __builtin_elementwise_minimum(e[q], e[(q + 1) & 7]);
#pragma unroll
    for (int q = 0; q < 8; q++) w[q] = __builtin_elementwise_minimum(p2[q], p2[(q + 2) & 7]);
    hi = pk_max3(w[0], w[1], w[2]);
    hi = pk_max3(hi, w[3], w[4]);
    hi = pk_max3(hi, w[5], w[6]);
    hi = __builtin_elementwise_maximum(hi, w[7]);
  }
  {
    orbx_h2 p2[8], w[8];
#pragma unroll
    for (int q = 0; q < 8; q++) p2[q] = __builtin_elementwise_maximum(e[q], e[(q + 1) & 7]);
#pragma unroll
    for (int q = 0; q < 8; q++) w[q] = __builtin_elementwise_maximum(p2[q], p2[(q + 2) & 7]);
    lo = pk_min3(w[0], w[1], w[2]);
    lo = pk_min3(lo, w[3], w[4]);
    lo = pk_min3(lo, w[5], w[6]);
    lo = __builtin_elementwise_minimum(lo, w[7]);
  }
  orbx_us2 cv;
  cv.x = a8[3 * TP + 3];
  cv.y = b8[3 * TP + 3];
  const orbx_h2 c = __builtin_bit_cast(orbx_h2, cv);
  return __builtin_elementwise_maximum(hi - c, c - lo);
}

// a8 / b8 point at the TOP-LEFT corner of each pixel's 7x7 window, so every ring offset is a non-negative ds_read immediate.
typedef unsigned short orbx_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ orbx_h2 fast_contrast_network(const orbx_h2 (&r)[16], orbx_h2 c) {
  // max over the 16 arcs of the arc's minimum, 36 packed operations per polarity instead of 40 (round 4): for even k the arcs
  // starting at k and k + 1 share the 8-window W = r[k+1 .. k+8], and max(min(W, r[k]), min(W, r[k+9])) = min(W, max(r[k], r[k+9])):
  // pair minima p (8), 4-windows w4 (8), e = max of the two end points (8), f = min3(w4[j], w4[j+2], e[j]) (8), max over f (4).
  // One polarity after the other (the scheduling barrier keeps them apart): the kernel's register peak is here.
  orbx_h2 maxmin, minmax;
  {
    orbx_h2 pr[8], w4[8], f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) pr[q] = __builtin_elementwise_minimum(r[(2 * q + 1) & 15], r[(2 * q + 2) & 15]);
#pragma unroll
    for (int q = 0; q < 8; q++) w4[q] = __builtin_elementwise_minimum(pr[q], pr[(q + 1) & 7]);
#pragma unroll
    for (int q = 0; q < 8; q++)
      f[q] = pk_min3(w4[q], w4[(q + 2) & 7], __builtin_elementwise_maximum(r[2 * q], r[(2 * q + 9) & 15]));
    maxmin = pk_max3(f[0], f[1], f[2]);
    maxmin = pk_max3(maxmin, f[3], f[4]);
    maxmin = pk_max3(maxmin, f[5], f[6]);
    maxmin = __builtin_elementwise_maximum(maxmin, f[7]);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    orbx_h2 pr[8], w4[8], f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) pr[q] = __builtin_elementwise_maximum(r[(2 * q + 1) & 15], r[(2 * q + 2) & 15]);
#pragma unroll
    for (int q = 0; q < 8; q++) w4[q] = __builtin_elementwise_maximum(pr[q], pr[(q + 1) & 7]);
#pragma unroll
    for (int q = 0; q < 8; q++)
      f[q] = pk_max3(w4[q], w4[(q + 2) & 7], __builtin_elementwise_minimum(r[2 * q], r[(2 * q + 9) & 15]));
    minmax = pk_min3(f[0], f[1], f[2]);
    minmax = pk_min3(minmax, f[3], f[4]);
    minmax = pk_min3(minmax, f[5], f[6]);
    minmax = __builtin_elementwise_minimum(minmax, f[7]);
  }
  return __builtin_elementwise_maximum(maxmin - c, c - minmax);
}

__device__ __forceinline__ orbx_h2 fast_contrast2_lds(const uint8_t* a8, const uint8_t* b8, int TP) {
  orbx_h2 r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int off = (kRingDY[k] + 3) * TP + kRingDX[k] + 3;
    orbx_us2 v;
    v.x = a8[off];
    v.y = b8[off];  // ds_read_u8_d16_hi: the pair is packed by the loads
    r[k] = __builtin_bit_cast(orbx_h2, v);
  }
  orbx_us2 cv;
  cv.x = a8[3 * TP + 3];
  cv.y = b8[3 * TP + 3];
  return fast_contrast_network(r, __builtin_bit_cast(orbx_h2, cv));
}

// The same contrast for a pair of DIFFERENT survivors with the byte pairs packed by the LOADS (round 5).  On gfx950 a d16 load
// zeroes the other half of its destination (tools/ubench/d16_probe.hip: ds_read_u8_d16 -> 0x000000bb, ds_read_u8_d16_hi ->
// 0x00bb0000), which is why the compiler will not merge two of them into one register -- but it also means that the pair is
// lo | hi: a 2-cycle v_or_b32 instead of the 4-cycle v_perm_b32 the compiler's ds_read_u8 pairs need (17 per pass; VALU issue is
// this kernel's limit).  The loads are inline asm (the compiler never selects the d16 forms on an SRAM-ECC target):
// it does not count them, so all 34 are issued, ONE s_waitcnt follows, and the registers only become visible to later code
// through the "+v" ties behind it.  Compile-time pitch only (the offsets are ds_read immediates).
template <int OFF>
__device__ __forceinline__ void lds_byte_pair_d16(uint32_t aA, uint32_t aB, uint32_t& lo, uint32_t& hi) {
  asm volatile("ds_read_u8_d16 %0, %2 offset:%4\n\tds_read_u8_d16_hi %1, %3 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(aA), "v"(aB), "n"(OFF));
}
template <int TP, int K>
__device__ __forceinline__ void ring_pairs_d16(uint32_t aA, uint32_t aB, uint32_t (&lo)[17], uint32_t (&hi)[17]) {
  if constexpr (K < 16) {
    lds_byte_pair_d16<(kRingDY[K] + 3) * TP + kRingDX[K] + 3>(aA, aB, lo[K], hi[K]);
    ring_pairs_d16<TP, K + 1>(aA, aB, lo, hi);
  } else {
    lds_byte_pair_d16<3 * TP + 3>(aA, aB, lo[16], hi[16]);   // the centre
  }
}
template <int TP>
__device__ __forceinline__ orbx_h2 fast_contrast2_lds_d16(const uint8_t* a8, const uint8_t* b8) {
  uint32_t lo[17], hi[17];
  ring_pairs_d16<TP, 0>((uint32_t)(uintptr_t)a8, (uint32_t)(uintptr_t)b8, lo, hi);
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]), "+v"(lo[8]),
                 "+v"(lo[9]), "+v"(lo[10]), "+v"(lo[11]), "+v"(lo[12]), "+v"(lo[13]), "+v"(lo[14]));
  asm volatile("" : "+v"(lo[15]), "+v"(lo[16]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]),
                     "+v"(hi[6]), "+v"(hi[7]), "+v"(hi[8]), "+v"(hi[9]), "+v"(hi[10]), "+v"(hi[11]), "+v"(hi[12]));
  asm volatile("" : "+v"(hi[13]), "+v"(hi[14]), "+v"(hi[15]), "+v"(hi[16]));
  orbx_h2 r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = __builtin_bit_cast(orbx_h2, lo[k] | hi[k]);
  const orbx_h2 c = __builtin_bit_cast(orbx_h2, lo[16] | hi[16]);
  return fast_contrast_network(r, c);
}

// wave mask of the lanes below n (n <= 0: none, n >= 64: all) -- scalar ALU only
__device__ __forceinline__ uint64_t low_lanes(int n) {
  return n > 0 ? ~0ull >> (64 - min(n, 64)) : 0ull;
}
__device__ __forceinline__ uint64_t low_lanes_pos(int n) {  // n >= 1
  return ~0ull >> (64 - min(n, 64));
}


// One wave per FAST cell (workgroup = 64 threads, so __syncthreads() is a wave barrier).
// Pass 1 runs at iniThFAST; only a cell whose post-NMS set is empty is redone at minThFAST (:942-959).
// The NMS needs no threshold masking: a neighbour that is not a corner at t has score < t <= the centre's.
// Output: no atomics.  Every cell owns cellCap slots of the sparse store (an NMS survivor set has at most
// ceil(w/2)*ceil(h/2) members) and writes its count; k_octree scans the counts and compacts.
// TAP: the test tap of orbx_debug_score_map, a separate instantiation (its registers cost the product kernel 3.5 %).
// TPC: compile-time LDS pitch of the image tile and of the score tile (the same), 0 = run-time pitch.  With a constant
// pitch the ring offsets of the contrast pass, the NMS neighbours and the three rows of a stage-1 quad fold into the
// ds_read immediate offsets instead of costing one v_add each.  The pitch must stay the tight one of the geometry:
// padding it to 64 made the kernel 6 % SLOWER.  Instantiated for the pitches of the usual cell widths (33..48 px).
template <bool TAP, int TPC>
__global__ __launch_bounds__(64) void k_detect(Geom g, Pyr p, uint32_t* __restrict__ cellCand,
                                               int* __restrict__ cellCount, int listCap, int cellBegin,
                                               uint8_t* __restrict__ dbgScore) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x;
  const int img = blockIdx.y;
#ifdef DET_CLK  // shader-clock measurement aid: cycles per 10 ns tick over the life of a few waves
  const long long clk0 = clock64(), wall0 = wall_clock64();
  struct ClkPrint {
    long long c0, w0;
    int on;
    __device__ ~ClkPrint() {
      if (on) {
        const long long dc = clock64() - c0, dw = wall_clock64() - w0;
        printf("det wave: %lld shader cycles in %lld x 10 ns -> %.3f GHz\n", dc, dw, dw ? (double)dc / (double)dw / 10.0 : 0.0);
      }
    }
  } clkPrint{clk0, wall0, (int)(lane == 0 && blockIdx.y == 31 && (blockIdx.x % 400) == 7)};
#endif
#ifdef DET_PROF  // measurement aid (make prof PROF_FLAGS=-DDET_PROF): the life of a few cell-waves inside one launch, x10 ns
  long long dq_[8]; int ndq_ = 0;
#define DET_MK() do { if (ndq_ < 8) dq_[ndq_++] = wall_clock64(); } while (0)
#else
#define DET_MK() do {} while (0)
#endif
  DET_MK();
  // Runs of xcdRun horizontally consecutive cells share an XCD and therefore the L2 lines of their common halo
  // columns (HBM-side fetch 410 -> 151 MB per 64-image launch; same duration, the kernel is VALU-bound).
  int cell = cellBegin + xcd_run_remap<kDetectXcdRun>(blockIdx.x, gridDim.x, blockIdx.y);
  int l = 0;  // the scalar ALU is nearly as busy as the vector ALU in this kernel: no search loop, no integer divisions
#pragma unroll
  for (int q = 1; q < ORBX_MAX_LEVELS; q++) l -= (g.levelCell[q] - 1 - cell) >> 31;   // += (cell >= levelCell[q]), on the scalar ALU
  // (written as a select the compiler went through v_cndmask + v_readfirstlane for every level)
  const LevelDev L = g.lv[l];
  int* myCount = cellCount + (long long)img * g.totalCells + cell;
  cell -= L.cellStart;
  // cell / nCols by a 1-ulp reciprocal: (cell + 0.5) / nCols stays 0.5 / nCols away from the next integer
  const int ci = __builtin_amdgcn_readfirstlane((int)(((float)cell + 0.5f) * __builtin_amdgcn_rcpf((float)L.nCols))),
            cj = cell - ci * L.nCols;
  const int maxBX = L.w - kBorder, maxBY = L.h - kBorder;
  const int iniY = kBorder + ci * L.hCell, iniX = kBorder + cj * L.wCell;
  const int maxY = min(iniY + L.hCell + 6, maxBY), maxX = min(iniX + L.wCell + 6, maxBX);
  const int rw = maxX - iniX, rh = maxY - iniY;
  const int dw = rw - 6, dh = rh - 6;  // detectable window of the cell (FAST needs a 3 px ring)
  if (iniY >= maxBY - 3 || iniX >= maxBX - 6 || dw <= 0 || dh <= 0) {  // src/ORBextractor.cc:913,919
    if (lane == 0) *myCount = 0;
    return;
  }

  const int TP = TPC ? TPC : g.tileP, SPB = TP;  // pitch of the image tile and of the score tile, bytes (tileP == scoreP)
  constexpr bool kRowRounds = ORBX_ROW_ROUNDS && (TPC == 44 || TPC == 48 || TPC == 56);
  const int TPd = TP >> 2, SPd = SPB >> 2;                          // and in dwords
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem);
  uint32_t* score = tile + ((TPd * g.tileH + 3) & ~3);  // 16-byte aligned: cleared with b128 stores
  uint8_t* score8 = reinterpret_cast<uint8_t*>(score);
  // One list of positions (y << 8 | x): the cell's corners so far in front, the survivors waiting for the contrast pass
  // behind them.  A flush turns survivors into (fewer) corners in place -- entry k of the survivors is read before corner
  // k is written -- so the 704 entries that used to be 256 corners + 448 survivors now hold up to 448 corners (busy cells
  // of the benchmark frames exceed 256 and paid for the tile-scan NMS) or ~650 survivors.  A round appends at most 256
  // survivors, hence the corner limit of total - 256.  Measured: 480 entries (32 cells per CU instead of 29) 270 us, 704
  // entries 245 us, 832 entries 246 us, 1024 entries (26 cells) 261 us.
  uint16_t* list = reinterpret_cast<uint16_t*>(score + SPd * g.scoreH);
  const int listTotal = min(listCap, kListTotal), cornerCap = listTotal - 256;
  const uint8_t* tile8 = reinterpret_cast<const uint8_t*>(tile);
  const int qpr = (dw + 3) >> 2;  // quads per detect row
  const int nq = qpr * dh;
  // 1-ulp reciprocals suffice: (lane + 0.5) / qpr and 64.5 / qpr stay 0.5 / qpr away from the next integer (an IEEE division is 12
  // instructions, an integer division ~ 30)
  const float inv_qpr = __builtin_amdgcn_rcpf((float)qpr);
  const float inv_tp = __builtin_amdgcn_rcpf((float)TP);

  int pitch;
  const uint8_t* im = level_ptr(g, p, img, l, pitch);
  if ((DET_ABLATE & 16) && dw != 4095) { if (lane == 0) *myCount = 0; return; }
  bool tileDone = (DET_ABLATE & 1) && dw != 4095;
  if (TPC != 0 && kDetectWideLoad) {
    // Fast loader for the compile-time tile pitches: a row is TPC / P pieces of P = 16, 8 or 4 bytes (the largest power of two that
    // divides the pitch: 48 -> 3 x 16, 56 -> 7 x 8, 44 / 52 -> 11 / 13 x 4); lane = (row, piece), one UNALIGNED global load
    // straight from the ROI's first byte and one aligned LDS store per piece -- no funnel shift, no second dword, a third (16-byte
    // pieces) to two thirds (dwords) of the dword loader's trips.  Only when all TPC bytes of a row lie inside the image row
    // (every cell but the last column of a level); bytes past the ROI width are never consumed (masked lanes).
    constexpr int P = (TPC % 16 == 0) ? 16 : (TPC % 8 == 0) ? 8 : 4, PPR = TPC ? TPC / P : 1;
    constexpr uint32_t kInv = (65536u + PPR - 1) / PPR;   // it / PPR == (it * kInv) >> 16 for it < 4096 (checked below)
    static_assert(PPR * ((4095u * kInv) >> 16) <= 4095u && (4095u / PPR) == ((4095u * kInv) >> 16), "reciprocal division");
    if (iniX + TPC <= L.w) {
      const uint8_t* rowBase = im + (long long)iniY * pitch + iniX;   // wave-uniform
      const int nItems = rh * PPR;                                    // <= 78 rows x 13 pieces
      constexpr int kB = 3;   // loads in flight per lane and batch
      for (int base = 0; base < nItems; base += 64 * kB) {
        uint4 v[kB];
        int dst[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) {
          const int it = min(base + lane + 64 * u, nItems - 1);   // the tail re-writes the last item
          const int r = (int)(((uint32_t)it * kInv) >> 16), c = it - PPR * r;
          const uint8_t* src = rowBase + (uint32_t)(__mul24(r, pitch) + P * c);
          if (P == 16) v[u] = load_u128_unaligned(src);
          else if (P == 8) { const uint2 t = load_u64_unaligned(src); v[u] = make_uint4(t.x, t.y, 0, 0); }
          else v[u] = make_uint4(load_u32_unaligned(src), 0, 0, 0);
          dst[u] = r * TPC + P * c;
        }
#pragma unroll
        for (int u = 0; u < kB; u++) {
          if (P == 16) *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
          else if (P == 8) *reinterpret_cast<uint2*>(smem + dst[u]) = make_uint2(v[u].x, v[u].y);
          else *reinterpret_cast<uint32_t*>(smem + dst[u]) = v[u].x;
        }
      }
      tileDone = true;
    }
  }
  if (!tileDone)
  {  // tile load: ROI column 0 -> LDS byte 0 of the row (funnel shift of two aligned global dwords)
    // lane = (row phase, dword column): 16 columns x 4 rows per pass, no index divisions in the loop.  All of a lane's
    // rows (kLoadRows per batch = 48 tile rows) are requested before the first one is consumed: one memory latency
    // per cell instead of one per row pass.
    const int mis = iniX & 3, xa = iniX - mis;
    const int dpr = (rw + 3) >> 2;
    const int r0 = lane >> 4;
    const uint8_t* rowBase = im + (long long)iniY * pitch + xa;  // wave-uniform: the loads take it as their scalar base
    const uint32_t pitch4 = 4u * (uint32_t)pitch;
    for (int cc = lane & 15; cc < dpr; cc += 16) {  // one trip unless the cell is wider than 58 px (tiny levels)
      const int gx = xa + 4 * cc;
      const uint8_t* src0 = rowBase + 4 * cc;
      if (gx + 8 <= L.w) {
        // rows past the ROI are clamped to its last row on both sides (load and store): the same bytes land on the
        // same LDS dword again, and the loop body needs no predicate
        const uint32_t offLast = (uint32_t)__mul24(rh - 1, pitch) + 4u * (uint32_t)cc;
        const int idxLast = 4 * (__mul24(rh - 1, TPd) + cc);  // LDS byte offsets
        for (int rb = r0; rb < rh; rb += 4 * kLoadRows) {
          const uint32_t off0 = (uint32_t)__mul24(rb, pitch) + 4u * (uint32_t)cc;
          const int idx0 = 4 * (__mul24(rb, TPd) + cc);
          uint32_t lo[kLoadRows], hi[kLoadRows];
#pragma unroll
          for (int u = 0; u < kLoadRows; u++) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(rowBase + min(off0 + (uint32_t)u * pitch4, offLast));
            lo[u] = src[0];
            hi[u] = src[1];
          }
#pragma unroll
          for (int u = 0; u < kLoadRows; u++)
            *reinterpret_cast<uint32_t*>(smem + min(idx0 + u * 4 * TP, idxLast)) = __builtin_amdgcn_alignbyte(hi[u], lo[u], mis);
        }
      } else {  // last dword columns of the level: byte loads inside the image
        for (int r = r0; r < rh; r += 4) {
          const uint8_t* src = src0 + __mul24(r, pitch);
          uint64_t v = 0;
          for (int k = 0; k < 8; k++)
            if (gx + k < L.w) v |= (uint64_t)src[k] << (8 * k);
          tile[__mul24(r, TPd) + cc] = __builtin_amdgcn_alignbyte((uint32_t)(v >> 32), (uint32_t)v, mis);
        }
      }
    }
  }
  DET_MK();
  uint32_t* out = cellCand + (long long)img * g.cellImg + L.cellOff + (long long)cell * L.cellCap;
  int kept = 0;
  // a round of 64 quads advances a lane by dq rows and rq quads (no division per round)
  const int dq = __builtin_amdgcn_readfirstlane((int)(64.5f * inv_qpr)), rq = 64 - dq * qpr;
  const int vlast = dw - 4 * (qpr - 1);  // pixels of a row's last quad inside the detectable window (1..4)
  const uint64_t keep1 = vlast > 1 ? ~0ull : 0ull, keep2 = vlast > 2 ? ~0ull : 0ull, keep3 = vlast > 3 ? ~0ull : 0ull;
  const int yd0 = (int)(((float)lane + 0.5f) * inv_qpr), j0 = lane - __mul24(yd0, qpr);
  const int nScore16 = (SPd * (dh + 2) + 3) >> 2;
  for (int pass = 0; pass < 2; pass++) {
    const int t = pass == 0 ? g.iniTh : g.minTh;
    const orbx_h2 tc2 = __builtin_bit_cast(orbx_h2, (uint32_t)t * 0x00010001u);  // t as two f16 subnormal patterns
    // clear the score tile (16-byte stores; its zero ring is part of it).  The barrier also publishes the image tile.
    for (int i = lane; i < nScore16; i += 64) reinterpret_cast<uint4*>(score)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (pass == 0) DET_MK();
    // dense corner test, 4 pixels per lane; corners are compacted into an LDS list and scored with dense
    // lanes (the score needs ~110 min/max ops: running it under per-lane divergence would dominate).
    // Stage 1 (4 pixels per lane, registers): compass pre-test -> survivor list.
    // Stage 2 (dense lanes over survivors): contrast M from the 16 ring pixels; corner iff M > t, score M - 1
    //          goes to the u8 score tile and the corner to the corner list (for the list-based NMS).
    // Lane validity is wave-uniform knowledge (entries base .. n-1 of a list are valid): it lives in scalar masks that
    // are ANDed with the v_cmp results, never in per-lane predicates.
    int nList = 0, sEnd = 0;
    bool overflowed = false;  // more corners than the list holds: the NMS falls back to scanning the score tile
    auto flush_survivors = [&]() {
      __syncthreads();
      const orbx_h2 th2 = __builtin_bit_cast(orbx_h2, (uint32_t)t * 0x00010001u);  // t in the same subnormal encoding
      const int s0 = nList;
      int nSurv = sEnd - s0;
#if ORBX_EVEN_FILTER
      {  // compaction of the survivor list in place by the even-ring test (entry k is read before entry k' <= k is written:
         // one wave, LDS operations complete in order)
        int wpos = 0;
        for (int base = 0; base < nSurv; base += 128) {
          const int rem = nSurv - base;
          const uint64_t vA = low_lanes(rem), vB = low_lanes(rem - 64);
          const int oA = list[s0 + min(base + lane, nSurv - 1)], oB = list[s0 + min(base + 64 + lane, nSurv - 1)];
          const orbx_h2 M = even_ring_contrast2_lds(tile8 + oA, tile8 + oB, TP);
          const uint64_t mA = __ballot(M.x > th2.x) & vA, mB = __ballot(M.y > th2.y) & vB;
          if (__builtin_amdgcn_inverse_ballot_w64(mA)) (list + s0 + wpos)[prefix_count(mA)] = (uint16_t)oA;
          wpos += __popcll(mA);
          if (__builtin_amdgcn_inverse_ballot_w64(mB)) (list + s0 + wpos)[prefix_count(mB)] = (uint16_t)oB;
          wpos += __popcll(mB);
        }
        nSurv = wpos;
        __syncthreads();
      }
#endif
      if ((DET_ABLATE & 4) && dw != 4095) nSurv = 0;
      for (int base = 0; base < nSurv; base += 128) {  // two survivors per lane (packed f16 contrast)
        const int rem = nSurv - base;
        const uint64_t vA = low_lanes(rem), vB = low_lanes(rem - 64);
        // a list entry IS the tile byte offset y * TP + x of the pixel's 7x7 window corner (image and score tile share the
        // pitch): no unpacking of (y, x) and no multiply per survivor / corner; (x, y) is only recovered when a corner is emitted
        const int oA = list[s0 + min(base + lane, nSurv - 1)], oB = list[s0 + min(base + 64 + lane, nSurv - 1)];
        // a last pass of <= 64 survivors gathers one pixel per lane only (17 instead of 34 byte reads: the LDS pipe is this
        // kernel's busiest); the packed network then carries the same pixel in both halves
        orbx_h2 M;
        if (rem <= 64) M = fast_contrast2_lds(tile8 + oA, tile8 + oA, TP);
        else if constexpr (TPC != 0 && ORBX_D16_PAIRS) M = fast_contrast2_lds_d16<TPC>(tile8 + oA, tile8 + oB);
        else M = fast_contrast2_lds(tile8 + oA, tile8 + oB, TP);
        const uint32_t Mbits = __builtin_bit_cast(uint32_t, M);  // a corner has M > t >= 0: the pattern is the integer
        const uint64_t mA = __ballot(M.x > th2.x) & vA, mB = __ballot(M.y > th2.y) & vB;
        if (__builtin_amdgcn_inverse_ballot_w64(mA)) {
          score8[oA + SPB + 4] = (uint8_t)((Mbits & 0xFFFFu) - 1);  // (y + 1) * pitch + x + 4: same pitch as the image tile
          const int k = prefix_count(mA);  // nList + k <= the position of the survivor it replaces
          if (k < cornerCap - nList) (list + nList)[k] = (uint16_t)oA;
        }
        nList += __popcll(mA);
        if (__builtin_amdgcn_inverse_ballot_w64(mB)) {
          score8[oB + SPB + 4] = (uint8_t)((Mbits >> 16) - 1);
          const int k = prefix_count(mB);
          if (k < cornerCap - nList) (list + nList)[k] = (uint16_t)oB;
        }
        nList += __popcll(mB);
      }
      if (nList > cornerCap) {
        overflowed = true;
        nList = cornerCap;
      }
      __syncthreads();
      sEnd = nList;
    };
    // Rounds.  kRowRounds (round 4, the pitches whose rows fill a wave well): a round is dq = floor(64 / qpr) WHOLE detect rows, a
    // lane keeps its (row in round, quad) and a round is one address increment -- no per-round stepping of (row, quad), no
    // clamp, the last-quad mask is loop invariant; 1 .. 4 lanes idle (9 / 10 / 12 quads per row: 63 / 60 / 60 lanes).  Otherwise
    // the quads are dealt to the lanes flat.  Lanes past the window read rows below the tile (still this block's LDS: masked).
    int yd = yd0, j = j0;
    int q0r = __mul24(yd0, TPd) + j0;
    const uint64_t notLastR = ~__ballot(j0 == qpr - 1);
    // ceil(dh / dq) = floor((dh - 0.5) / dq) + 1; as a quad count, so that one loop serves both schemes
    const int nRounds = kRowRounds ? ((int)(((float)dh - 0.5f) * __builtin_amdgcn_rcpf((float)dq)) + 1) * 64 : nq;
    for (int qb = 0; qb < ((DET_ABLATE & 2) && dw != 4095 ? 0 : nRounds); qb += 64) {
      uint64_t actM;
      int q0;
      if (kRowRounds) {
        actM = low_lanes_pos(min(dq, dh - (qb >> 6) * dq) * qpr);
        q0 = q0r;
        q0r += dq * TPd;
      } else {
        actM = low_lanes_pos(nq - qb);
        const int ydc = min(yd, dh - 1);  // idle lanes of the last round stay inside the tile (masked out below)
        q0 = __mul24(ydc, TPd) + j;       // (24-bit multiplies are full rate, v_mul_lo_u32 a quarter)
      }
      const uint32_t* row0 = tile + q0;
      uint32_t r[7][3];
#pragma unroll
      for (int i = 0; i < 7; i++) {
        r[i][0] = row0[i * TPd];
        r[i][1] = row0[i * TPd + 1];
        r[i][2] = row0[i * TPd + 2];
      }
      const uint64_t notLast = kRowRounds ? notLastR : ~__ballot(j == qpr - 1);  // a row's last quad may reach past the detectable window
      uint64_t sm[4];
      compass_pair<0>(r, tc2, sm[0], sm[1]);
      compass_pair<2>(r, tc2, sm[2], sm[3]);
      sm[0] &= actM;
      sm[1] &= actM & (notLast | keep1);
      sm[2] &= actM & (notLast | keep2);
      sm[3] &= actM & (notLast | keep3);
      // flush first when this round's survivors (<= 256) would not fit: the list then only needs room for a typical cell
      if (sEnd + (int)(__popcll(sm[0]) + __popcll(sm[1]) + __popcll(sm[2]) + __popcll(sm[3])) > listTotal) flush_survivors();
      const int yx = q0 << 2;   // tile byte offset of the quad's first pixel window (a multiple of 4: | pI adds the pixel)
      // ROW-MAJOR list order (second half of round 4): a survivor goes behind the survivors of the lower lanes -- all four pixels of
      // their quads: one chain of four mbcnt pairs -- and behind the lane's own lower pixels (a pointer bumped under the pixel's
      // lane mask).  The per-pixel segments of rounds 1 - 4 (all pixel-0 survivors of the round's seven rows, then all pixel-1 ...)
      // made the 32 lanes of a contrast-pass byte gather span seven tile rows: 2.04 LDS cycles per half-wave gather in a simulation on
      // the benchmark frames, against 1.31 for row-major order -- the gathers are the largest part of this kernel's LDS time, and
      // the LDS pipe is its busiest.  (Nothing downstream depends on the order of a cell's candidates.)
      uint32_t below = 0;
#pragma unroll
      for (int pI = 0; pI < 4; pI++)
        below = __builtin_amdgcn_mbcnt_hi((uint32_t)(sm[pI] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm[pI], below));
      uint16_t* wp = list + sEnd + below;
#pragma unroll
      for (int pI = 0; pI < 4; pI++) {
        const uint64_t m = sm[pI];
        if (__builtin_amdgcn_inverse_ballot_w64(m)) {  // this lane's bit of the SGPR mask, without a 64-bit vector shift
          *wp = (uint16_t)(yx | pI);
          wp++;
        }
        sEnd += __popcll(m);
      }
      if (!kRowRounds) {
        j += rq;
        yd += dq;
        if (j >= qpr) {
          j -= qpr;
          yd++;
        }
      }
    }
    if (pass == 0) DET_MK();
    flush_survivors();
    if (pass == 0) DET_MK();
    const int nCorners = ((DET_ABLATE & 8) && dw != 4095) ? 0 : nList;
    if ((DET_ABLATE & 8) && dw != 4095) kept = 1;
    // 3x3 non-max suppression (strict '>') inside the cell + emission
    if (TAP && pass == 0) {  // test tap (orbx_debug_score_map): the cell's FAST scores at iniThFAST, 0 = no corner
      uint8_t* dm = dbgScore + (long long)img * g.pyrImg + L.off;
      for (int i = lane; i < dw * dh; i += 64) {
        const int y = i / dw, x = i - y * dw;
        dm[(long long)(iniY + 3 + y) * L.pitch + iniX + 3 + x] = score8[(y + 1) * SPB + x + 4];
      }
    }
    if (!overflowed) {
      // common case: every corner of the cell is still in the list -> dense lanes, 9 LDS byte reads each, no branches
      const int SP = SPB;
      for (int base = 0; base < nCorners; base += 64) {
        const int oc = list[min(base + lane, nCorners - 1)];
        const uint8_t* c8 = score8 + oc + 3;  // top-left of the 3x3 neighbourhood
        const int sc = c8[SP + 1];
        const int n0 = max(max((int)c8[0], (int)c8[1]), (int)c8[2]);
        const int n1 = max(max((int)c8[SP], (int)c8[SP + 2]), (int)c8[2 * SP]);
        const int n2 = max(max((int)c8[2 * SP + 1], (int)c8[2 * SP + 2]), n0);
        const uint64_t m = __ballot(sc > max(n1, n2)) & low_lanes(nCorners - base);
        if (__builtin_amdgcn_inverse_ballot_w64(m)) {
          const int o = kept + prefix_count(m);
          // (x, y) from the offset: oc / TP by a 1-ulp reciprocal ((oc + 0.5) / TP stays 0.5 / TP away from the next integer)
          const int y = (int)(((float)oc + 0.5f) * inv_tp), x = oc - __mul24(y, SP);
          if (o < L.cellCap) out[o] = pack_key(iniX + 3 + x - kBorder, iniY + 3 + y - kBorder, sc);
        }
        kept += __popcll(m);
      }
    } else
    for (int qb = 0; qb < nq; qb += 64) {
      const int q = qb + lane;
      uint32_t sw = 0;
      int yd = 0, j = 0;
      if (q < nq) {
        yd = (int)(((float)q + 0.5f) * inv_qpr);
        j = q - yd * qpr;
        sw = score[(yd + 1) * SPd + j + 1];
      }
      uint32_t keepmask = 0;
      if (sw) {
        const int SP = SPB;
        const uint8_t* q8 = score8 + (yd + 1) * SP + 4 * (j + 1);
#pragma unroll
        for (int pI = 0; pI < 4; pI++) {
          const int sc = (sw >> (8 * pI)) & 0xFF;
          if (sc) {
            const uint8_t* c8 = q8 + pI;
            const bool keep = sc > c8[-1] && sc > c8[1] && sc > c8[-SP - 1] && sc > c8[-SP] && sc > c8[-SP + 1] &&
                              sc > c8[SP - 1] && sc > c8[SP] && sc > c8[SP + 1];
            keepmask |= (uint32_t)keep << pI;
          }
        }
      }
      const uint64_t any = __ballot(keepmask != 0);
      if (any) {
        int before = 0, total = 0;
#pragma unroll
        for (int pI = 0; pI < 4; pI++) {
          const uint64_t m = __ballot((keepmask >> pI) & 1u);
          before += prefix_count(m);
          total += __popcll(m);
        }
        const int pos = kept + before;
        int lanebefore = 0;
#pragma unroll
        for (int pI = 0; pI < 4; pI++) {
          if ((keepmask >> pI) & 1u) {
            const int o = pos + lanebefore;
            lanebefore++;
            if (o < L.cellCap)
              out[o] = pack_key(iniX + 3 + 4 * j + pI - kBorder, iniY + 3 + yd - kBorder, (sw >> (8 * pI)) & 0xFF);
          }
        }
        kept += total;
      }
    }
    if (kept > 0) break;
    __syncthreads();
  }
  if (lane == 0) *myCount = min(kept, L.cellCap);
#ifdef DET_PROF
  DET_MK();
  if (lane == 0 && (blockIdx.x % 16) == 3) printf("detw %d %d %lld %lld\n", l, kept, dq_[0] % 100000000, dq_[ndq_ - 1] % 100000000);
  if (lane == 0 && ((blockIdx.x % 293) == 7 || blockIdx.x + 1 == gridDim.x || blockIdx.x == 0) && ndq_ >= 6)
    printf("det img %d blk %4d level %d start %6lld: issue tile loads %d | tile in LDS + clear %d | stage 1 %d | flush (stage 2) %d | "
           "NMS + emit%s %d | total %d  kept %d\n", img, (int)blockIdx.x, l, dq_[0] % 1000000, (int)(dq_[1] - dq_[0]), (int)(dq_[2] - dq_[1]),
           (int)(dq_[3] - dq_[2]), (int)(dq_[4] - dq_[3]), ndq_ > 6 ? " + pass 2" : "", (int)(dq_[ndq_ - 1] - dq_[4]), (int)(dq_[ndq_ - 1] - dq_[0]), kept);
#endif
}

static int g_detect_list_cap = kListTotal;  // test hook: a smaller list forces the flush / carry / corner-overflow paths
void debug_set_detect_list_cap(int cap) { g_detect_list_cap = cap < 320 ? 320 : (cap > kListTotal ? kListTotal : cap); }

static size_t cell_lds_bytes(const Geom& g) {
  return (size_t)g.tileP * g.tileH + (size_t)g.scoreP * g.scoreH + 2 * kListTotal + 32;
}

// Cells of levels [level0, level1) only.
// (Round 4 also built k_detect_stack -- a wave owning a vertical stack of 2 .. 3 cells with pooled quads / survivors, scores kept
// in place of the image tile and a direct-mode fallback; bit-exact, 7 % fewer instructions per cell, and SLOWER: 6.9 KB of LDS per
// wave leave 23 instead of 29 waves per CU, 276 vs 241 us.  Commit a461b2d holds it; DESIGN.md 4 has the numbers.)
hipError_t launch_detect(const Geom& g, const Pyr& p, int nimg, uint32_t* cellCand, int* cellCount, int level0,
                         int level1, uint8_t* dbgScore, hipStream_t s) {
  const size_t lds = cell_lds_bytes(g);
  const int cellBegin = g.lv[level0].cellStart;
  const int cellEnd = level1 < g.nlevels ? g.lv[level1].cellStart : g.totalCells;
  if (cellEnd <= cellBegin) return hipSuccess;
  dim3 grid(cellEnd - cellBegin, nimg);
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, grid, dim3(64), lds, s, g, p, cellCand, cellCount, g_detect_list_cap, cellBegin, dbgScore);
  };
  const int tp = (!dbgScore && g.scoreP == g.tileP) ? g.tileP : 0;
  switch (tp) {
    case 44: go(k_detect<false, 44>); break;
    case 48: go(k_detect<false, 48>); break;
    case 52: go(k_detect<false, 52>); break;
    case 56: go(k_detect<false, 56>); break;
    default:
      if (dbgScore) go(k_detect<true, 0>); else go(k_detect<false, 0>);
  }
  return hipGetLastError();
}

// raises the dynamic-LDS limit of every detect instantiation (called from prepare_kernels, once per configure)
hipError_t prepare_detect(const Geom& g) {
  const size_t lds_det = cell_lds_bytes(g);
  const void* dk[6] = {reinterpret_cast<const void*>(k_detect<true, 0>),   reinterpret_cast<const void*>(k_detect<false, 0>),
                       reinterpret_cast<const void*>(k_detect<false, 44>), reinterpret_cast<const void*>(k_detect<false, 48>),
                       reinterpret_cast<const void*>(k_detect<false, 52>), reinterpret_cast<const void*>(k_detect<false, 56>)};
  for (const void* f : dk) {
    hipError_t e = raise_dynamic_lds(f, lds_det);   // (per device, never lowered by another handle: orbx_kernels.hip)
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace orbx
