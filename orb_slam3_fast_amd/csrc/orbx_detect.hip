// orbx_detect.hip — FAST-9-16 detection kernels of the ORB front-end (gfx950 / CDNA4, wave64).
//
// ORBextractor::ComputeKeyPointsOctTree's cell loop (src/ORBextractor.cc:892-971 of the reference): per-cell cv::FAST with
// non-max suppression at iniThFAST, minThFAST for cells that come out empty.  SURVEY A3 / B3.
#include <algorithm>
#include <cstdlib>

#include "orbx_device.h"

namespace orbx {

// ================================================================================================ detect
// FAST-9-16 on an LDS tile, in two stages (k_detect).
//
// Layout: the cell ROI (cell + 3 px FAST halo each side) sits in LDS with ROI column 0 on a dword boundary
// (the loader funnel-shifts the unaligned global row).  Stage 1: a lane owns a "quad" of 4 detectable pixels and reads
// the three rows that hold ring pixels 0 / 4 / 8 / 12 of the quad (7 dwords, kept in registers, every ring byte a
// compile-time (register, byte) pair -> SDWA operands) for the compass pre-test; survivors go to an LDS list.
// Stage 2: dense lanes, two survivors per lane in packed f16: the exact contrast M from the 16 ring pixels (sliding
// 9-windows of minima / maxima built from 3-windows); corner iff M > t, score M - 1.  Then list-based 3x3 NMS.
constexpr int kRingDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
constexpr int kRingDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
#ifndef ORBX_COMPASS_PAIRS
#define ORBX_COMPASS_PAIRS 1
#endif
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

// Necessary condition for a 9-arc: two cyclically adjacent compass pixels (ring 0, 4, 8, 12) of the same
// polarity.  Per pixel slot 10 VALU operations and one scalar OR.  Returns the wave mask of lanes whose pixel survives.
template <int P>
__device__ __forceinline__ uint64_t compass_wave(const uint32_t (&r)[7][3], int t) {
  const int c = (r[3][(3 + P) >> 2] >> (8 * ((3 + P) & 3))) & 0xFF;
  int v[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int k = 4 * q;
    const int col = 3 + P + kRingDX[k];
    v[q] = (r[3 + kRingDY[k]][col >> 2] >> (8 * (col & 3))) & 0xFF;
  }
  // two cyclically adjacent compass points above c + t  <=>  one of {0, 2} and one of {1, 3} (in a 4-cycle every even
  // position is adjacent to every odd one)  <=>  min(max(v0, v2), max(v1, v3)) > c + t; likewise below c - t.  The byte
  // selects ride on the SDWA operands of v_max / v_min, and only two masks reach the scalar ALU (it is nearly as busy
  // as the vector ALU in this kernel).
  const int hiMin = min(max(v[0], v[2]), max(v[1], v[3])), loMax = max(min(v[0], v[2]), min(v[1], v[3]));
  return __ballot(hiMin > c + t) | __ballot(loMax < c - t);
}

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
// The same necessary condition on TWO horizontally adjacent pixels (P, P + 1) in packed f16 (round 4): the five operands of
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

// a8 / b8 point at the TOP-LEFT corner of each pixel's 7x7 window, so every ring offset is a non-negative ds_read immediate.
typedef unsigned short orbx_us2 __attribute__((ext_vector_type(2)));
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
  orbx_us2 cv;
  cv.x = a8[3 * TP + 3];
  cv.y = b8[3 * TP + 3];
  const orbx_h2 c = __builtin_bit_cast(orbx_h2, cv);
  return __builtin_elementwise_maximum(maxmin - c, c - minmax);
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
  // Runs of xcdRun horizontally consecutive cells share an XCD and therefore the L2 lines of their common halo
  // columns (HBM-side fetch 410 -> 151 MB per 64-image launch; same duration, the kernel is VALU-bound).
  int cell = cellBegin + xcd_run_remap<kDetectXcdRun>(blockIdx.x, gridDim.x, blockIdx.y);
  int l = 0;  // the scalar ALU is nearly as busy as the vector ALU in this kernel: no search loop, no integer divisions
#pragma unroll
  for (int q = 1; q < ORBX_MAX_LEVELS; q++) l += cell >= g.levelCell[q] ? 1 : 0;
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
  bool tileDone = false;
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
      const int s0 = nList, nSurv = sEnd - s0;
      for (int base = 0; base < nSurv; base += 128) {  // two survivors per lane (packed f16 contrast)
        const int rem = nSurv - base;
        const uint64_t vA = low_lanes(rem), vB = low_lanes(rem - 64);
        // a list entry IS the tile byte offset y * TP + x of the pixel's 7x7 window corner (image and score tile share the
        // pitch): no unpacking of (y, x) and no multiply per survivor / corner; (x, y) is only recovered when a corner is emitted
        const int oA = list[s0 + min(base + lane, nSurv - 1)], oB = list[s0 + min(base + 64 + lane, nSurv - 1)];
        const orbx_h2 M = fast_contrast2_lds(tile8 + oA, tile8 + oB, TP);
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
    for (int qb = 0; qb < nRounds; qb += 64) {
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
#if ORBX_COMPASS_PAIRS
      compass_pair<0>(r, tc2, sm[0], sm[1]);
      compass_pair<2>(r, tc2, sm[2], sm[3]);
      sm[0] &= actM;
      sm[1] &= actM & (notLast | keep1);
      sm[2] &= actM & (notLast | keep2);
      sm[3] &= actM & (notLast | keep3);
#else
      sm[0] = compass_wave<0>(r, t) & actM;
      sm[1] = compass_wave<1>(r, t) & actM & (notLast | keep1);
      sm[2] = compass_wave<2>(r, t) & actM & (notLast | keep2);
      sm[3] = compass_wave<3>(r, t) & actM & (notLast | keep3);
#endif
      // flush first when this round's survivors (<= 256) would not fit: the list then only needs room for a typical cell
      if (sEnd + (int)(__popcll(sm[0]) + __popcll(sm[1]) + __popcll(sm[2]) + __popcll(sm[3])) > listTotal) flush_survivors();
      const int yx = q0 << 2;   // tile byte offset of the quad's first pixel window (a multiple of 4: | pI adds the pixel)
#pragma unroll
      for (int pI = 0; pI < 4; pI++) {
        const uint64_t m = sm[pI];
        if (__builtin_amdgcn_inverse_ballot_w64(m))  // this lane's bit of the SGPR mask, without a 64-bit vector shift
          (list + sEnd)[prefix_count(m)] = (uint16_t)(yx | pI);  // (the running end of the list is the store's scalar base)
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
    flush_survivors();
    const int nCorners = nList;
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
}

// ================================================================================================ detect, stacked cells
// Round 4: one wave owns a vertical STACK of NC cells of one cell column (k_detect_stack).
//  * The ROIs of the NC cells sit one under the other in ONE LDS tile, each with its own 3-px halo rows (sub-tile stride
//    RH = hCell + 6 rows): between the detectable rows of two cells lie 6 rows nobody scores, so stage 2 and the NMS need no
//    knowledge of cells at all -- only stage 1's row walk (tile row = detect row + 6 per cell boundary crossed) and the
//    emission (cell = offset / sub-tile bytes) do.
//  * Quads and survivors of the whole stack are pooled: the last, partly filled round of every cell (13 % of stage 1) and
//    the last, partly filled contrast pass (19 % of stage 2: 2.9 passes for 2.4 passes' worth of survivors) are paid once
//    per stack instead of once per cell; the prologue as well.
//  * No score tile: corners carry their score in a byte array beside the list; when the stack is through, the image tile
//    is cleared and the scores are scattered INTO it for the list-based NMS.  LDS per cell 5.4 -> 3.5 KB.
//  * Jobs: the stack at iniThFAST; a cell whose post-NMS set is empty again at minThFAST (:942-959; its sub-tile is
//    fetched again, the clear destroyed it).  More corners than the list holds: the stack is redone cell by cell, and a
//    single cell that still overflows in "direct" mode -- no stored corners, a corner's eight neighbours are scored on
//    the fly (slow, exact, a few cells of white-noise images).
// Levels with degenerate cells (images above ~1260 px high) or other pitches keep the one-wave-per-cell kernel above.
struct DetStackPlan {
  int levelStack[ORBX_MAX_LEVELS + 1];  // first stack of level l among the launch's stacks (levels it skips: empty range)
  uint8_t ncl[ORBX_MAX_LEVELS];         // cells per stack at level l (1 .. NC: levels with tall cells stack fewer)
  int tileBytes;                        // LDS bytes of the image tile (max over the levels of ncl * (hCell + 6) * pitch, 16-aligned)
};

// An opaque copy of a wave-uniform value: what is computed from it cannot be hoisted out of the enclosing loop, so the
// level / cell geometry the rare blocks need (loader, emission) is re-read from the kernel arguments where it is used
// instead of occupying scalar registers through the hot loops (the kernel sat at the 100-SGPR limit and spilled).
__device__ __forceinline__ int opaque_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

#ifndef ORBX_STACK_SGPRS
#define ORBX_STACK_SGPRS 80
#endif
template <bool TAP, int NC, int TPC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(ORBX_STACK_SGPRS))) void k_detect_stack(Geom g, Pyr p, DetStackPlan sp, uint32_t* __restrict__ cellCand,
                                                     int* __restrict__ cellCount, int listCap,
                                                     uint8_t* __restrict__ dbgScore) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr int TP = TPC, TPd = TPC / 4;
  const int lane = threadIdx.x, img = blockIdx.y;
  const int blk = xcd_run_remap<kDetectXcdRun>(blockIdx.x, gridDim.x, blockIdx.y);
  int l = 0;
#pragma unroll
  for (int q = 1; q < ORBX_MAX_LEVELS; q++) l += blk >= sp.levelStack[q] ? 1 : 0;
  int ci0, cj, nc, dw, hC, Rall;
  {
    const LevelDev& L = g.lv[l];
    const int b2 = blk - sp.levelStack[l];
    const int srow = __builtin_amdgcn_readfirstlane((int)(((float)b2 + 0.5f) * __builtin_amdgcn_rcpf((float)L.nCols)));
    cj = b2 - srow * L.nCols;
    const int ncl = sp.ncl[l];
    ci0 = srow * ncl;
    nc = min(ncl, L.nRows - ci0);
    hC = L.hCell;
    dw = min(L.wCell + 6, L.w - kBorder - (kBorder + cj * L.wCell)) - 6;  // detectable columns (the plan only admits levels with dw, dh >= 1)
    Rall = min(nc * hC, L.h - kBorder - 6 - (kBorder + ci0 * hC));        // detectable rows of the stack (a level's last cell may be short)
  }
  const int RH = hC + 6, subBytes = RH * TP;

  uint32_t* tile = reinterpret_cast<uint32_t*>(smem);
  uint8_t* tile8 = smem;
  uint16_t* list = reinterpret_cast<uint16_t*>(smem + sp.tileBytes);   // corners [0, nList), compass survivors [nList, sEnd)
  uint8_t* cs = smem + sp.tileBytes + 2 * listCap;                     // score of corner k
  const int listTotal = listCap, cornerCap = listTotal - 256;

  int kept[NC];
#pragma unroll
  for (int k = 0; k < NC; k++) kept[k] = 0;

  const int qpr = (dw + 3) >> 2;  // quads per detect row
  const float inv_qpr = 1.0f / (float)qpr;
  constexpr float inv_tp = 1.0f / (float)TP;
  const int dq = __builtin_amdgcn_readfirstlane((int)(64.5f * inv_qpr)), rq = 64 - dq * qpr;
  const int vlast = dw - 4 * (qpr - 1);
  const int yd0 = (int)(((float)lane + 0.5f) * inv_qpr), j0 = lane - __mul24(yd0, qpr);

  // a kept corner -> its cell's slots.  oc = tile offset of the pixel's window corner, sc = score
  auto emit = [&](uint64_t m, int oc, int sc) {
    if (!m) return;
    const LevelDev& L = g.lv[opaque_s(l)];
    const int cjq = opaque_s(cj), ci0q = opaque_s(ci0);
    const int yt = (int)(((float)oc + 0.5f) * inv_tp), x = oc - __mul24(yt, TP);
    int kc = 0;
#pragma unroll
    for (int b = 1; b < NC; b++) kc += oc >= b * subBytes ? 1 : 0;
    const uint32_t key = pack_key(cjq * L.wCell + 3 + x, ci0q * L.hCell + 3 + yt - 6 * kc, sc);  // relative to the (16, 16) window origin
    uint32_t* outBase = cellCand + (long long)img * g.cellImg + L.cellOff + (long long)(ci0q * L.nCols + cjq) * L.cellCap;
    const int outStride = L.nCols * L.cellCap, cellCap = L.cellCap;
#pragma unroll
    for (int k = 0; k < NC; k++) {
      const uint64_t mk = NC == 1 ? m : (m & __ballot(kc == k));
      if (__builtin_amdgcn_inverse_ballot_w64(mk)) {
        const int o = kept[k] + prefix_count(mk);
        if (o < cellCap) outBase[k * outStride + o] = key;
      }
      kept[k] += __popcll(mk);
    }
  };
  auto tap_score = [&](int oc, int sc) {  // test tap: the FAST score of a corner at iniThFAST into the level-sized map
    const LevelDev& L = g.lv[opaque_s(l)];
    const int yt = (int)(((float)oc + 0.5f) * inv_tp), x = oc - __mul24(yt, TP);
    int kc = 0;
#pragma unroll
    for (int b = 1; b < NC; b++) kc += oc >= b * subBytes ? 1 : 0;
    uint8_t* dm = dbgScore + (long long)img * g.pyrImg + L.off;
    dm[(long long)(kBorder + ci0 * L.hCell + 3 + yt - 6 * kc) * L.pitch + kBorder + cj * L.wCell + 3 + x] = (uint8_t)sc;
  };

  // ---- tile load: the sub-tiles of cells [k0, k1).  Tile row tr = k * RH + r  <-  level row iniY0 + tr - 6 k (clamped into
  // the level: rows past a short last cell are never consumed)
  auto load_tiles = [&](int k0, int k1) {
    __syncthreads();
    const int nRows = (k1 - k0) * RH, rowBase = k0 * RH;
    const int lq = opaque_s(l);
    const LevelDev& L = g.lv[lq];
    int pitch;
    const uint8_t* im = level_ptr(g, p, img, lq, pitch);
    const int iniX = kBorder + cj * L.wCell, iniY0 = kBorder + ci0 * L.hCell;
    if (iniX + TPC <= L.w) {
      // a row is TPC / P pieces of P = 16, 8 or 4 bytes; lane = (row, piece), one UNALIGNED global load straight from the
      // ROI's first byte and one aligned LDS store per piece (see k_detect)
      constexpr int P = (TPC % 16 == 0) ? 16 : (TPC % 8 == 0) ? 8 : 4, PPR = TPC / P;
      constexpr uint32_t kInv = (65536u + PPR - 1) / PPR;
      static_assert(PPR * ((4095u * kInv) >> 16) <= 4095u && (4095u / PPR) == ((4095u * kInv) >> 16), "reciprocal division");
      const uint8_t* colBase = im + iniX;
      const int nItems = nRows * PPR;
      constexpr int kB = 3;
      for (int base = 0; base < nItems; base += 64 * kB) {
        uint4 v[kB];
        int dst[kB];
#pragma unroll
        for (int u = 0; u < kB; u++) {
          const int it = min(base + lane + 64 * u, nItems - 1);
          const int rl = (int)(((uint32_t)it * kInv) >> 16), c = it - PPR * rl;
          int kk = 0;
#pragma unroll
          for (int b = 1; b < NC; b++) kk += rl >= b * RH ? 1 : 0;
          const int sy = min(iniY0 + rowBase + rl - 6 * (k0 + kk), L.h - 1);
          const uint8_t* src = colBase + (uint32_t)(__mul24(sy, pitch) + P * c);
          if (P == 16) v[u] = load_u128_unaligned(src);
          else if (P == 8) { const uint2 w2 = load_u64_unaligned(src); v[u] = make_uint4(w2.x, w2.y, 0, 0); }
          else v[u] = make_uint4(load_u32_unaligned(src), 0, 0, 0);
          dst[u] = (rowBase + rl) * TPC + P * c;
        }
#pragma unroll
        for (int u = 0; u < kB; u++) {
          if (P == 16) *reinterpret_cast<uint4*>(smem + dst[u]) = v[u];
          else if (P == 8) *reinterpret_cast<uint2*>(smem + dst[u]) = make_uint2(v[u].x, v[u].y);
          else *reinterpret_cast<uint32_t*>(smem + dst[u]) = v[u].x;
        }
      }
    } else {  // last cell column of a level: dwords, bytes where the image row ends
      const int nItems = nRows * TPd;
      for (int it = lane; it < nItems; it += 64) {
        const int rl = (int)(((float)it + 0.5f) * (1.0f / (float)TPd)), c = it - TPd * rl;
        int kk = 0;
#pragma unroll
        for (int b = 1; b < NC; b++) kk += rl >= b * RH ? 1 : 0;
        const int sy = min(iniY0 + rowBase + rl - 6 * (k0 + kk), L.h - 1), gx = iniX + 4 * c;
        const uint8_t* src = im + (long long)sy * pitch + gx;
        uint32_t v = 0;
        if (gx + 4 <= L.w) v = load_u32_unaligned(src);
        else
          for (int b = 0; b < 4; b++)
            if (gx + b < L.w) v |= (uint32_t)src[b] << (8 * b);
        tile[(rowBase + rl) * TPd + c] = v;
      }
    }
    __syncthreads();
  };

  // ---- stage 1 over the detect rows of cells [k0, k1): compass pre-test, 4 pixels per lane; the survivors go to the list
  // behind the corners and `flush` (stage 2) turns the survivors [nList, sEnd) into whatever the mode keeps.  flush returns
  // true to abandon the job.
  int nList = 0, sEnd = 0;
  auto stage1 = [&](int k0, int k1, int t, auto&& flush) {
    const int nck = k1 - k0, rowBase = k0 * RH;
    const int R = k1 == nc ? Rall - k0 * hC : nck * hC;  // detect rows of the job
    const int hB = nck > 1 ? hC : (1 << 20);             // rows per cell inside the job (one cell: no boundary)
    const int nq = qpr * R;
    nList = 0;
    sEnd = 0;
    int yd = yd0, j = j0;
    for (int qb = 0;; qb += 64) {
      const bool last = qb >= nq;
      uint64_t sm[4] = {0ull, 0ull, 0ull, 0ull};
      int cnt = 0, yx = 0;
      if (!last) {
        const uint64_t actM = low_lanes_pos(nq - qb);
        const int ydc = min(yd, R - 1);  // idle lanes of the last round stay inside the tile (masked out below)
        int trow = ydc + rowBase;
#pragma unroll
        for (int b = 1; b < NC; b++) trow += ydc >= b * hB ? 6 : 0;
        const int q0 = __mul24(trow, TPd) + j;
        const uint32_t* row0 = tile + q0;
        uint32_t r[7][3];
#pragma unroll
        for (int i = 0; i < 7; i++) {
          r[i][0] = row0[i * TPd];
          r[i][1] = row0[i * TPd + 1];
          r[i][2] = row0[i * TPd + 2];
        }
        const uint64_t notLast = ~__ballot(j == qpr - 1);  // a row's last quad may reach past the detectable window
#if ORBX_COMPASS_PAIRS
        const orbx_h2 tc2 = __builtin_bit_cast(orbx_h2, (uint32_t)t * 0x00010001u);
        compass_pair<0>(r, tc2, sm[0], sm[1]);
        compass_pair<2>(r, tc2, sm[2], sm[3]);
        sm[0] &= actM;
        sm[1] &= actM & (vlast > 1 ? ~0ull : notLast);
        sm[2] &= actM & (vlast > 2 ? ~0ull : notLast);
        sm[3] &= actM & (vlast > 3 ? ~0ull : notLast);
#else
        sm[0] = compass_wave<0>(r, t) & actM;
        sm[1] = compass_wave<1>(r, t) & actM & (vlast > 1 ? ~0ull : notLast);
        sm[2] = compass_wave<2>(r, t) & actM & (vlast > 2 ? ~0ull : notLast);
        sm[3] = compass_wave<3>(r, t) & actM & (vlast > 3 ? ~0ull : notLast);
#endif
        cnt = (int)(__popcll(sm[0]) + __popcll(sm[1]) + __popcll(sm[2]) + __popcll(sm[3]));
        yx = q0 << 2;  // tile byte offset of the quad's first pixel window (a multiple of 4: | pI adds the pixel)
      }
      if (last || sEnd + cnt > listTotal) {
        if (flush() || last) break;
      }
#pragma unroll
      for (int pI = 0; pI < 4; pI++) {
        const uint64_t m = sm[pI];
        if (__builtin_amdgcn_inverse_ballot_w64(m))
          list[__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, (uint32_t)sEnd))] =
              (uint16_t)(yx | pI);
        sEnd += __popcll(m);
      }
      j += rq;
      yd += dq;
      if (j >= qpr) {
        j -= qpr;
        yd++;
      }
    }
  };
  auto kept_at = [&](int k) {
    int v = kept[0];
#pragma unroll
    for (int q = 1; q < NC; q++) v = k == q ? kept[q] : v;
    return v;
  };

  // ---- jobs: cells [jk0, jk1) at threshold t
  int jk0 = 0, jk1 = nc, t = g.iniTh, phase = 0, cur = 0;  // phase 0: the stack at iniTh, 1: one cell at iniTh, 2: one cell at minTh
  bool split = false;
  uint32_t intact = 0;      // sub-tiles that hold their image rows (the in-place scores destroy them)
  uint32_t directMask = 0;  // cells left to the direct mode below
  for (;;) {
    const uint32_t need = ((1u << jk1) - 1u) & ~((1u << jk0) - 1u);
    if ((intact & need) != need) {
      load_tiles(jk0, jk1);
      intact |= need;
    }
    const bool tapJob = TAP && phase != 2;
    bool overflowed = false;
    // stage 2: the survivors [nList, sEnd) -> corners (list + score array), in place
    stage1(jk0, jk1, t, [&]() {
      __syncthreads();
      const orbx_h2 th2 = __builtin_bit_cast(orbx_h2, (uint32_t)t * 0x00010001u);  // t in the subnormal encoding
      const int s0 = nList, nSurv = sEnd - s0;
      for (int base = 0; base < nSurv; base += 128) {  // two survivors per lane (packed f16 contrast)
        const int rem = nSurv - base;
        const uint64_t vA = low_lanes(rem), vB = low_lanes(rem - 64);
        const int oA = list[s0 + min(base + lane, nSurv - 1)], oB = list[s0 + min(base + 64 + lane, nSurv - 1)];
        const orbx_h2 M = fast_contrast2_lds(tile8 + oA, tile8 + oB, TP);
        const uint32_t Mbits = __builtin_bit_cast(uint32_t, M);  // a corner has M > t >= 0: the pattern is the integer
        const uint64_t mA = __ballot(M.x > th2.x) & vA, mB = __ballot(M.y > th2.y) & vB;
        if (__builtin_amdgcn_inverse_ballot_w64(mA)) {
          const int o = nList + prefix_count(mA);  // <= the position of the survivor it replaces
          if (o < cornerCap) {
            list[o] = (uint16_t)oA;
            cs[o] = (uint8_t)((Mbits & 0xFFFFu) - 1);
          }
        }
        nList += __popcll(mA);
        if (__builtin_amdgcn_inverse_ballot_w64(mB)) {
          const int o = nList + prefix_count(mB);
          if (o < cornerCap) {
            list[o] = (uint16_t)oB;
            cs[o] = (uint8_t)((Mbits >> 16) - 1);
          }
        }
        nList += __popcll(mB);
      }
      if (nList > cornerCap) overflowed = true;
      __syncthreads();
      sEnd = nList;
      return overflowed;
    });

    if (overflowed) {  // more corners than the list holds: nothing was emitted, the tile is intact
      if (jk1 - jk0 > 1) {  // the stack cell by cell ...
        split = true;
        phase = 1;
        cur = 0;
        jk0 = 0;
        jk1 = 1;
        continue;
      }
      directMask |= 1u << jk0;  // ... a single cell without stored corners (below)
    } else if (nList > 0) {
      // ---- the scores move into the cleared image tile (no separate score tile), then list-based 3x3 NMS (strict '>')
      const int nCorners = nList;
      const int z0 = (jk0 * subBytes) >> 4, z1 = (jk1 * subBytes + 15) >> 4;
      for (int i = z0 + lane; i < z1; i += 64) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
      intact = 0;
      __syncthreads();
      for (int base = 0; base < nCorners; base += 64) {
        const int k = min(base + lane, nCorners - 1);
        const int oc = list[k], sc = cs[k];
        tile8[oc + 3 * TP + 3] = (uint8_t)sc;
        if (tapJob) tap_score(oc, sc);
      }
      __syncthreads();
      for (int base = 0; base < nCorners; base += 64) {
        const int oc = list[min(base + lane, nCorners - 1)];
        const uint8_t* c8 = tile8 + oc + 2 * TP + 2;  // top-left of the 3x3 neighbourhood
        const int sc = c8[TP + 1];
        const int n0 = max(max((int)c8[0], (int)c8[1]), (int)c8[2]);
        const int n1 = max(max((int)c8[TP], (int)c8[TP + 2]), (int)c8[2 * TP]);
        const int n2 = max(max((int)c8[2 * TP + 1], (int)c8[2 * TP + 2]), n0);
        emit(__ballot(sc > max(n1, n2)) & low_lanes(nCorners - base), oc, sc);
      }
    }
    // ---- next job: a cell that came out empty at iniTh is redone at minTh; after a split, the next cell at iniTh
    const bool redo = g.minTh != g.iniTh;
    if (phase == 0) cur = 0;
    else if (phase == 1 && redo && kept_at(cur) == 0 && !(directMask >> cur & 1u)) {
      phase = 2;
      t = g.minTh;
      continue;  // same cell
    } else cur++;
    for (; cur < nc; cur++) {
      if (directMask >> cur & 1u) continue;
      if (split) {
        phase = 1;
        t = g.iniTh;
        break;
      }
      if (redo && kept_at(cur) == 0) {
        phase = 2;
        t = g.minTh;
        break;
      }
    }
    if (cur >= nc) break;
    jk0 = cur;
    jk1 = cur + 1;
  }

#ifndef ORBX_STACK_NO_DIRECT
  // ---- direct mode (a single cell whose corners do not fit the list: a few cells of white-noise images, and the tests):
  // nothing is stored, the eight neighbours of a corner are scored on the fly (outside the cell's detectable window: 0, like
  // the zero ring of a score tile); kept corners are emitted at once.  Slow and exact.
  for (int k = 0; directMask >> k; k++) {
    if (!(directMask >> k & 1u)) continue;
    for (int pass = 0; pass < 2; pass++) {
      const int td = pass == 0 ? g.iniTh : g.minTh;
      load_tiles(k, k + 1);
      const int dhK = k == nc - 1 ? Rall - (nc - 1) * hC : hC;
      stage1(k, k + 1, td, [&]() {
        __syncthreads();
        const orbx_h2 th2 = __builtin_bit_cast(orbx_h2, (uint32_t)td * 0x00010001u);
        const int nSurv = sEnd;
        for (int base = 0; base < nSurv; base += 128) {
          const int rem = nSurv - base;
          const uint64_t vA = low_lanes(rem), vB = low_lanes(rem - 64);
          const int oA = list[min(base + lane, nSurv - 1)], oB = list[min(base + 64 + lane, nSurv - 1)];
          const orbx_h2 M = fast_contrast2_lds(tile8 + oA, tile8 + oB, TP);
          const uint32_t Mbits = __builtin_bit_cast(uint32_t, M);
          const uint64_t mA = __ballot(M.x > th2.x) & vA, mB = __ballot(M.y > th2.y) & vB;
          if (!(mA | mB)) continue;
          const int scA = (int)(Mbits & 0xFFFFu) - 1, scB = (int)(Mbits >> 16) - 1;
          const int ytA = (int)(((float)oA + 0.5f) * inv_tp), xA = oA - __mul24(ytA, TP), ylA = ytA - k * RH;
          const int ytB = (int)(((float)oB + 0.5f) * inv_tp), xB = oB - __mul24(ytB, TP), ylB = ytB - k * RH;
          int nmA = 0, nmB = 0;
#pragma unroll 1
          for (int n = 0; n < 8; n++) {
            const int n9 = n + (n >= 4 ? 1 : 0), dy = n9 / 3 - 1, dx = n9 - 3 * (n9 / 3) - 1, d = dy * TP + dx;
            const orbx_h2 Mn = fast_contrast2_lds(tile8 + max(oA + d, 0), tile8 + max(oB + d, 0), TP);
            const uint32_t nb = __builtin_bit_cast(uint32_t, Mn);
            const bool okA = (unsigned)(xA + dx) < (unsigned)dw && (unsigned)(ylA + dy) < (unsigned)dhK && Mn.x > th2.x;
            const bool okB = (unsigned)(xB + dx) < (unsigned)dw && (unsigned)(ylB + dy) < (unsigned)dhK && Mn.y > th2.y;
            nmA = max(nmA, okA ? (int)(nb & 0xFFFFu) - 1 : 0);
            nmB = max(nmB, okB ? (int)(nb >> 16) - 1 : 0);
          }
          if (TAP && pass == 0) {
            if (__builtin_amdgcn_inverse_ballot_w64(mA)) tap_score(oA, scA);
            if (__builtin_amdgcn_inverse_ballot_w64(mB)) tap_score(oB, scB);
          }
          emit(mA & __ballot(scA > nmA), oA, scA);
          emit(mB & __ballot(scB > nmB), oB, scB);
        }
        __syncthreads();
        sEnd = 0;
        return false;
      });
      if (kept_at(k) > 0 || g.minTh == g.iniTh) break;
    }
  }
#endif
  if (lane < nc) {
    int v = kept[0];
#pragma unroll
    for (int q = 1; q < NC; q++) v = lane == q ? kept[q] : v;
    const LevelDev& L = g.lv[l];
    cellCount[(long long)img * g.totalCells + L.cellStart + ci0 * L.nCols + cj + lane * L.nCols] = min(v, L.cellCap);
  }
}

static int g_detect_list_cap = 1 << 20;  // test hook: a smaller list forces the flush / carry / corner-overflow paths
void debug_set_detect_list_cap(int cap) { g_detect_list_cap = cap < 320 ? 320 : cap; }

// cells per stack of k_detect_stack: ORBX_DETECT_NC = 0 (default: one wave per cell only), 1, 2, 3 (experimental); read once
static int detect_nc() {
  static const int nc = [] {
    const char* e = getenv("ORBX_DETECT_NC");
    const int v = e ? atoi(e) : 0;
    return v < 0 ? 0 : (v > 3 ? 3 : v);
  }();
  return nc;
}
// list entries of a stack (corners + waiting survivors); ORBX_DETECT_LCAP overrides (experiments)
static int stack_list_cap(int nc) {
  static const int env = [] { const char* e = getenv("ORBX_DETECT_LCAP"); return e ? atoi(e) : 0; }();
  const int v = env > 0 ? env : (nc == 1 ? 704 : nc == 2 ? 1024 : 1344);
  return std::max(320, std::min(v, 4096)) & ~7;
}
static size_t cell_lds_bytes(const Geom& g) {
  return (size_t)g.tileP * g.tileH + (size_t)g.scoreP * g.scoreH + 2 * kListTotal + 32;
}
// k_detect_stack takes a level when every cell of it has a detectable window (no degenerate last row / column: images up to
// ~1260 px high) and the tile pitch is one of the compile-time ones
static bool stack_level_ok(const Geom& g, int l) {
  const LevelDev& L = g.lv[l];
  if (g.scoreP != g.tileP || (g.tileP != 44 && g.tileP != 48 && g.tileP != 52 && g.tileP != 56)) return false;
  const int maxBX = L.w - kBorder, maxBY = L.h - kBorder;
  const int iniXl = kBorder + (L.nCols - 1) * L.wCell, iniYl = kBorder + (L.nRows - 1) * L.hCell;
  return std::min(L.wCell + 6, maxBX - iniXl) - 6 >= 1 && std::min(L.hCell + 6, maxBY - iniYl) - 6 >= 1;
}

template <bool TAP, int NC>
static void launch_stack_nc(const Geom& g, const Pyr& p, const DetStackPlan& sp, int nStacks, int nimg, uint32_t* cellCand,
                            int* cellCount, uint8_t* dbgScore, hipStream_t s) {
  const int cap = std::min(g_detect_list_cap, stack_list_cap(NC)) & ~7;
  const size_t lds = (size_t)sp.tileBytes + 2 * (size_t)cap + (size_t)(cap - 256) + 16;
  dim3 grid(nStacks, nimg);
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(64), lds, s, g, p, sp, cellCand, cellCount, cap, dbgScore); };
  switch (g.tileP) {
    case 44: go(k_detect_stack<TAP, NC, 44>); break;
    case 48: go(k_detect_stack<TAP, NC, 48>); break;
    case 52: go(k_detect_stack<TAP, NC, 52>); break;
    default: go(k_detect_stack<TAP, NC, 56>); break;
  }
}

// Cells of levels [level0, level1) only.
hipError_t launch_detect(const Geom& g, const Pyr& p, int nimg, uint32_t* cellCand, int* cellCount, int level0,
                         int level1, uint8_t* dbgScore, hipStream_t s) {
  level1 = std::min(level1, g.nlevels);
  const int NC = detect_nc();
  // ---- levels the stack kernel takes (one launch)
  DetStackPlan sp;
  int nStacks = 0;
  uint32_t stackMask = 0;
  for (int l = 0; l <= ORBX_MAX_LEVELS; l++) sp.levelStack[l] = 0x7fffffff;
  // rows of the LDS tile: NC cells of the first stacked level; levels with taller cells put fewer cells on a stack
  int rowsBudget = 0, rowsMax = 0;
  for (int l = level0; l < level1; l++)
    if (NC > 0 && stack_level_ok(g, l)) {
      if (!rowsBudget) rowsBudget = NC * (g.lv[l].hCell + 6);
      rowsBudget = std::max(rowsBudget, g.lv[l].hCell + 6);
    }
  for (int l = 0; l < ORBX_MAX_LEVELS; l++) sp.ncl[l] = 1;
  for (int l = level0; l < level1; l++) {
    sp.levelStack[l] = nStacks;
    if (NC > 0 && stack_level_ok(g, l)) {
      const int n = std::max(1, std::min(NC, rowsBudget / (g.lv[l].hCell + 6)));
      sp.ncl[l] = (uint8_t)n;
      rowsMax = std::max(rowsMax, n * (g.lv[l].hCell + 6));
      nStacks += g.lv[l].nCols * ((g.lv[l].nRows + n - 1) / n);
      stackMask |= 1u << l;
    }
  }
  sp.tileBytes = (rowsMax * g.tileP + 15) & ~15;
  for (int l = 0; l < level0; l++) sp.levelStack[l] = 0;
  if (nStacks > 0) {
    if (dbgScore) {
      if (NC == 1) launch_stack_nc<true, 1>(g, p, sp, nStacks, nimg, cellCand, cellCount, dbgScore, s);
      else if (NC == 2) launch_stack_nc<true, 2>(g, p, sp, nStacks, nimg, cellCand, cellCount, dbgScore, s);
      else launch_stack_nc<true, 3>(g, p, sp, nStacks, nimg, cellCand, cellCount, dbgScore, s);
    } else {
      if (NC == 1) launch_stack_nc<false, 1>(g, p, sp, nStacks, nimg, cellCand, cellCount, dbgScore, s);
      else if (NC == 2) launch_stack_nc<false, 2>(g, p, sp, nStacks, nimg, cellCand, cellCount, dbgScore, s);
      else launch_stack_nc<false, 3>(g, p, sp, nStacks, nimg, cellCand, cellCount, dbgScore, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  // ---- the rest: one wave per cell, one launch per run of consecutive levels
  const size_t lds = cell_lds_bytes(g);
  const int listCap = std::min(g_detect_list_cap, kListTotal);
  for (int la = level0; la < level1;) {
    if (stackMask >> la & 1u) { la++; continue; }
    int lb = la;
    while (lb < level1 && !(stackMask >> lb & 1u)) lb++;
    const int cellBegin = g.lv[la].cellStart;
    const int cellEnd = lb < g.nlevels ? g.lv[lb].cellStart : g.totalCells;
    la = lb;
    if (cellEnd <= cellBegin) continue;
    dim3 grid(cellEnd - cellBegin, nimg);
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(64), lds, s, g, p, cellCand, cellCount, listCap, cellBegin, dbgScore); };
    const int tp = (!dbgScore && g.scoreP == g.tileP) ? g.tileP : 0;
    switch (tp) {
      case 44: go(k_detect<false, 44>); break;
      case 48: go(k_detect<false, 48>); break;
      case 52: go(k_detect<false, 52>); break;
      case 56: go(k_detect<false, 56>); break;
      default:
        if (dbgScore) go(k_detect<true, 0>); else go(k_detect<false, 0>);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <bool TAP, int NC>
static hipError_t prepare_stack_nc(int lds) {
  const void* ks[4] = {reinterpret_cast<const void*>(k_detect_stack<TAP, NC, 44>), reinterpret_cast<const void*>(k_detect_stack<TAP, NC, 48>),
                       reinterpret_cast<const void*>(k_detect_stack<TAP, NC, 52>), reinterpret_cast<const void*>(k_detect_stack<TAP, NC, 56>)};
  for (const void* f : ks) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// raises the dynamic-LDS limit of every detect instantiation (called from prepare_kernels, once per configure)
hipError_t prepare_detect(const Geom& g) {
  const size_t lds_det = cell_lds_bytes(g);
  const void* dk[6] = {reinterpret_cast<const void*>(k_detect<true, 0>),   reinterpret_cast<const void*>(k_detect<false, 0>),
                       reinterpret_cast<const void*>(k_detect<false, 44>), reinterpret_cast<const void*>(k_detect<false, 48>),
                       reinterpret_cast<const void*>(k_detect<false, 52>), reinterpret_cast<const void*>(k_detect<false, 56>)};
  for (const void* f : dk) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_det);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipSuccess;
  const int NC = detect_nc();
  // (an upper bound that does not depend on the image size: the limit is per device and must never go down)
  const int lds_stack = 3 * 128 * 56 + 2 * 4096 + 4096;
  if (NC == 1) { e = prepare_stack_nc<false, 1>(lds_stack); if (e == hipSuccess) e = prepare_stack_nc<true, 1>(lds_stack); }
  if (NC == 2) { e = prepare_stack_nc<false, 2>(lds_stack); if (e == hipSuccess) e = prepare_stack_nc<true, 2>(lds_stack); }
  if (NC == 3) { e = prepare_stack_nc<false, 3>(lds_stack); if (e == hipSuccess) e = prepare_stack_nc<true, 3>(lds_stack); }
  return e;
}

}  // namespace orbx
