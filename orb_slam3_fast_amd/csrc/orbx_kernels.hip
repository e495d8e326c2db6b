// orbx_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the ORB front-end.
//
// One kernel per stage of ORBextractor::operator() (src/ORBextractor.cc:1015-1106 of the reference); every
// kernel covers all images of the batch (and all pyramid levels where the stage allows) in one launch:
//   k_resize    ComputePyramid               :1108-1145  (cv::resize INTER_LINEAR 8U)
//   k_detect    per-cell cv::FAST + NMS + ini/min threshold fallback :892-971, one wave per cell
//   k_octree    DistributeOctTree            :557-757    one workgroup per (image, level), node tables in LDS
//   k_blur      GaussianBlur 7x7 sigma 2     :1074-1076  
//   k_slots     mono/lapping slot assignment :1062-1099  (serial-order semantics)
//   k_describe  IC_Angle + computeOrbDescriptor + output :75-147, one wave per keypoint, ballot-packed bits
// and of the matchers:
//   k_stereo*   Frame::ComputeStereoMatches  src/Frame.cc:921-1084
//   k_bf_knn2   BFMatcher::knnMatch(k=2)     src/Frame.cc:1293-1302
//   k_init_*    ORBmatcher::SearchForInitialization src/ORBmatcher.cc:618-764
//
// Integer/bitwise work: no MFMA.  Float steps that decide bits (fastAtan2, the rotated sampling
// coordinates, sub-pixel disparity) use IEEE ops without contraction (-ffp-contract=off for this TU) and
// round-half-even conversions, mirroring the x86-64 baseline (no FMA) build of the reference.
#include "orbx_internal.h"
#include "orbx_introsort.h"

namespace orbx {

__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern31.inc"
};
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

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

// ================================================================================================ resize
// cv::resize INTER_LINEAR 8U, level l from level l-1.  Coefficient tables (sx, a0/a1 ; sy, b0/b1)
// are built on the host exactly as OpenCV builds them and uploaded once per image size.
// Block = 256 dst columns x RS_DR dst rows.  The source footprint is staged in LDS with coalesced dword loads;
// the horizontal pass (one thread per dst column, walking the footprint rows) leaves (S0*a0 + S1*a1) >> 4 as u16
// in LDS; the vertical pass emits 4 pixels per thread with one u32 store.
#define RS_DW 256
#define RS_DR 16
__global__ __launch_bounds__(256) void k_resize(Geom g, Pyr p, int l, const int* __restrict__ xofs,
                                                const short* __restrict__ xab, const int* __restrict__ yofs,
                                                const short* __restrict__ yab, int srcRowsMax, int srcDwMax) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const LevelDev D = g.lv[l];
  const LevelDev S = g.lv[l - 1];
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  const int x0 = blockIdx.x * RS_DW, y0 = blockIdx.y * RS_DR;
  const int x1 = min(x0 + RS_DW, D.w) - 1, y1 = min(y0 + RS_DR, D.h) - 1;  // last dst column / row of the block
  int sp;
  const uint8_t* src = level_ptr(g, p, img, l - 1, sp);
  uint32_t* st = reinterpret_cast<uint32_t*>(smem);                            // [srcRowsMax][srcDwMax] dwords
  uint16_t* ht = reinterpret_cast<uint16_t*>(st + srcRowsMax * srcDwMax);      // [srcRowsMax][RS_DW] u16
#ifdef RS_PROF
  long long tq0 = wall_clock64();
#endif
  const int rb = min(max(yofs[D.ycoef + y0], 0), S.h - 1);                      // first source row needed
  const int re = min(max(yofs[D.ycoef + y1] + 1, 0), S.h - 1);                  // last source row needed
  const int nrows = re - rb + 1;
  const int cb = xofs[D.xcoef + x0] & ~3;                                       // first source byte (dword aligned)
  const int ce = min(xofs[D.xcoef + x1] + 1, S.w - 1);
  const int ndw = ((ce - cb) >> 2) + 1;
  {
    // Footprint -> LDS.  All of a thread's global loads are issued before the first LDS store: with a plain loop
    // every trip waited for its own load (7 serialized HBM latencies per block, the bulk of the kernel's time).
    const float inv = 1.0f / (float)ndw;
    const int n = nrows * ndw;
    constexpr int kTrips = 8;  // one batch covers the footprint at scale 1.2 (21 x 78 dwords); larger scales loop
    for (int base = 0; base < n; base += 256 * kTrips) {
    uint32_t v[kTrips];
    int slot[kTrips];
#pragma unroll
    for (int k = 0; k < kTrips; k++) {
      const int i = base + tid + 256 * k;
      slot[k] = -1;
      v[k] = 0;
      if (i < n) {
        const int r = (int)(((float)i + 0.5f) * inv), c = i - r * ndw;
        const int gx = cb + 4 * c;
        const uint8_t* q = src + (long long)(rb + r) * sp + gx;
        slot[k] = r * srcDwMax + c;
        if (gx + 4 <= S.w) {
          v[k] = *reinterpret_cast<const uint32_t*>(q);
        } else {
          for (int b = 0; b < 4; b++)
            if (gx + b < S.w) v[k] |= (uint32_t)q[b] << (8 * b);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kTrips; k++)
      if (slot[k] >= 0) st[slot[k]] = v[k];
    }
  }
#ifdef RS_PROF
  long long tq1 = wall_clock64();
#endif
  __syncthreads();
#ifdef RS_PROF
  long long tq2 = wall_clock64();
#endif
  {  // horizontal pass: lane = quad of 4 dst columns, wave = source-row phase (rows w, w + 4, ...).  Per output one
     // ds_read2_b32 (the aligned dword pair holding S[sx], S[sx+1]), one v_perm with a per-lane selector that spreads
     // the two bytes into u16 halves, one v_dot2_u32_u16 against (a0, a1), one shift; four results leave as one b64.
    const int qc = tid & 63, w = tid >> 6;
    uint32_t sel[4], coef[4];
    int dwi[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int dx = min(x0 + 4 * qc + j, D.w - 1);
      const int o = xofs[D.xcoef + dx] - cb;  // S[sx + 1] is only weighted by a1 != 0 when it exists (build_coefs)
      const int sh = o & 3;
      dwi[j] = o >> 2;
      sel[j] = (uint32_t)sh | 0x0c000c00u | ((uint32_t)(sh + 1) << 16);
      coef[j] = reinterpret_cast<const uint32_t*>(xab)[D.xcoef + dx];  // a0 | a1 << 16
    }
    for (int r = w; r < nrows; r += 4) {
      const uint32_t* row = st + r * srcDwMax;
      uint32_t t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t lo = row[dwi[j]], hi = row[dwi[j] + 1];
        t[j] = udot2_u16(__builtin_amdgcn_perm(hi, lo, sel[j]), coef[j], 0u) >> 4;
      }
      uint2 pk;
      pk.x = t[0] | (t[1] << 16);
      pk.y = t[2] | (t[3] << 16);
      *reinterpret_cast<uint2*>(ht + r * RS_DW + 4 * qc) = pk;
    }
  }
#ifdef RS_PROF
  long long tq3 = wall_clock64();
#endif
  __syncthreads();
#ifdef RS_PROF
  long long tq4 = wall_clock64();
#endif
  // vertical pass: lane = quad, wave w owns dst rows w, w + 4, ... of the block: the row constants are wave-uniform
  {
    const int qx = tid & 63;
    const int dx = x0 + 4 * qx;
    for (int dyl = __builtin_amdgcn_readfirstlane(tid >> 6); dyl < RS_DR; dyl += 4) {
      const int dy = y0 + dyl;
      if (dy >= D.h) break;
      const int sy = yofs[D.ycoef + dy];
      const uint32_t bb = reinterpret_cast<const uint32_t*>(yab)[D.ycoef + dy];
      const int b0 = (int)(bb & 0xFFFF), b1 = (int)(bb >> 16);
      const int r0 = min(max(sy, 0), S.h - 1) - rb, r1 = min(max(sy + 1, 0), S.h - 1) - rb;
      if (dx >= D.w) continue;
      const uint2 t0 = *reinterpret_cast<const uint2*>(ht + r0 * RS_DW + 4 * qx);
      const uint2 t1 = *reinterpret_cast<const uint2*>(ht + r1 * RS_DW + 4 * qx);
      const int u0[4] = {(int)(t0.x & 0xFFFF), (int)(t0.x >> 16), (int)(t0.y & 0xFFFF), (int)(t0.y >> 16)};
      const int u1[4] = {(int)(t1.x & 0xFFFF), (int)(t1.x >> 16), (int)(t1.y & 0xFFFF), (int)(t1.y >> 16)};
      uint32_t outw = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int v = (((b0 * u0[j]) >> 16) + ((b1 * u1[j]) >> 16) + 2) >> 2;
        outw |= (uint32_t)(v & 255) << (8 * j);
      }
      uint8_t* dst = p.pyr + (long long)img * g.pyrImg + D.off + (long long)dy * D.pitch;
      *reinterpret_cast<uint32_t*>(dst + dx) = outw;
    }
  }
#ifdef RS_PROF
  if (tid == 0 && blockIdx.z == 7 && blockIdx.x == 1 && (blockIdx.y % 9) == 3)
    printf("L%d by %d: load %d wait %d horiz %d wait %d vert %d (x10ns)\n", l, (int)blockIdx.y, (int)(tq1 - tq0), (int)(tq2 - tq1),
           (int)(tq3 - tq2), (int)(tq4 - tq3), (int)(wall_clock64() - tq4));
#endif
}

hipError_t launch_resize(const Geom& g, const Pyr& p, int nimg, int level, const int* xofs, const short* xab,
                         const int* yofs, const short* yab, hipStream_t s) {
  const LevelDev& D = g.lv[level];
  const LevelDev& S = g.lv[level - 1];
  // footprint bounds of a 256 x 8 dst block for this level's scale (+ slack for the floor/ceil of the taps)
  const int srcRowsMax = (int)((double)RS_DR * S.h / D.h) + 4;
  const int srcDwMax = ((int)((double)RS_DW * S.w / D.w) + 12) / 4 + 1;
  const size_t lds = (size_t)srcRowsMax * srcDwMax * 4 + (size_t)srcRowsMax * RS_DW * 2;
  dim3 grid((D.w + RS_DW - 1) / RS_DW, (D.h + RS_DR - 1) / RS_DR, nimg);
  hipLaunchKernelGGL(k_resize, grid, dim3(256), lds, s, g, p, level, xofs, xab, yofs, yab, srcRowsMax, srcDwMax);
  return hipGetLastError();
}

// ================================================================================================ detect
// FAST-9-16 on an LDS tile, four horizontally adjacent pixels per lane.
//
// Layout: the cell ROI (cell + 3 px FAST halo each side) sits in LDS with ROI column 0 on a dword boundary
// (the loader funnel-shifts the unaligned global row).  A lane owns a "quad" of 4 detectable pixels; the 7x10
// byte neighbourhood it needs is 7 rows x 3 dwords, read with 21 ds_read_b32 and kept in registers, so every
// ring byte is a compile-time (register, byte) pair.  Per ring pixel and polarity: one subtract and one
// v_alignbit funnel shift that appends the sign bit to a 16-bit arc mask; a 9-arc exists iff the doubled
// mask has 9 contiguous ones (shift-and ladder).
constexpr int kRingDX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
constexpr int kRingDY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
constexpr int kListCap = 640;    // upper bound of the test hook orbx_debug_set_detect_list_cap
constexpr int kSurvCap = 448;    // LDS list of compass-test survivors of k_detect (flushed before it would overflow)
constexpr int kCornerCap = 256;  // LDS corner list (a cell with more corners takes the tile-scan NMS)

__device__ __forceinline__ bool has_arc9(uint32_t m) {  // 9 contiguous set bits in a circular 16-bit mask
  uint32_t d = m | (m << 16);
  uint32_t x = d & (d >> 1);
  x &= x >> 2;
  x &= x >> 4;
  x &= d >> 8;
  return (x & 0xFFFFu) != 0;
}

// Necessary condition for a 9-arc: two cyclically adjacent compass pixels (ring 0, 4, 8, 12) of the same
// polarity.  v_cmp leaves each comparison as a 64-lane mask in SGPRs, so the combination is scalar-ALU work:
// per pixel slot 8 VALU compares + 7 scalar ops.  Returns the wave mask of lanes whose pixel survives.
template <int P>
__device__ __forceinline__ uint64_t compass_wave(const uint32_t (&r)[7][3], int t) {
  const int c = (r[3][(3 + P) >> 2] >> (8 * ((3 + P) & 3))) & 0xFF;
  const int hi = c + t, lo = c - t;
  uint64_t B[4], D[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int k = 4 * q;
    const int col = 3 + P + kRingDX[k];
    const int v = (r[3 + kRingDY[k]][col >> 2] >> (8 * (col & 3))) & 0xFF;
    B[q] = __ballot(v > hi);
    D[q] = __ballot(v < lo);
  }
  // two cyclically adjacent compass points set  <=>  one of {0, 2} and one of {1, 3} set (in a 4-cycle every even
  // position is adjacent to every odd one): 7 scalar ops instead of 15
  return ((B[0] | B[2]) & (B[1] | B[3])) | ((D[0] | D[2]) & (D[1] | D[3]));
}

// FAST contrast of one pixel from the LDS tile: M = max over the 16 nine-pixel arcs of the arc's minimum
// one-signed contrast = max( max_s min(arc_s) - c , c - min_s max(arc_s) ).  Sliding 9-windows are built from
// 3-windows (v_min3 / v_max3): 16 + 16 + 8 three-input ops per polarity.  The pixel is a corner at threshold t
// iff M > t, and its cornerScore is M - 1 — one pass gives both the decision and the score.
__device__ __forceinline__ int fast_contrast_lds(const uint8_t* c8, int TP) {
  int r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) r[k] = c8[kRingDY[k] * TP + kRingDX[k]];
  int lo3[16], hi3[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    lo3[i] = min(min(r[i], r[(i + 1) & 15]), r[(i + 2) & 15]);
    hi3[i] = max(max(r[i], r[(i + 1) & 15]), r[(i + 2) & 15]);
  }
  int lo9[16], hi9[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    lo9[i] = min(min(lo3[i], lo3[(i + 3) & 15]), lo3[(i + 6) & 15]);
    hi9[i] = max(max(hi3[i], hi3[(i + 3) & 15]), hi3[(i + 6) & 15]);
  }
  int maxmin = lo9[0], minmax = hi9[0];
#pragma unroll
  for (int i = 1; i < 16; i++) {
    maxmin = max(maxmin, lo9[i]);
    minmax = min(minmax, hi9[i]);
  }
  const int c = c8[0];
  return max(maxmin - c, c - minmax);
}

// Two pixels per lane: the same contrast computation in packed half precision.  A pixel value v (0..255) is used
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
__device__ __forceinline__ orbx_h2 fast_contrast2_lds(const uint8_t* a8, const uint8_t* b8, int TP) {
  orbx_h2 r[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int off = kRingDY[k] * TP + kRingDX[k];
    r[k] = __builtin_bit_cast(orbx_h2, (uint32_t)a8[off] | ((uint32_t)b8[off] << 16));  // one v_perm
  }
  orbx_h2 lo3[16], hi3[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    lo3[i] = pk_min3(r[i], r[(i + 1) & 15], r[(i + 2) & 15]);
    hi3[i] = pk_max3(r[i], r[(i + 1) & 15], r[(i + 2) & 15]);
  }
  orbx_h2 lo9[16], hi9[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    lo9[i] = pk_min3(lo3[i], lo3[(i + 3) & 15], lo3[(i + 6) & 15]);
    hi9[i] = pk_max3(hi3[i], hi3[(i + 3) & 15], hi3[(i + 6) & 15]);
  }
  orbx_h2 maxmin = pk_max3(lo9[0], lo9[1], lo9[2]), minmax = pk_min3(hi9[0], hi9[1], hi9[2]);
#pragma unroll
  for (int i = 3; i < 15; i += 2) {
    maxmin = pk_max3(maxmin, lo9[i], lo9[i + 1]);
    minmax = pk_min3(minmax, hi9[i], hi9[i + 1]);
  }
  maxmin = __builtin_elementwise_maximum(maxmin, lo9[15]);
  minmax = __builtin_elementwise_minimum(minmax, hi9[15]);
  const orbx_h2 c = __builtin_bit_cast(orbx_h2, (uint32_t)a8[0] | ((uint32_t)b8[0] << 16));
  return __builtin_elementwise_maximum(maxmin - c, c - minmax);
}

// XCD-aware block -> tile mapping.  Workgroups go to the 8 XCDs round-robin by flat workgroup id, so neighbouring
// tiles land in 8 different L2s and every shared halo line is fetched from HBM once per XCD.  This permutation keeps
// the dispatch order balanced (every aligned chunk of 8*K flat ids still covers the same 8*K tiles) but hands each
// XCD a run of K consecutive tiles.  `by` is the slower grid index (image); chunks cut by an image boundary keep
// the identity order.
__device__ __forceinline__ int xcd_run_remap(int bx, int nbx, int by, int K) {
  if (K <= 1) return bx;
  const int o = (int)(((unsigned)by * (unsigned)nbx) & 7u), xs = bx + o, ch = 8 * K;
  const int c0 = (xs / ch) * ch;
  if (c0 < o || c0 + ch > nbx + o) return bx;
  const int r = xs - c0;
  return c0 + (r & 7) * K + (r >> 3) - o;
}

// One wave per FAST cell (workgroup = 64 threads, so __syncthreads() is a wave barrier).
// Pass 1 runs at iniThFAST; only a cell whose post-NMS set is empty is redone at minThFAST (:942-959).
// The NMS needs no threshold masking: a neighbour that is not a corner at t has score < t <= the centre's.
// Output: no atomics.  Every cell owns cellCap slots of the sparse store (an NMS survivor set has at most
// ceil(w/2)*ceil(h/2) members) and writes its count; k_octree scans the counts and compacts.
__global__ __launch_bounds__(64) void k_detect(Geom g, Pyr p, uint32_t* __restrict__ cellCand,
                                               int* __restrict__ cellCount, int ablate, int listCap,
                                               int cellBegin, int xcdRun) {
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
  int cell = cellBegin + xcd_run_remap(blockIdx.x, gridDim.x, blockIdx.y, xcdRun);
  int l = 0;
  while (l + 1 < g.nlevels && cell >= g.lv[l + 1].cellStart) l++;
  const LevelDev L = g.lv[l];
  int* myCount = cellCount + (long long)img * g.totalCells + cell;
  cell -= L.cellStart;
  const int ci = cell / L.nCols, cj = cell - ci * L.nCols;
  const int maxBX = L.w - kBorder, maxBY = L.h - kBorder;
  const int iniY = kBorder + ci * L.hCell, iniX = kBorder + cj * L.wCell;
  const int maxY = min(iniY + L.hCell + 6, maxBY), maxX = min(iniX + L.wCell + 6, maxBX);
  const int rw = maxX - iniX, rh = maxY - iniY;
  const int dw = rw - 6, dh = rh - 6;  // detectable window of the cell (FAST needs a 3 px ring)
  if (iniY >= maxBY - 3 || iniX >= maxBX - 6 || dw <= 0 || dh <= 0) {  // src/ORBextractor.cc:913,919
    if (lane == 0) *myCount = 0;
    return;
  }

  const int TPd = g.tileP >> 2, SPd = g.scoreP >> 2;  // pitches in dwords
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem);
  uint32_t* score = tile + TPd * g.tileH;
  uint8_t* score8 = reinterpret_cast<uint8_t*>(score);
  uint16_t* list = reinterpret_cast<uint16_t*>(score + SPd * g.scoreH);  // kCornerCap corner positions (y << 8 | x)
  uint16_t* slist = list + kCornerCap;                                   // kSurvCap compass-test survivors
  const int survCap = min(listCap, kSurvCap), cornerCap = min(listCap, kCornerCap);
  const uint8_t* tile8 = reinterpret_cast<const uint8_t*>(tile);
  const int qpr = (dw + 3) >> 2;  // quads per detect row
  const int nq = qpr * dh;
  const float inv_qpr = 1.0f / (float)qpr;

  int pitch;
  const uint8_t* im = level_ptr(g, p, img, l, pitch);
  if (ablate & 16) return;
  if (!(ablate & 1)) {  // tile load: ROI column 0 -> LDS byte 0 of the row (funnel shift of two aligned global dwords)
    // lane = (row phase, dword column): 16 columns x 4 rows per pass, no index divisions in the loop
    const int mis = iniX & 3, xa = iniX - mis;
    const int dpr = (rw + 3) >> 2;
    const int r0 = lane >> 4;
    for (int cc = lane & 15; cc < dpr; cc += 16) {  // one trip unless the cell is wider than 58 px (tiny levels)
      const int gx = xa + 4 * cc;
      const bool wide = gx + 8 <= L.w;
      const uint8_t* src = im + (long long)(iniY + r0) * pitch + gx;
      uint32_t* dstw = tile + r0 * TPd + cc;
      for (int r = r0; r < rh; r += 4, src += 4 * (long long)pitch, dstw += 4 * TPd) {
        uint32_t lo, hi = 0;
        if (wide) {
          lo = reinterpret_cast<const uint32_t*>(src)[0];
          hi = reinterpret_cast<const uint32_t*>(src)[1];
        } else {
          uint64_t v = 0;
          for (int k = 0; k < 8; k++)
            if (gx + k < L.w) v |= (uint64_t)src[k] << (8 * k);
          lo = (uint32_t)v;
          hi = (uint32_t)(v >> 32);
        }
        *dstw = __builtin_amdgcn_alignbyte(hi, lo, mis);
      }
    }
    // zero ring of the score tile: rows 0 and dh+1, dword columns 0 and qpr+1
    for (int idx = lane; idx < SPd; idx += 64) {
      score[idx] = 0;
      score[(dh + 1) * SPd + idx] = 0;
    }
    for (int idx = lane; idx < dh; idx += 64) {
      score[(idx + 1) * SPd] = 0;
      score[(idx + 1) * SPd + qpr + 1] = 0;
    }
  }
  __syncthreads();

  uint32_t* out = cellCand + (long long)img * g.cellImg + L.cellOff + (long long)cell * L.cellCap;
  int kept = 0;
  for (int pass = 0; pass < 2; pass++) {
    const int t = pass == 0 ? g.iniTh : g.minTh;
    // dense corner test, 4 pixels per lane; corners are compacted into an LDS list and scored with dense
    // lanes (the score needs ~110 min/max ops: running it under per-lane divergence would dominate).
    // Stage 1 (4 pixels per lane, registers): compass pre-test -> survivor list.
    // Stage 2 (dense lanes over survivors): contrast M from the 16 ring pixels; corner iff M > t, score M - 1
    //          goes to the u8 score tile and the corner to the corner list (for the list-based NMS).
    int nList = 0, nSurv = 0;
    bool overflowed = false;  // more corners than the list holds: the NMS falls back to scanning the score tile
    auto flush_survivors = [&]() {
      __syncthreads();
      const orbx_h2 th2 = __builtin_bit_cast(orbx_h2, (uint32_t)t * 0x00010001u);  // t in the same subnormal encoding
      for (int base = 0; base < nSurv; base += 128) {  // two survivors per lane (packed f16 contrast)
        const int eA = base + lane, eB = base + 64 + lane;
        const int yxA = slist[min(eA, nSurv - 1)], yxB = slist[min(eB, nSurv - 1)];
        const int yA = yxA >> 8, xA = yxA & 255, yB = yxB >> 8, xB = yxB & 255;
        const orbx_h2 M = fast_contrast2_lds(tile8 + (yA + 3) * g.tileP + xA + 3, tile8 + (yB + 3) * g.tileP + xB + 3,
                                            g.tileP);
        const bool cornerA = eA < nSurv && M.x > th2.x, cornerB = eB < nSurv && M.y > th2.y;
        const uint32_t Mbits = __builtin_bit_cast(uint32_t, M);  // a corner has M > t >= 0: the pattern is the integer
        if (cornerA) score8[(yA + 1) * g.scoreP + xA + 4] = (uint8_t)((Mbits & 0xFFFFu) - 1);
        if (cornerB) score8[(yB + 1) * g.scoreP + xB + 4] = (uint8_t)((Mbits >> 16) - 1);
        const uint64_t mA = __ballot(cornerA), mB = __ballot(cornerB);
        const int oA = nList + prefix_count(mA);
        const int oB = nList + __popcll(mA) + prefix_count(mB);
        if (cornerA && oA < cornerCap) list[oA] = (uint16_t)yxA;
        if (cornerB && oB < cornerCap) list[oB] = (uint16_t)yxB;
        nList += __popcll(mA) + __popcll(mB);
      }
      if (nList > cornerCap) {
        overflowed = true;
        nList = cornerCap;
      }
      __syncthreads();
      nSurv = 0;
    };
    const int nq_round = (nq + 63) & ~63;
    for (int q = lane; q < nq_round; q += 64) {
      const bool act = q < nq && !(ablate & 2);
      const int qq = min(q, nq - 1);  // idle lanes of the last round redo the last quad (masked out below)
      const int yd = (int)(((float)qq + 0.5f) * inv_qpr);
      const int j = qq - yd * qpr;
      uint32_t r[7][3];
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const uint32_t* row = tile + (yd + i) * TPd + j;
        r[i][0] = row[0];
        r[i][1] = row[1];
        r[i][2] = row[2];
      }
      const int valid = dw - 4 * j;  // pixels of this quad inside the detectable window
      uint64_t sm[4];
      sm[0] = compass_wave<0>(r, t) & __ballot(act);
      sm[1] = compass_wave<1>(r, t) & __ballot(act && valid > 1);
      sm[2] = compass_wave<2>(r, t) & __ballot(act && valid > 2);
      sm[3] = compass_wave<3>(r, t) & __ballot(act && valid > 3);
      if (act) score[(yd + 1) * SPd + j + 1] = 0;
      // flush first when this round's survivors (<= 256) would not fit: the list then only needs room for a typical cell
      if (nSurv + (int)(__popcll(sm[0]) + __popcll(sm[1]) + __popcll(sm[2]) + __popcll(sm[3])) > survCap) flush_survivors();
#pragma unroll
      for (int pI = 0; pI < 4; pI++) {
        const uint64_t m = sm[pI];
        if (__builtin_amdgcn_inverse_ballot_w64(m))  // this lane's bit of the SGPR mask, without a 64-bit vector shift
          slist[nSurv + prefix_count(m)] = (uint16_t)((yd << 8) | (4 * j + pI));
        nSurv += __popcll(m);
      }
    }
    flush_survivors();
    const int nCorners = nList;
    // 3x3 non-max suppression (strict '>') inside the cell + emission
    if (ablate & 4) kept = 1;
    if (!(ablate & 4) && !overflowed) {
      // common case: every corner of the cell is still in the list -> dense lanes, 9 LDS byte reads each
      const int SP = g.scoreP;
      for (int base = 0; base < nCorners; base += 64) {
        const int e = base + lane;
        const int yx = list[min(e, nCorners - 1)], y = yx >> 8, x = yx & 255;
        const uint8_t* c8 = score8 + (y + 1) * SP + x + 4;
        const int sc = c8[0];
        const bool keep = e < nCorners && sc > c8[-1] && sc > c8[1] && sc > c8[-SP - 1] && sc > c8[-SP] &&
                          sc > c8[-SP + 1] && sc > c8[SP - 1] && sc > c8[SP] && sc > c8[SP + 1];
        const uint64_t m = __ballot(keep);
        if (keep) {
          const int o = kept + prefix_count(m);
          if (o < L.cellCap) out[o] = pack_key(iniX + 3 + x - kBorder, iniY + 3 + y - kBorder, sc);
        }
        kept += __popcll(m);
      }
    } else if (!(ablate & 4))
    for (int q = lane; q < nq_round; q += 64) {
      uint32_t sw = 0;
      int yd = 0, j = 0;
      if (q < nq) {
        yd = (int)(((float)q + 0.5f) * inv_qpr);
        j = q - yd * qpr;
        sw = score[(yd + 1) * SPd + j + 1];
      }
      uint32_t keepmask = 0;
      if (sw) {
        const int SP = g.scoreP;
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
  if (lane == 0) *myCount = (ablate & 4) ? 0 : min(kept, L.cellCap);
}

static int g_detect_list_cap = kListCap;
void debug_set_detect_list_cap(int cap) { g_detect_list_cap = cap < 320 ? 320 : (cap > kListCap ? kListCap : cap); }

// Cells of levels [level0, level1) only: level 0 needs no resize and is launched beside the pyramid chain.
hipError_t launch_detect(const Geom& g, const Pyr& p, int nimg, uint32_t* cellCand, int* cellCount, int level0,
                         int level1, hipStream_t s) {
  const size_t lds = (size_t)g.tileP * g.tileH + (size_t)g.scoreP * g.scoreH + 2 * (kSurvCap + kCornerCap) + 16;
  const int cellBegin = g.lv[level0].cellStart;
  const int cellEnd = level1 < g.nlevels ? g.lv[level1].cellStart : g.totalCells;
  if (cellEnd <= cellBegin) return hipSuccess;
  dim3 grid(cellEnd - cellBegin, nimg);
  static const int ablate = getenv("ORBX_DETECT_ABLATE") ? atoi(getenv("ORBX_DETECT_ABLATE")) : 0;
  static const int xcdRun = getenv("ORBX_DETECT_XCD_RUN") ? atoi(getenv("ORBX_DETECT_XCD_RUN")) : 8;
  hipLaunchKernelGGL(k_detect, grid, dim3(64), lds, s, g, p, cellCand, cellCount, ablate, g_detect_list_cap,
                     cellBegin, xcdRun);
  return hipGetLastError();
}

// ================================================================================================ octree
// DistributeOctTree without moving keys: every candidate keeps a node id (knode), a pass counts the keys of
// each child quadrant with LDS atomics, and the node list of the next pass is laid out by prefix sums in
// exactly the order the reference's std::list ends up in (children push_front'ed as n1..n4 => a visited
// node's children appear as n4,n3,n2,n1, later-visited nodes first; untouched nodes keep their order).
// The final largest-first expansion sorts with the libstdc++ introsort replica (orbx_introsort.h).

// ---- wave-cooperative exact replica of std::sort -----------------------------------------------------------
// Same result as introsort() (orbx_introsort.h) — verified against std::sort on the CPU model of this
// formulation and on the device (orbx_debug_introsort_device) — but the partition and the final insertion sort
// are data parallel:
//  * Hoare partition with an unguarded pivot == "pair the k-th element >= pivot from the left with the k-th
//    element <= pivot from the right and swap them while they have not crossed"; the cut is
//    min(L[K], R[K-1]).  Both stopper lists come from ballot-compaction over the range.
//  * __final_insertion_sort is a stable sort of an array whose elements are at most 16 positions away from
//    home, i.e. a stable rank inside a +-16 window.
// All 64 lanes of ONE wave call it with uniform arguments; `a`, the scratch arrays and the stack are in LDS.
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ int partition_wave(uint64_t* a, int first, int last, uint16_t* Li, uint16_t* Ri, int lane) {
  const KeyLess less;
  const int mid = first + (last - first) / 2;
  {
    const uint64_t va = a[first + 1], vb = a[mid], vc = a[last - 1];
    int sel;
    if (less(va, vb)) sel = less(vb, vc) ? mid : (less(va, vc) ? last - 1 : first + 1);
    else if (less(va, vc)) sel = first + 1;
    else sel = less(vb, vc) ? last - 1 : mid;
    if (lane == 0) {
      const uint64_t t = a[first];
      a[first] = a[sel];
      a[sel] = t;
    }
  }
  wsync();
  const uint64_t pivot = a[first];
  int nL = 0, nR = 0;
  for (int base = first + 1; base < last; base += 64) {
    const int i = base + lane;
    const bool st = i < last && !less(a[min(i, last - 1)], pivot);
    const uint64_t m = __ballot(st);
    if (st) Li[nL + prefix_count(m)] = (uint16_t)i;
    nL += __popcll(m);
  }
  for (int top = last - 1; top >= first + 1; top -= 64) {
    const int i = top - lane;
    const bool st = i >= first + 1 && !less(pivot, a[max(i, first + 1)]);
    const uint64_t m = __ballot(st);
    if (st) Ri[nR + prefix_count(m)] = (uint16_t)i;
    nR += __popcll(m);
  }
  wsync();
  const int nmin = min(nL, nR);
  int K = 0;
  for (int base = 0; base < nmin; base += 64) {
    const int k = base + lane;
    K += __popcll(__ballot(k < nmin && Li[min(k, nmin - 1)] < Ri[min(k, nmin - 1)]));
  }
  const int INF = 1 << 30;
  const int lk = K < nL ? (int)Li[K] : INF, rk = K > 0 ? (int)Ri[K - 1] : INF;
  for (int k = lane; k < K; k += 64) {
    const int i = Li[k], j = Ri[k];
    const uint64_t t = a[i];
    a[i] = a[j];
    a[j] = t;
  }
  wsync();
  return min(lk, rk);
}

__device__ __forceinline__ void introsort_wave(uint64_t* a, int n, uint64_t* tmp, uint16_t* Li, uint16_t* Ri, int* stk, int lane) {
  if (n <= 1) return;
  const KeyLess less;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  int sp = 0;
  if (lane == 0) {
    stk[0] = 0;
    stk[1] = n;
    stk[2] = 2 * lg;
  }
  sp = 1;
  wsync();
  while (sp > 0) {
    --sp;
    int first = stk[3 * sp], last = stk[3 * sp + 1], depth = stk[3 * sp + 2];
    wsync();
    while (last - first > 16) {
      if (depth == 0) {
        if (lane == 0) is_heapsort<uint64_t, KeyLess>(a, first, last, less);
        wsync();
        break;
      }
      --depth;
      const int cut = partition_wave(a, first, last, Li, Ri, lane);
      if (lane == 0) {
        stk[3 * sp] = cut;
        stk[3 * sp + 1] = last;
        stk[3 * sp + 2] = depth;
      }
      ++sp;
      wsync();
      last = cut;
    }
  }
  // __final_insertion_sort == stable rank inside a +-16 window
  for (int i = lane; i < n; i += 64) {
    const uint64_t vi = a[i];
    const int w0 = max(0, i - 16), w1 = min(n, i + 17);
    int c = 0;
    for (int j = w0; j < w1; j++) {
      const uint64_t vj = a[j];
      c += (less(vj, vi) || (!less(vi, vj) && j < i)) ? 1 : 0;
    }
    tmp[w0 + c] = vi;
  }
  wsync();
  for (int i = lane; i < n; i += 64) a[i] = tmp[i];
  wsync();
}

// Test entry: sort n elements with the wave version (one block, dynamic LDS).
__global__ __launch_bounds__(64) void k_debug_sort(uint64_t* v, int n) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);
  uint64_t* tmp = a + n;
  uint16_t* Li = reinterpret_cast<uint16_t*>(tmp + n);
  uint16_t* Ri = Li + n + 4;
  int* stk = reinterpret_cast<int*>(Ri + n + 4 + ((n & 1) ? 1 : 0) + 2);
  const int lane = threadIdx.x;
  for (int i = lane; i < n; i += 64) a[i] = v[i];
  __syncthreads();
  introsort_wave(a, n, tmp, Li, Ri, stk, lane);
  __syncthreads();
  for (int i = lane; i < n; i += 64) v[i] = a[i];
}
hipError_t launch_debug_sort(uint64_t* d_v, int n, hipStream_t s) {
  const size_t lds = (size_t)n * 16 + (size_t)(2 * n + 16) * 2 + 3 * 64 * 4 + 64;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_debug_sort),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3(64), lds, s, d_v, n);
  return hipGetLastError();
}

struct OctLds {  // byte offsets into dynamic LDS, all 8-byte aligned
  int nx0[2], nx1[2], ny0[2], ny1[2], ncnt[2];
  int cnt4, cntr, cpos, scan, e[2], mark, bestk, tsum, cellpre;
  int total;
};
__host__ __device__ inline OctLds oct_layout(int maxn, int maxcells, int rep) {
  OctLds o;
  int off = 0;
  auto take = [&](int bytes) {
    int r = off;
    off += (bytes + 7) & ~7;
    return r;
  };
  for (int b = 0; b < 2; b++) {
    o.nx0[b] = take(maxn * 2);
    o.nx1[b] = take(maxn * 2);
    o.ny0[b] = take(maxn * 2);
    o.ny1[b] = take(maxn * 2);
    o.ncnt[b] = take(maxn * 4);
  }
  o.cnt4 = take(maxn * 16);
  o.cntr = take(maxn * 16 * rep);  // quadrant counters, `rep` replicas each (see OctCtx::cntr)
  o.cpos = take(maxn * 8);
  o.scan = take(maxn * 8);  // u64 scan values; reused as best[] at the end
  o.e[0] = take(maxn * 8);
  o.e[1] = take(maxn * 8);
  o.mark = take(maxn * 2);
  o.bestk = take(maxn * 4);
  o.tsum = take(256 * 8 + 64);
  o.cellpre = take((maxcells + 1) * 4);
  o.total = off;
  return o;
}
__host__ __device__ inline int oct_maxn(const Geom& g) {
  int q = 0;
  for (int l = 0; l < g.nlevels; l++) q = g.lv[l].quota > q ? g.lv[l].quota : q;
  return q + 4 * kMaxIni + 8;
}
__host__ __device__ inline int oct_maxcells(const Geom& g) {
  int c = 0;
  for (int l = 0; l < g.nlevels; l++) c = g.lv[l].nCols * g.lv[l].nRows > c ? g.lv[l].nCols * g.lv[l].nRows : c;
  return c;
}
// Counter replicas: 4 when two blocks still fit one CU's 160 KB of LDS, fewer for very large per-level quotas.
__host__ __device__ inline int oct_rep(const Geom& g) {
  const int maxn = oct_maxn(g), maxcells = oct_maxcells(g);
  if (oct_layout(maxn, maxcells, 4).total <= 78 * 1024) return 4;
  if (oct_layout(maxn, maxcells, 2).total <= 78 * 1024) return 2;
  return 1;
}
size_t octree_lds_bytes(const Geom& g) { return (size_t)oct_layout(oct_maxn(g), oct_maxcells(g), oct_rep(g)).total; }

constexpr int OCT_NT = 512;  // threads per quadtree block: halves the key-loop trip counts vs 256, 2 blocks/CU stay resident
// Exclusive scan of n u64 values in LDS (in place) by an OCT_NT-thread block; returns the total.
// Packed fields must not overflow into each other (callers keep each field < 2^21).
// Per-thread chunk sums are scanned inside each wave with DPP/bpermute shuffles (no barriers); only the four
// wave totals go through LDS: 2 barriers per call instead of the 18 of a Hillis-Steele scan over 256 threads.
__device__ __forceinline__ uint64_t block_scan_u64(uint64_t* v, int n, uint64_t* tsum) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n + OCT_NT - 1) / OCT_NT;
  const int b = min(tid * per, n), e = min(b + per, n);
  uint64_t s = 0;
  for (int i = b; i < e; i++) s += v[i];
  uint64_t incl = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up((unsigned long long)incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) tsum[wv] = incl;
  __syncthreads();
  uint64_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < OCT_NT / 64; w++) {
    const uint64_t t = tsum[w];
    if (w < wv) wbase += t;
    total += t;
  }
  uint64_t run = wbase + incl - s;
  for (int i = b; i < e; i++) {
    const uint64_t t = v[i];
    v[i] = run;
    run += t;
  }
  __syncthreads();
  return total;
}

__device__ __forceinline__ int quadrant(int x, int y, int x0, int x1, int y0, int y1) {
  const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;  // ceil(float(len) / 2), :494-495
  return (x < x0 + hx ? 0 : 1) | (y < y0 + hy ? 0 : 2);
}

// Candidates of one (image, level) as seen by the quadtree passes.  REG: every thread keeps its candidates
// k = tid + j * OCT_NT (key and current node id) in registers for the whole kernel, so the ~12 passes over the
// candidate set touch only registers and LDS (the passes were latency-bound on L2 round trips: 163 -> see DESIGN).
// !REG (more than OCT_KMAX * OCT_NT candidates): keys / node ids live in global memory, same thread <-> k mapping.
constexpr int OCT_KMAX = 32;
constexpr int OCT_GROUP = 4;  // candidates per thread processed together in a sweep
template <bool REG>
struct OctCands {
  uint32_t rk[REG ? OCT_KMAX : 1];
  uint32_t rn2[REG ? OCT_KMAX / 2 : 1];  // node ids, two 16-bit ids per register (node | quadrant << 14, or kNoCand)
  uint32_t* keys;
  uint16_t* kn;
  int n;
  static constexpr uint32_t kNoCand = 0xFFFFu;
  template <class F>
  __device__ __forceinline__ void sweep(F f) {  // f(k, key, node&)
    if constexpr (REG) {
#pragma unroll
      for (int j0 = 0; j0 < OCT_KMAX; j0 += OCT_GROUP) {  // (no early exit: its phi copies double the live arrays)
#pragma unroll
        for (int j = j0; j < j0 + OCT_GROUP; j++)  // padding entries carry the sentinel node id (a per-j `k < n`
        {                                          // test would be hoisted out of every pass: 64 live SGPRs)
          uint32_t key = rk[j], nd = (rn2[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
          asm volatile("" : "+v"(key), "+v"(nd));  // opaque: keeps LICM from hoisting per-key values (key_x, key_y,
          if (nd != kNoCand) {                     // validity) of all 32 slots out of the passes
            f(0, key, nd);
            rn2[j >> 1] = (j & 1) ? (rn2[j >> 1] & 0xFFFFu) | (nd << 16) : (rn2[j >> 1] & 0xFFFF0000u) | nd;
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the groups apart: interleaving all 32 costs > 200 VGPRs
      }
    } else {
      for (int k = threadIdx.x; k < n; k += OCT_NT) {
        uint32_t nd = kn[k];
        const uint32_t nd0 = nd;
        f(k, keys[k], nd);
        if (nd != nd0) kn[k] = (uint16_t)nd;
      }
    }
  }
  template <class F>
  __device__ __forceinline__ void fill(F f) {  // key = f(k), node = 0
    if constexpr (REG) {
#pragma unroll
      for (int j = 0; j < OCT_KMAX; j++) {  // all gather loads are issued before the first use (one latency, not 8)
        const int k = (int)threadIdx.x + j * OCT_NT;
        rk[j] = 0;
        if (k < n) rk[j] = f(k);
      }
#pragma unroll
      for (int j = 0; j < OCT_KMAX; j++) {
        const int k = (int)threadIdx.x + j * OCT_NT;
        const uint32_t nd = k < n ? 0u : kNoCand;
        if (k < n) keys[k] = rk[j];  // kept for orbx_debug_candidates
        rn2[j >> 1] = (j & 1) ? rn2[j >> 1] | (nd << 16) : nd;
      }
    } else {
      for (int k = threadIdx.x; k < n; k += OCT_NT) {
        keys[k] = f(k);
        kn[k] = 0;
      }
    }
  }
};

struct OctCtx {  // node tables are double buffered: buffer b of table T sits at T + b * nodeStride bytes
  int16_t *nx0, *nx1, *ny0, *ny1;
  uint32_t* ncnt;
  int nodeStride;
  uint32_t* cnt4;
  int nrep;        // counter replicas (1, 2 or 4)
  uint32_t* cntr;  // [node * 4 + q][nrep]: a lane adds to replica (lane % nrep) -- a wave's candidates fall into 1-4 nodes,
                   // so unreplicated ds_add_u32 serialise 64 deep on one address; the replicas sit in different banks
  uint16_t* cpos;
  uint64_t* scan;
  uint64_t* ebuf;  // two buffers of maxn entries
  uint16_t* mark;
  uint32_t* nmid;  // per node of the current list: split point x | y << 12, bit 31 = "this pass splits the node"
  uint64_t* tsum;
  int* cellpre;
  int* s_i;
  int maxn;
};

template <bool REG>
__device__ __forceinline__ void octree_body(const Geom& g, const LevelDev& L, const OctCtx& c, int n, int cells,
                                            const uint32_t* __restrict__ sparse, uint32_t* __restrict__ keys,
                                            uint16_t* __restrict__ kn, uint32_t* __restrict__ out,
                                            int* __restrict__ outCount, int ablate) {
#define OCT_EXIT(stage) if (ablate == stage) { if (threadIdx.x == 0) *outCount = 0; return; }
  const int tid = threadIdx.x;
#ifdef OCT_PROF  // section timing of one block (tools/octree_prof.py; build with -DOCT_PROF, select the level with
                 // ORBX_OCTREE_ABLATE=100+level): thread 0 prints the 10 ns ticks between the MK() markers
  long long tmk[96]; int nmk = 0;
#define MK() do { if (nmk < 96) tmk[nmk++] = wall_clock64(); } while (0)
#else
#define MK() do {} while (0)
#endif
  MK();
  OctCands<REG> cd;
  cd.keys = keys;
  cd.kn = kn;
  cd.n = n;
  // dense index k -> (cell, i) by binary search over the LDS-resident prefix: balanced, no per-cell loops
  cd.fill([&](int k) {
    int lo = 0, hi = cells;  // largest cell with cellpre[cell] <= k
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (c.cellpre[mid] <= k) lo = mid; else hi = mid;
    }
    return sparse[(long long)lo * L.cellCap + (k - c.cellpre[lo])];
  });
  MK();
  OCT_EXIT(1)
  const int N = L.quota;
  struct Buf {  // nx0[b][i] etc. as before, by address arithmetic (no pointer arrays -> no scratch)
    char* base;
    int stride;
    __device__ __forceinline__ int16_t* operator[](int b) const { return reinterpret_cast<int16_t*>(base + b * stride); }
  };
  struct BufU {
    char* base;
    int stride;
    __device__ __forceinline__ uint32_t* operator[](int b) const { return reinterpret_cast<uint32_t*>(base + b * stride); }
  };
  const Buf nx0{(char*)c.nx0, c.nodeStride}, nx1{(char*)c.nx1, c.nodeStride}, ny0{(char*)c.ny0, c.nodeStride},
      ny1{(char*)c.ny1, c.nodeStride};
  const BufU ncnt{(char*)c.ncnt, c.nodeStride};
  uint32_t* cnt4 = c.cnt4;
  uint32_t* cntr = c.cntr;
  const int nrep = c.nrep, rep = tid & (nrep - 1);
  auto zero_counters = [&](int nslots) {  // the caller syncs
    for (int i = tid; i < nslots * nrep; i += OCT_NT) cntr[i] = 0;
  };
  auto reduce_counters = [&](int nslots) {  // cnt4[i] = sum of the replicas; the caller syncs before and after
    for (int i = tid; i < nslots; i += OCT_NT) {
      uint32_t v = 0;
      for (int r = 0; r < nrep; r++) v += cntr[i * nrep + r];
      cnt4[i] = v;
    }
  };
  uint16_t* cpos = c.cpos;
  uint64_t* scan = c.scan;
  uint16_t* mark = c.mark;
  uint32_t* nmid = c.nmid;
  uint64_t* tsum = c.tsum;
  // split point of node i of buffer b (DivideNode :494-495: halfX = ceil(width / 2)) + the phase-1 split flag
  auto pack_mid = [&](int b, int i) {
    const int x0 = nx0[b][i], x1 = nx1[b][i], y0 = ny0[b][i], y1 = ny1[b][i];
    return (uint32_t)(x0 + ((x1 - x0 + 1) >> 1)) | ((uint32_t)(y0 + ((y1 - y0 + 1) >> 1)) << 12) |
           (ncnt[b][i] > 1 ? 0x80000000u : 0u);
  };
  auto classify = [&](uint32_t key, uint32_t mid) {  // quadrant(), from the packed split point
    return (key_x(key) >= (int)(mid & 0xFFF) ? 1 : 0) | (key_y(key) >= (int)((mid >> 12) & 0xFFF) ? 2 : 0);
  };
  int* s_i = c.s_i;

  const int W = L.w - 2 * kBorder, H = L.h - 2 * kBorder;
  const int nIni = (int)roundf((float)W / (float)H);  // :566
  const float hX = (float)W / (float)nIni;             // :568

  // ---- roots (:575-601)
  zero_counters(kMaxIni);
  __syncthreads();
  cd.sweep([&](int, uint32_t key, uint32_t& nd) {
    const int r = (int)((float)key_x(key) / hX);
    nd = (uint32_t)r;
    atomicAdd(&cntr[r * nrep + rep], 1u);
  });
  __syncthreads();
  reduce_counters(kMaxIni);
  __syncthreads();
  if (tid == 0) {
    int na = 0;
    for (int i = 0; i < nIni; i++) {
      if (cnt4[i] == 0) {
        cpos[i] = 0xFFFF;
        continue;
      }
      nx0[0][na] = (int16_t)(int)(hX * (float)i);
      nx1[0][na] = (int16_t)(int)(hX * (float)(i + 1));
      ny0[0][na] = 0;
      ny1[0][na] = (int16_t)H;
      ncnt[0][na] = cnt4[i];
      cpos[i] = (uint16_t)na;
      na++;
    }
    s_i[0] = na;
    for (int i = 0; i < na; i++) nmid[i] = pack_mid(0, i);
  }
  __syncthreads();
  cd.sweep([&](int, uint32_t, uint32_t& nd) { nd = cpos[nd]; });
  int nA = s_i[0];
  int cur = 0;
  __syncthreads();

  MK();
  OCT_EXIT(2)
  bool finish = false;
  int nE = 0, ecur = 0;
  // ---- phase 1: split every expandable node per pass (:610-677)
  while (!finish) {
    const int prevSize = nA;
    zero_counters(nA * 4);
    __syncthreads();
    MK();
    cd.sweep([&](int, uint32_t key, uint32_t& nd) {
      const uint32_t mid = nmid[nd];
      if (mid >> 31) {
        const int q = classify(key, mid);
        atomicAdd(&cntr[(nd * 4 + q) * nrep + rep], 1u);
        nd |= (uint32_t)q << 14;
      }
    });
    __syncthreads();
    reduce_counters(nA * 4);
    __syncthreads();
    MK();
    for (int i = tid; i < nA; i += OCT_NT) {
      uint64_t cc = 0, nm = 1, ce = 0;
      if (ncnt[cur][i] > 1) {
        nm = 0;
        for (int q = 0; q < 4; q++) {
          cc += cnt4[i * 4 + q] > 0;
          ce += cnt4[i * 4 + q] > 1;
        }
      }
      scan[i] = cc | (nm << 21) | (ce << 42);
    }
    __syncthreads();
    const uint64_t tot = block_scan_u64(scan, nA, tsum);
    MK();
    const int tc = (int)(tot & 0x1FFFFF), tnm = (int)((tot >> 21) & 0x1FFFFF), tce = (int)(tot >> 42);
    const int nxt = cur ^ 1;
    for (int i = tid; i < nA; i += OCT_NT) {
      const uint64_t pre = scan[i];
      const int pc = (int)(pre & 0x1FFFFF), pnm = (int)((pre >> 21) & 0x1FFFFF), pce = (int)(pre >> 42);
      if (ncnt[cur][i] > 1) {
        const int x0 = nx0[cur][i], x1 = nx1[cur][i], y0 = ny0[cur][i], y1 = ny1[cur][i];
        const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
        int cc = 0;
        for (int q = 0; q < 4; q++) cc += cnt4[i * 4 + q] > 0;
        const int base = tc - (pc + cc);
        int after = 0, eb = pce;
        for (int q = 0; q < 4; q++) {  // E entries in creation order n1..n4
          const uint32_t cq = cnt4[i * 4 + q];
          if (cq > 1) {
            int rank_after = 0;
            for (int q2 = q + 1; q2 < 4; q2++) rank_after += cnt4[i * 4 + q2] > 0;
            const int cx0 = (q & 1) ? x0 + hx : x0;
            (c.ebuf + ecur * c.maxn)[eb++] = ((uint64_t)cq << 28) | ((uint64_t)(uint32_t)cx0 << 16) | (uint64_t)(base + rank_after);
          }
        }
        for (int q = 3; q >= 0; q--) {  // list order n4,n3,n2,n1
          const uint32_t cq = cnt4[i * 4 + q];
          if (cq == 0) continue;
          const int pos = base + after++;
          nx0[nxt][pos] = (int16_t)((q & 1) ? x0 + hx : x0);
          nx1[nxt][pos] = (int16_t)((q & 1) ? x1 : x0 + hx);
          ny0[nxt][pos] = (int16_t)((q & 2) ? y0 + hy : y0);
          ny1[nxt][pos] = (int16_t)((q & 2) ? y1 : y0 + hy);
          ncnt[nxt][pos] = cq;
          cpos[i * 4 + q] = (uint16_t)pos;
        }
      } else {
        const int pos = tc + pnm;
        nx0[nxt][pos] = nx0[cur][i];
        nx1[nxt][pos] = nx1[cur][i];
        ny0[nxt][pos] = ny0[cur][i];
        ny1[nxt][pos] = ny1[cur][i];
        ncnt[nxt][pos] = ncnt[cur][i];
        cpos[i * 4] = (uint16_t)pos;
      }
    }
    __syncthreads();
    MK();
    // (a node that is not split never had its quadrant tagged: tag 0 -> cpos[nd * 4])
    cd.sweep([&](int, uint32_t, uint32_t& v) { v = cpos[(v & 0x3FFF) * 4 + (v >> 14)]; });
    for (int i = tid; i < tc + tnm; i += OCT_NT) nmid[i] = pack_mid(nxt, i);
    __syncthreads();
    MK();
    cur = nxt;
    nA = tc + tnm;
    nE = tce;
    if (nA >= N || nA == prevSize) {
      finish = true;
    } else if (nA + 3 * nE > N) {
      break;  // -> phase 2
    }
  }

  MK();
  OCT_EXIT(3)
  // ---- phase 2: expand the largest nodes first until the quota is reached (:678-735)
  while (!finish) {
    const int prevSize = nA;
    uint64_t* E = c.ebuf + ecur * c.maxn;
    uint64_t* E2 = c.ebuf + (ecur ^ 1) * c.maxn;
    if (tid < 64 && ablate != 5)  // wave 0 sorts; scratch: scan (stopper lists), E2 (rank scatter), tsum (stack)
      introsort_wave(E, nE, E2, reinterpret_cast<uint16_t*>(scan), reinterpret_cast<uint16_t*>(scan) + c.maxn + 4,
                     reinterpret_cast<int*>(tsum), tid);
    MK();
    if (tid == 0) {
      s_i[1] = nE;  // cut (exclusive count of processed) defaults to all
      s_i[2] = 0;   // broke
    }
    zero_counters(nA * 4);
    for (int i = tid; i < nA; i += OCT_NT) {
      mark[i] = 0;
      nmid[i] &= 0x7FFFFFFFu;  // this pass splits exactly the nodes of the E list
    }
    __syncthreads();
    for (int m = tid; m < nE; m += OCT_NT) {
      const int nd = (int)(E[nE - 1 - m] & 0xFFFF);
      mark[nd] = (uint16_t)(m + 1);
      nmid[nd] |= 0x80000000u;
    }
    __syncthreads();
    cd.sweep([&](int, uint32_t key, uint32_t& nd) {
      const uint32_t mid = nmid[nd];
      if (mid >> 31) {
        const int q = classify(key, mid);
        atomicAdd(&cntr[(nd * 4 + q) * nrep + rep], 1u);
        nd |= (uint32_t)q << 14;
      }
    });
    __syncthreads();
    reduce_counters(nA * 4);
    __syncthreads();
    // scan over the processing order m: c (children), ce (expandable children)
    for (int m = tid; m < nE; m += OCT_NT) {
      const int nd = (int)(E[nE - 1 - m] & 0xFFFF);
      uint64_t cc = 0, ce = 0;
      for (int q = 0; q < 4; q++) {
        cc += cnt4[nd * 4 + q] > 0;
        ce += cnt4[nd * 4 + q] > 1;
      }
      scan[m] = cc | (ce << 21);
    }
    __syncthreads();
    block_scan_u64(scan, nE, tsum);
    // first m at which the list reaches N nodes: size after m+1 expansions = nA + C_incl(m) - (m+1)
    for (int m = tid; m < nE; m += OCT_NT) {
      const int nd = (int)(E[nE - 1 - m] & 0xFFFF);
      int cc = 0;
      for (int q = 0; q < 4; q++) cc += cnt4[nd * 4 + q] > 0;
      const int cincl = (int)(scan[m] & 0x1FFFFF) + cc;
      if (nA + cincl - (m + 1) >= N) {
        atomicMin(&s_i[1], m + 1);
        s_i[2] = 1;
      }
    }
    __syncthreads();
    const int nP = s_i[1];  // nodes m < nP are expanded
    const bool broke = s_i[2] != 0;
    int tc, tce;
    if (nP < nE) {
      tc = (int)(scan[nP] & 0x1FFFFF);
      tce = (int)(scan[nP] >> 21);
    } else {
      const int nd = (int)(E[0] & 0xFFFF);  // m = nE-1
      int cc = 0, ce = 0;
      for (int q = 0; q < 4; q++) {
        cc += cnt4[nd * 4 + q] > 0;
        ce += cnt4[nd * 4 + q] > 1;
      }
      tc = nE ? (int)(scan[nE - 1] & 0x1FFFFF) + cc : 0;
      tce = nE ? (int)(scan[nE - 1] >> 21) + ce : 0;
    }
    const int nxt = cur ^ 1;
    // children of processed nodes: later processed first, each group n4..n1
    for (int m = tid; m < nP; m += OCT_NT) {
      const int nd = (int)(E[nE - 1 - m] & 0xFFFF);
      const int x0 = nx0[cur][nd], x1 = nx1[cur][nd], y0 = ny0[cur][nd], y1 = ny1[cur][nd];
      const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
      int cc = 0;
      for (int q = 0; q < 4; q++) cc += cnt4[nd * 4 + q] > 0;
      const int pc = (int)(scan[m] & 0x1FFFFF);
      int eb = (int)(scan[m] >> 21);
      const int base = tc - (pc + cc);
      for (int q = 0; q < 4; q++) {
        const uint32_t cq = cnt4[nd * 4 + q];
        if (cq > 1) {
          int rank_after = 0;
          for (int q2 = q + 1; q2 < 4; q2++) rank_after += cnt4[nd * 4 + q2] > 0;
          const int cx0 = (q & 1) ? x0 + hx : x0;
          E2[eb++] = ((uint64_t)cq << 28) | ((uint64_t)(uint32_t)cx0 << 16) | (uint64_t)(base + rank_after);
        }
      }
      int after = 0;
      for (int q = 3; q >= 0; q--) {
        const uint32_t cq = cnt4[nd * 4 + q];
        if (cq == 0) continue;
        const int pos = base + after++;
        nx0[nxt][pos] = (int16_t)((q & 1) ? x0 + hx : x0);
        nx1[nxt][pos] = (int16_t)((q & 1) ? x1 : x0 + hx);
        ny0[nxt][pos] = (int16_t)((q & 2) ? y0 + hy : y0);
        ny1[nxt][pos] = (int16_t)((q & 2) ? y1 : y0 + hy);
        ncnt[nxt][pos] = cq;
        cpos[nd * 4 + q] = (uint16_t)pos;
      }
    }
    __syncthreads();
    // untouched nodes keep their relative order behind the new children
    for (int i = tid; i < nA; i += OCT_NT) scan[i] = (mark[i] == 0 || mark[i] > nP) ? 1 : 0;
    __syncthreads();
    const int nKeep = (int)block_scan_u64(scan, nA, tsum);
    for (int i = tid; i < nA; i += OCT_NT) {
      if (mark[i] == 0 || mark[i] > nP) {
        const int pos = tc + (int)scan[i];
        nx0[nxt][pos] = nx0[cur][i];
        nx1[nxt][pos] = nx1[cur][i];
        ny0[nxt][pos] = ny0[cur][i];
        ny1[nxt][pos] = ny1[cur][i];
        ncnt[nxt][pos] = ncnt[cur][i];
        // counted (tagged) but not expanded nodes map every quadrant tag back to the node itself
        cpos[i * 4] = cpos[i * 4 + 1] = cpos[i * 4 + 2] = cpos[i * 4 + 3] = (uint16_t)pos;
      }
    }
    __syncthreads();
    cd.sweep([&](int, uint32_t, uint32_t& v) { v = cpos[(v & 0x3FFF) * 4 + (v >> 14)]; });
    for (int i = tid; i < tc + nKeep; i += OCT_NT) nmid[i] = pack_mid(nxt, i);
    __syncthreads();
    cur = nxt;
    nA = tc + nKeep;
    nE = tce;
    ecur ^= 1;
    MK();
    if (broke || nA == prevSize) finish = true;
  }

  MK();
  OCT_EXIT(4)
  // ---- best response per node, first candidate (reference order) wins ties (:741-754): the canonical rank
  // (cell row, cell column, y, x) is unique per candidate, so exactly one candidate equals its node's maximum
  // and writes the node's output slot itself.
  // (replicated like the counters: all candidates of a node hammer one 64-bit LDS word otherwise)
  uint64_t* best = scan;
  uint64_t* bestr = reinterpret_cast<uint64_t*>(cntr);  // [node][nrepB]; cntr holds maxn * 4 * nrep words
  const int nrepB = nrep >= 2 ? nrep * 2 : 1, repB = tid & (nrepB - 1);
  if (nrepB > 1)
    for (int i = tid; i < nA * nrepB; i += OCT_NT) bestr[i] = 0;
  else
    for (int i = tid; i < nA; i += OCT_NT) best[i] = 0;
  __syncthreads();
  const float invH = 1.0f / (float)L.hCell, invW = 1.0f / (float)L.wCell;
  auto rank_key = [&](uint32_t key) {
    const int xr = key_x(key) - 3, yr = key_y(key) - 3;  // relative to the first detectable pixel (19,19)
    // floor(v / cell) for 0 <= v < 4096: (v + 0.5) / cell is >= 0.5 / cell away from every integer, far above float error
    const int cy = (int)(((float)yr + 0.5f) * invH), cx = (int)(((float)xr + 0.5f) * invW);
    const uint32_t rank = (uint32_t)(((cy * L.nCols + cx) * L.hCell + (yr - cy * L.hCell)) * L.wCell + (xr - cx * L.wCell));
    return ((unsigned long long)key_r(key) << 32) | (0xFFFFFFFFu - rank);
  };
  if (nrepB > 1) {
    cd.sweep([&](int, uint32_t key, uint32_t& nd) {
      atomicMax((unsigned long long*)&bestr[nd * nrepB + repB], rank_key(key));
    });
    __syncthreads();
    for (int i = tid; i < nA; i += OCT_NT) {
      uint64_t v = 0;
      for (int r = 0; r < nrepB; r++) v = bestr[i * nrepB + r] > v ? bestr[i * nrepB + r] : v;
      best[i] = v;
    }
  } else {
    cd.sweep([&](int, uint32_t key, uint32_t& nd) { atomicMax((unsigned long long*)&best[nd], rank_key(key)); });
  }
  __syncthreads();
  const int nOut = min(nA, L.selCap);
  cd.sweep([&](int, uint32_t key, uint32_t& nd) {
    if ((int)nd < nOut && best[nd] == rank_key(key)) out[nd] = pack_key(key_x(key) + kBorder, key_y(key) + kBorder, key_r(key));
  });
  if (tid == 0) *outCount = nOut;
  MK();
#ifdef OCT_PROF
  if (tid == 0 && blockIdx.y == 0 && (int)blockIdx.x == ablate - 100) {
    printf("L%d n=%d nA=%d :", (int)blockIdx.x, n, nA);
    for (int i = 1; i < nmk; i++) printf(" %d", (int)(tmk[i] - tmk[i - 1]));
    printf("\n");
  }
#endif
#undef MK
#undef OCT_EXIT
}

__global__ __launch_bounds__(OCT_NT, 4) void k_octree(Geom g, const uint32_t* __restrict__ cellCand,
                                                const int* __restrict__ cellCount, int* __restrict__ cellPrefix,
                                                uint32_t* __restrict__ cand, int* __restrict__ candCount,
                                                uint16_t* __restrict__ knode, uint32_t* __restrict__ sel,
                                                int* __restrict__ selCount, int ablate, int forceGlobal) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int s_i[8];
  const int tid = threadIdx.x;
  const int l = blockIdx.x, img = blockIdx.y;
  const LevelDev L = g.lv[l];
  const int maxn = oct_maxn(g);
  const int nrep = oct_rep(g);
  const OctLds o = oct_layout(maxn, oct_maxcells(g), nrep);
  OctCtx c;
  c.nx0 = (int16_t*)(smem + o.nx0[0]);
  c.nx1 = (int16_t*)(smem + o.nx1[0]);
  c.ny0 = (int16_t*)(smem + o.ny0[0]);
  c.ny1 = (int16_t*)(smem + o.ny1[0]);
  c.ncnt = (uint32_t*)(smem + o.ncnt[0]);
  c.nodeStride = o.nx0[1] - o.nx0[0];  // the five tables of one buffer are laid out back to back
  c.ebuf = (uint64_t*)(smem + o.e[0]);  // e[1] follows e[0] (maxn entries each)
  c.cnt4 = (uint32_t*)(smem + o.cnt4);
  c.cntr = (uint32_t*)(smem + o.cntr);
  c.nrep = nrep;
  c.cpos = (uint16_t*)(smem + o.cpos);
  c.scan = (uint64_t*)(smem + o.scan);
  c.mark = (uint16_t*)(smem + o.mark);
  c.nmid = (uint32_t*)(smem + o.bestk);
  c.tsum = (uint64_t*)(smem + o.tsum);
  c.cellpre = (int*)(smem + o.cellpre);
  c.s_i = s_i;
  c.maxn = maxn;

  // ---- exclusive scan of the level's per-cell counts (the sparse per-cell slots are compacted by octree_body;
  // candidate order is irrelevant: ties are broken by the canonical rank)
  const int cells = L.nCols * L.nRows;
  int n;
  {
    const int* cc = cellCount + (long long)img * g.totalCells + L.cellStart;
    const int per = (cells + OCT_NT - 1) / OCT_NT;
    const int cb = min(tid * per, cells), ce = min(cb + per, cells);
    int sum = 0;
    for (int ci = cb; ci < ce; ci++) sum += cc[ci];
    int incl = sum;
    {
      const int lane = tid & 63;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      if (lane == 63) c.tsum[tid >> 6] = (uint64_t)incl;
    }
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < OCT_NT / 64; w++) {
      const int t = (int)c.tsum[w];
      if (w < (tid >> 6)) wbase += t;
      total += t;
    }
    n = min(total, L.candCap);
    int run = wbase + incl - sum;
    for (int ci = cb; ci < ce; ci++) {
      c.cellpre[ci] = run;
      run += cc[ci];
    }
    if (tid == 0) {
      c.cellpre[cells] = total;
      candCount[img * g.nlevels + l] = n;
    }
    __syncthreads();
  }
  const uint32_t* sparse = cellCand + (long long)img * g.cellImg + L.cellOff;
  uint32_t* keys = cand + (long long)img * g.candImg + L.candOff;
  uint16_t* kn = knode + (long long)img * g.candImg + L.candOff;
  uint32_t* out = sel + (long long)img * g.selImg + L.selOff;
  int* outCount = selCount + img * g.nlevels + l;
  if (n <= OCT_KMAX * OCT_NT && !forceGlobal)
    octree_body<true>(g, L, c, n, cells, sparse, keys, kn, out, outCount, ablate);
  else
    octree_body<false>(g, L, c, n, cells, sparse, keys, kn, out, outCount, ablate);
}

static int g_octree_force_global_host = 0;
void debug_set_octree_global(int on) { g_octree_force_global_host = on ? 1 : 0; }

hipError_t launch_octree(const Geom& g, int nimg, const uint32_t* cellCand, const int* cellCount, int* cellPrefix,
                         uint32_t* cand, int* candCount, uint16_t* knode, uint32_t* sel, int* selCount,
                         hipStream_t s) {
  dim3 grid(g.nlevels, nimg);
  static const int ablate = getenv("ORBX_OCTREE_ABLATE") ? atoi(getenv("ORBX_OCTREE_ABLATE")) : 0;
  hipLaunchKernelGGL(k_octree, grid, dim3(OCT_NT), octree_lds_bytes(g), s, g, cellCand, cellCount, cellPrefix, cand,
                     candCount, knode, sel, selCount, ablate, g_octree_force_global_host);
  return hipGetLastError();
}

// ================================================================================================ blur
// 7x7 Gaussian, fixed point 8.8 taps {18,34,48,56,48,34,18} (OpenCV >= 4.5.1), BORDER_REFLECT_101 at the
// level's own edges, exact integer accumulation with one rounding.  64x32 output tile / block.
__device__ __forceinline__ int reflect101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return p;
}

#define BL_TW 128
#define BL_TH 32
// Tile of 128x32 outputs per 256-thread block.  In-tile: rows y0-3 .. y0+34, columns x0-4 .. x0+131 as aligned
// dwords.  Horizontal pass: a thread turns 3 dwords into 4 outputs with 6 v_alignbyte + 8 v_dot4_u32_u8 (taps
// 18,34,48,56 | 48,34,18,0) for TWO vertically adjacent rows and stores them as u16 pairs (row r | row r+1 << 16;
// max 255*256 fits).  Vertical pass: a thread owns a 4x4 output block; with rows packed in pairs the 7-tap
// column filter is 3 v_dot2_u32_u16 + 1 mad per output.  All integer, exact; one rounding (+32768 >> 16).
__global__ __launch_bounds__(256) void k_blur(Geom g, Pyr p, int level0, int level1, int xcdRun) {
  __shared__ uint32_t in[BL_TH + 6][BL_TW / 4 + 2 + 1];    // +1: pad against bank conflicts
  __shared__ uint32_t hp[(BL_TH + 6) / 2][BL_TW + 1];      // [row pair][32*(x%4) + x/4] = H(2j, x) | H(2j+1, x) << 16
                                                           // (quad-transposed columns: both passes bank-conflict free)
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  int tile = xcd_run_remap(blockIdx.x, gridDim.x, blockIdx.z, xcdRun);
  int l = level0;
  for (;; l++) {  // tiles of levels [level0, level1) are enumerated in one grid dimension
    const int tx = (g.lv[l].w + BL_TW - 1) / BL_TW, ty = (g.lv[l].h + BL_TH - 1) / BL_TH;
    if (tile < tx * ty || l + 1 == level1) break;
    tile -= tx * ty;
  }
  const LevelDev L = g.lv[l];
  const int tx = (L.w + BL_TW - 1) / BL_TW;
  if (tile >= tx * ((L.h + BL_TH - 1) / BL_TH)) return;
  const int x0 = (tile % tx) * BL_TW, y0 = (tile / tx) * BL_TH;
  int pitch;
  const uint8_t* im = level_ptr(g, p, img, l, pitch);
  constexpr int IW = BL_TW / 4 + 2;  // in-tile dwords per row
  if (x0 >= 4 && y0 >= 3 && x0 + BL_TW + 4 <= L.w && y0 + BL_TH + 3 <= L.h) {
    // interior tile: every dword is inside the image; thread = (row phase, dword column), 7 rows per pass
    const int c = tid % IW, r0 = tid / IW;  // IW = 34: 238 of the 256 threads take part
    if (r0 < 7) {
      const uint8_t* src = im + (long long)(y0 - 3 + r0) * pitch + (x0 - 4) + 4 * c;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int r = r0 + 7 * k;
        if (r < BL_TH + 6) in[r][c] = *reinterpret_cast<const uint32_t*>(src + (long long)(7 * k) * pitch);
      }
    }
  } else {
    for (int i = tid; i < (BL_TH + 6) * IW; i += 256) {
      const int r = i / IW, c = i - r * IW;
      const int y = y0 + r - 3, x = x0 - 4 + 4 * c;
      uint32_t v;
      if (y >= 0 && y < L.h && x >= 0 && x + 4 <= L.w) {
        v = *reinterpret_cast<const uint32_t*>(im + (long long)y * pitch + x);
      } else {  // image border: BORDER_REFLECT_101 (coordinates further out only feed discarded outputs)
        const int yy = reflect101(min(max(y, -3), L.h + 2), L.h);
        v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int xx = reflect101(min(max(x + k, -3), L.w + 2), L.w);
          v |= (uint32_t)im[(long long)yy * pitch + xx] << (8 * k);
        }
      }
      in[r][c] = v;
    }
  }
  __syncthreads();
  // horizontal pass: item = (row pair j, dword column c): 19 x 32 items
  for (int i = tid; i < ((BL_TH + 6) / 2) * (BL_TW / 4); i += 256) {
    const int j = i / (BL_TW / 4), c = i - j * (BL_TW / 4);
    uint32_t h[2][4];
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const uint32_t Lw = in[2 * j + rr][c], C = in[2 * j + rr][c + 1], R = in[2 * j + rr][c + 2];
      const uint32_t wA = 0x38302212u, wB = 0x00122230u;  // taps -3..0 and +1..+3 (LSB = lowest x)
      h[rr][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, Lw, 1), wA,
                                        __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(R, C, 1), wB, 0, false), false);
      h[rr][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, Lw, 2), wA,
                                        __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(R, C, 2), wB, 0, false), false);
      h[rr][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, Lw, 3), wA,
                                        __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(R, C, 3), wB, 0, false), false);
      h[rr][3] = __builtin_amdgcn_udot4(C, wA, __builtin_amdgcn_udot4(R, wB, 0, false), false);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) hp[j][32 * k + c] = h[0][k] | (h[1][k] << 16);  // column 4c+k -> slot 32k+c
  }
  __syncthreads();
  {
    const int bc = tid & 31, br = tid >> 5;  // 32 x 8 blocks of 4x4 outputs; block rows 4*br .. 4*br+3 (even start)
    uint8_t* dst = p.blur + (long long)img * g.pyrImg + L.off;
    uint32_t outw[4];
    uint32_t accs[4][4];
#pragma unroll
    for (int cI = 0; cI < 4; cI++) {
      uint32_t pr[5];  // row pairs (4br .. 4br+9) of column 4*bc + cI
#pragma unroll
      for (int k = 0; k < 5; k++) pr[k] = hp[2 * br + k][32 * cI + bc];  // lanes read consecutive dwords
      // taps {18,34,48,56,48,34,18} on rows r..r+6; the lone 7th tap is a dot2 with a zero partner and the rounding
      // constant rides in as the first accumulator, so an output is 4 v_dot2_u32_u16
      const uint32_t w01 = 18u | (34u << 16), w23 = 48u | (56u << 16), w45 = 48u | (34u << 16), w6 = 18u;  // even r
      const uint32_t v0 = 18u << 16, v12 = 34u | (48u << 16), v34 = 56u | (48u << 16), v56 = 34u | (18u << 16);  // odd r
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int k0 = rr >> 1;
        uint32_t acc;
        if ((rr & 1) == 0) {  // rows r = 4br + rr (even): pairs k0 .. k0+2, then the low half of pair k0+3
          acc = udot2_u16(pr[k0], w01, 32768u);
          acc = udot2_u16(pr[k0 + 1], w23, acc);
          acc = udot2_u16(pr[k0 + 2], w45, acc);
          acc = udot2_u16(pr[k0 + 3], w6, acc);
        } else {              // odd r: high half of pair k0, then pairs k0+1 .. k0+3
          acc = udot2_u16(pr[k0], v0, 32768u);
          acc = udot2_u16(pr[k0 + 1], v12, acc);
          acc = udot2_u16(pr[k0 + 2], v34, acc);
          acc = udot2_u16(pr[k0 + 3], v56, acc);
        }
        accs[rr][cI] = acc;  // result byte = bits 16..23
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {  // gather byte 2 of the four accumulators with v_perm_b32
      const uint32_t lo = __builtin_amdgcn_perm(accs[rr][1], accs[rr][0], 0x0c0c0602u);  // [acc0.b2, acc1.b2, 0, 0]
      const uint32_t hi = __builtin_amdgcn_perm(accs[rr][3], accs[rr][2], 0x06020c0cu);  // [0, 0, acc2.b2, acc3.b2]
      outw[rr] = lo | hi;
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int y = y0 + 4 * br + rr, x = x0 + 4 * bc;
      if (y < L.h && x < L.w) *reinterpret_cast<uint32_t*>(dst + (long long)y * L.pitch + x) = outw[rr];
    }
  }
}

hipError_t launch_blur(const Geom& g, const Pyr& p, int nimg, int level0, int level1, hipStream_t s) {
  int tiles = 0;
  for (int l = level0; l < level1; l++)
    tiles += ((g.lv[l].w + BL_TW - 1) / BL_TW) * ((g.lv[l].h + BL_TH - 1) / BL_TH);
  if (tiles == 0) return hipSuccess;
  static const int xcdRun = getenv("ORBX_BLUR_XCD_RUN") ? atoi(getenv("ORBX_BLUR_XCD_RUN")) : 1;
  hipLaunchKernelGGL(k_blur, dim3(tiles, 1, nimg), dim3(256), 0, s, g, p, level0, level1, xcdRun);
  return hipGetLastError();
}

// ================================================================================================ slots
// Output slot of every selected keypoint under the serial semantics of :1062-1099: walk levels then list
// order; keypoints inside the lapping area fill from the back, the others from the front.
__global__ __launch_bounds__(256) void k_slots(Geom g, const uint32_t* __restrict__ sel,
                                               const int* __restrict__ selCount, const int* __restrict__ lap,
                                               int* __restrict__ slot, int* __restrict__ nOut,
                                               int* __restrict__ mono) {
  __shared__ int cum[ORBX_MAX_LEVELS + 1];
  __shared__ int tsum[256];
  const int tid = threadIdx.x, img = blockIdx.x;
  if (tid == 0) {
    int c = 0;
    for (int l = 0; l < g.nlevels; l++) {
      cum[l] = c;
      c += selCount[img * g.nlevels + l];
    }
    cum[g.nlevels] = c;
  }
  __syncthreads();
  const int n = cum[g.nlevels];
  const float lap0 = (float)lap[2 * img], lap1 = (float)lap[2 * img + 1];
  const int per = (n + 255) >> 8;
  const int b = min(tid * per, n), e = min(b + per, n);
  auto in_lap = [&](int gidx, int& lvl, int& idx) {
    lvl = 0;
    while (gidx >= cum[lvl + 1]) lvl++;
    idx = gidx - cum[lvl];
    const uint32_t key = sel[(long long)img * g.selImg + g.lv[lvl].selOff + idx];
    float x = (float)key_x(key);
    if (lvl != 0) x = x * g.lv[lvl].scale;
    return x >= lap0 && x <= lap1;
  };
  int cnt = 0;
  for (int i = b; i < e; i++) {
    int lv, ix;
    cnt += in_lap(i, lv, ix) ? 1 : 0;
  }
  tsum[tid] = cnt;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int t = tid >= d ? tsum[tid - d] : 0;
    __syncthreads();
    tsum[tid] += t;
    __syncthreads();
  }
  int lapBefore = tid ? tsum[tid - 1] : 0;
  for (int i = b; i < e; i++) {
    int lv, ix;
    const bool il = in_lap(i, lv, ix);
    const int s = il ? (n - 1 - lapBefore) : (i - lapBefore);
    slot[(long long)img * g.selImg + g.lv[lv].selOff + ix] = s;
    lapBefore += il ? 1 : 0;
  }
  if (tid == 0) {
    nOut[img] = n;
    mono[img] = n - tsum[255];
  }
}

hipError_t launch_slots(const Geom& g, int nimg, const uint32_t* sel, const int* selCount, const int* lap,
                        int* slot, int* nOut, int* mono, hipStream_t s) {
  hipLaunchKernelGGL(k_slots, dim3(nimg), dim3(256), 0, s, g, sel, selCount, lap, slot, nOut, mono);
  return hipGetLastError();
}

// ================================================================================================ describe
// cv::fastAtan2: every product / sum rounded separately.
__device__ __forceinline__ float fast_atan2_dev(float y, float x) {
  const float s = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s,
              p7 = -0.04432655554792128f * s;
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// sin/cos of the descriptor rotation: IEEE double, fixed operation sequence (identical to the oracle's
// definition, oracle/orb_oracle.cpp orb_sincosf), rounded once to float.
__device__ __forceinline__ void orb_sincos_dev(float ang, float& s_out, float& c_out) {
  const double x = (double)ang;
  const double fk = floor(__dadd_rn(__dmul_rn(x, 6.36619772367581382433e-01), 0.5));
  const int k = (int)fk;
  const double r = __dsub_rn(__dsub_rn(x, __dmul_rn(fk, 1.57079632673412561417e+00)),
                             __dmul_rn(fk, 6.07710050650619224932e-11));
  const double z = __dmul_rn(r, r);
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
#define DM(a, b) __dmul_rn(a, b)
#define DA(a, b) __dadd_rn(a, b)
  const double ps = DA(S2, DM(z, DA(S3, DM(z, DA(S4, DM(z, DA(S5, DM(z, S6))))))));
  const double sn = DA(r, DM(DM(z, r), DA(S1, DM(z, ps))));
  const double pc = DM(z, DA(C1, DM(z, DA(C2, DM(z, DA(C3, DM(z, DA(C4, DM(z, DA(C5, DM(z, C6)))))))))));
  const double cs = __dsub_rn(1.0, __dsub_rn(DM(0.5, z), DM(z, pc)));
#undef DM
#undef DA
  double sv, cv;
  switch (k & 3) {
    case 0: sv = sn; cv = cs; break;
    case 1: sv = cs; cv = -sn; break;
    case 2: sv = -sn; cv = -cs; break;
    default: sv = -cs; cv = sn; break;
  }
  s_out = (float)sv;
  c_out = (float)cv;
}

// One wave per selected keypoint.
//  * IC_Angle (:75-99): lanes run along patch COLUMNS so that every load instruction reads one 31-byte row
//    segment (1-2 cache lines) — two half-waves take the upper / lower 15 rows; integer moments are reduced
//    with DPP shuffles.
//  * rBRIEF (:102-147): the 37x37 footprint of the blurred level is staged in LDS with aligned dword loads,
//    then lane t evaluates tests t, t+64, t+128, t+192; each __ballot is 8 descriptor bytes already in the
//    reference's byte/bit order.
//  * the keypoint + descriptor are written to their serial-order output slot.
#define DS_PITCH 48
__global__ __launch_bounds__(256) void k_describe(Geom g, Pyr p, const uint32_t* __restrict__ sel,
                                                  const int* __restrict__ selCount, const int* __restrict__ slot,
                                                  orbx_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                  int* __restrict__ nOut, int* __restrict__ mono, int ablate) {
  __shared__ uint32_t patch_all[4][37 * DS_PITCH / 4];
  if (ablate == 1) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int s = blockIdx.x * 4 + wv;
  const int img = blockIdx.y;
  if (s >= g.selImg) return;
  int l = 0;
  while (l + 1 < g.nlevels && s >= g.lv[l + 1].selOff) l++;
  const LevelDev L = g.lv[l];
  const int idx = s - L.selOff;
  if (!slot && s == 0 && lane == 0) {  // k_slots skipped: this wave publishes the image's counts
    int total = 0;
    for (int q = 0; q < g.nlevels; q++) total += selCount[img * g.nlevels + q];
    nOut[img] = total;
    mono[img] = total;
  }
  if (idx >= selCount[img * g.nlevels + l]) return;
  const uint32_t key = sel[(long long)img * g.selImg + s];
  const int X = key_x(key), Y = key_y(key);
  if (ablate == 2) return;
  uint32_t* patch = patch_all[wv];
  const uint8_t* patch8 = reinterpret_cast<const uint8_t*>(patch);
  // IC_Angle on the unblurred level (:75-99): the 31 rows x 9 aligned dwords covering the circular patch are
  // loaded as 279 coalesced dword items (5 per lane, all issued up front); every lane folds its bytes into the
  // integer moments (u * I, v * I) under the circle mask |u| <= umax[|v|].
  int pitch;
  const uint8_t* im = level_ptr(g, p, img, l, pitch);
  int m10 = 0, m01 = 0;
    const int xr = (X - 15) & ~3, misr = (X - 15) - xr;
    uint32_t w[5];  // (loads issued before the footprint staging below, so 12 loads are in flight together)
    int um[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const int i = min(lane + 64 * t, 278);
      const int r = i / 9, c = i - r * 9;
      w[t] = *reinterpret_cast<const uint32_t*>(im + (long long)(Y - 15 + r) * pitch + xr + 4 * c);
      um[t] = c_umax[r < 15 ? 15 - r : r - 15];
    }
  // stage the blurred 37x37 footprint: rows Y-18..Y+18, aligned dwords covering columns X-18..X+18
  const uint8_t* bl = p.blur + (long long)img * g.pyrImg + L.off;
  const int xs = (X - 18) & ~3, mis = (X - 18) - xs;  // level pitch is a multiple of 64 -> rows are dword aligned
  for (int i = lane; i < 37 * 11; i += 64) {
    const int r = i / 11, c = i - r * 11;
    patch[r * (DS_PITCH / 4) + c] =
        *reinterpret_cast<const uint32_t*>(bl + (long long)(Y - 18 + r) * L.pitch + xs + 4 * c);
  }
  if (ablate == 3) return;
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const int i = lane + 64 * t;
      const int ii = min(i, 278);
      const int r = ii / 9, c = ii - r * 9;
      const int v = r - 15;
      const int lim = i < 279 ? um[t] : -1;
      int rs = 0;
#pragma unroll
      for (int bI = 0; bI < 4; bI++) {
        const int u = 4 * c + bI - misr - 15;
        const int au = u < 0 ? -u : u;
        const int val = au <= lim ? (int)((w[t] >> (8 * bI)) & 0xFF) : 0;
        rs += val;
        m10 += u * val;
      }
      m01 += v * rs;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m10 += __shfl_xor(m10, o);
    m01 += __shfl_xor(m01, o);
  }
  if (ablate == 4) { if (m10 == 12345 && lane == 0) kps[0].x = 1; return; }
  const float angle = fast_atan2_dev((float)m01, (float)m10);
  const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
  float a, b;
  orb_sincos_dev(__fmul_rn(angle, factorPI), b, a);  // a = cos, b = sin
  // slot == nullptr: no keypoint can lie in the lapping area (lap1 < 19 <= every x), so the serial-order slot is
  // simply (keypoints of the earlier levels) + idx and k_slots is not launched at all
  int n_out_slot;
  if (slot) {
    n_out_slot = slot[(long long)img * g.selImg + s];
  } else {
    int before = 0;
    for (int q = 0; q < l; q++) before += selCount[img * g.nlevels + q];
    n_out_slot = before + idx;
  }
  uint8_t* dout = desc + ((long long)img * g.outCap + n_out_slot) * 32;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // patch writes of this wave before its own reads
  __builtin_amdgcn_wave_barrier();
  const uint8_t* centre = patch8 + 18 * DS_PITCH + 18 + mis;
#pragma unroll
  for (int gI = 0; gI < 4; gI++) {
    const int8_t* pt = c_pattern + 4 * (64 * gI + lane);
    const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
    const int iy0 = rne_f(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
    const int ix0 = rne_f(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
    const int iy1 = rne_f(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
    const int ix1 = rne_f(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
    const int t0 = centre[iy0 * DS_PITCH + ix0], t1 = centre[iy1 * DS_PITCH + ix1];
    const uint64_t bits = __ballot(t0 < t1);
    if (lane == 0) *reinterpret_cast<uint64_t*>(dout + 8 * gI) = bits;
  }
  if (lane == 0) {
    orbx_keypoint kp;
    kp.x = l ? __fmul_rn((float)X, L.scale) : (float)X;
    kp.y = l ? __fmul_rn((float)Y, L.scale) : (float)Y;
    kp.size = L.patch;
    kp.angle = angle;
    kp.response = (float)key_r(key);
    kp.octave = l;
    kp.class_id = -1;
    kps[(long long)img * g.outCap + n_out_slot] = kp;
  }
}

hipError_t launch_describe(const Geom& g, const Pyr& p, int nimg, const uint32_t* sel, const int* selCount,
                           const int* slot, orbx_keypoint* kps, uint8_t* desc, int* nOut, int* mono, hipStream_t s) {
  static const int ablate = getenv("ORBX_DESC_ABLATE") ? atoi(getenv("ORBX_DESC_ABLATE")) : 0;
  hipLaunchKernelGGL(k_describe, dim3((g.selImg + 3) / 4, nimg), dim3(256), 0, s, g, p, sel, selCount, slot,
                     kps, desc, nOut, mono, ablate);
  return hipGetLastError();
}

// ================================================================================================ stereo
__device__ __forceinline__ int hamming256(const uint32_t* a, const uint32_t* b) {
  int d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d += __popc(a[i] ^ b[i]);
  return d;
}

// Right keypoints bucketed by integer row (counting sort, one block per pair): the analogue of the
// reference's vRowIndices table (src/Frame.cc:930-949), but one entry per keypoint; the +-2*scale band is
// applied by the matcher, which only has to visit rows [vL - band, vL + band].
__global__ __launch_bounds__(256) void k_stereo_rows(StereoArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int* hist = reinterpret_cast<int*>(smem);  // imgH + 1 counters, then running offsets
  __shared__ int tsum[256];
  const int tid = threadIdx.x, pair = blockIdx.x;
  const int imgR = a.firstR + pair;
  const int nR = a.nR[imgR];
  const orbx_keypoint* kR = a.kR + (long long)imgR * a.capR;
  int* rowStart = a.rowStart + (long long)pair * (a.imgH + 1);
  int* items = a.rowItems + (long long)pair * a.capR;
  for (int r = tid; r <= a.imgH; r += 256) hist[r] = 0;
  __syncthreads();
  for (int i = tid; i < nR; i += 256) atomicAdd(&hist[min(max((int)kR[i].y, 0), a.imgH - 1)], 1);
  __syncthreads();
  const int per = (a.imgH + 256) >> 8;
  const int b = min(tid * per, a.imgH + 1), e = min(b + per, a.imgH + 1);
  int sum = 0;
  for (int r = b; r < e; r++) sum += hist[r];
  tsum[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int t = tid >= d ? tsum[tid - d] : 0;
    __syncthreads();
    tsum[tid] += t;
    __syncthreads();
  }
  int run = tid ? tsum[tid - 1] : 0;
  for (int r = b; r < e; r++) {
    const int c = hist[r];
    hist[r] = run;
    rowStart[r] = run;
    run += c;
  }
  __syncthreads();
  for (int i = tid; i < nR; i += 256) items[atomicAdd(&hist[min(max((int)kR[i].y, 0), a.imgH - 1)], 1)] = i;
}

hipError_t launch_stereo_rows(const StereoArgs& a, int npairs, hipStream_t s) {
  hipLaunchKernelGGL(k_stereo_rows, dim3(npairs), dim3(256), (size_t)(a.imgH + 2) * 4, s, a);
  return hipGetLastError();
}

// One wave per left keypoint.  The reference scans vRowIndices[vL] (right keypoints whose +-2*scale row band
// covers row vL, ascending iR) and keeps the first strict minimum; that is the minimum of (dist, iR) over
// all right keypoints passing the same band/octave/disparity filters, which is what the lanes compute.
__global__ __launch_bounds__(256) void k_stereo_match(Geom g, Pyr pl, Pyr pr, StereoArgs a) {
  const int lane = threadIdx.x & 63;
  const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int pair = blockIdx.y;
  const int imgL = a.firstL + pair, imgR = a.firstR + pair;
  const int nL = a.nL[imgL];
  if (iL >= nL) return;
  const long long oL = (long long)pair * a.capL + iL;
  const orbx_keypoint kpL = a.kL[(long long)imgL * a.capL + iL];
  const orbx_keypoint* kR = a.kR + (long long)imgR * a.capR;
  const uint32_t* dR = reinterpret_cast<const uint32_t*>(a.dR + (long long)imgR * a.capR * 32);
  uint32_t dl[8];
  {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(a.dL + ((long long)imgL * a.capL + iL) * 32);
#pragma unroll
    for (int i = 0; i < 8; i++) dl[i] = q[i];
  }
  float uR_out = -1.f, depth_out = -1.f;
  int sad_out = -1;
  const float uL = kpL.x, vL = kpL.y;
  const int levelL = kpL.octave;
  const float maxD = __fdiv_rn(a.bf, a.b);
  const float minU = __fsub_rn(uL, maxD), maxU = uL;
  const int row = (int)vL;
  uint32_t best = (100u << 16);  // TH_HIGH, strict '<'
  if (!(maxU < 0)) {
    const int* rowStart = a.rowStart + (long long)pair * (a.imgH + 1);
    const int* items = a.rowItems + (long long)pair * a.capR;
    const int jb = rowStart[min(max(row - a.band, 0), a.imgH)], je = rowStart[min(max(row + a.band + 1, 0), a.imgH)];
    for (int base = jb; base < je; base += 64) {
      const int j = base + lane;
      if (j < je) {
        const int iR = items[j];
        const orbx_keypoint k = kR[iR];
        const float r = __fmul_rn(2.0f, g.lv[k.octave].scale);
        const int maxr = (int)ceilf(__fadd_rn(k.y, r)), minr = (int)floorf(__fsub_rn(k.y, r));
        const bool ok = !(k.y == 0.0f && k.x == 0.0f) && row >= minr && row <= maxr &&
                        k.octave >= levelL - 1 && k.octave <= levelL + 1 && k.x >= minU && k.x <= maxU;
        if (ok) {
          const uint32_t cand = ((uint32_t)hamming256(dl, dR + (long long)iR * 8) << 16) | (uint32_t)iR;
          best = min(best, cand);
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
  const int bestDist = (int)(best >> 16);
  if (bestDist < 75) {  // thOrbDist = (TH_HIGH + TH_LOW) / 2
    const int bestIdxR = (int)(best & 0xFFFF);
    const float uR0 = kR[bestIdxR].x;
    const float sf = 1.0f / g.lv[levelL].scale;  // mvInvScaleFactors
    const float su = roundf(__fmul_rn(kpL.x, sf)), sv = roundf(__fmul_rn(kpL.y, sf)), sr = roundf(__fmul_rn(uR0, sf));
    const LevelDev L = g.lv[levelL];
    const float endu = sr + 11.0f;
    if (!(sr < 0 || endu >= (float)L.w)) {
      int pitchL, pitchR;
      const uint8_t* imL = level_ptr(g, pl, imgL, levelL, pitchL);
      const uint8_t* imR = level_ptr(g, pr, imgR, levelL, pitchR);
      const int yl = (int)sv - 5, xl = (int)su - 5, xr0 = (int)sr - 5;
      // 11x11 SAD for the 11 shifts; lanes cover the 121 pixels
      int sadv[11];
      int l0 = 0, l1 = 0;
      const int p0 = lane, p1 = lane + 64;
      const int y0 = p0 / 11, x0 = p0 - y0 * 11, y1 = p1 / 11, x1 = p1 - y1 * 11;
      l0 = imL[(long long)(yl + y0) * pitchL + xl + x0];
      if (p1 < 121) l1 = imL[(long long)(yl + y1) * pitchL + xl + x1];
#pragma unroll
      for (int inc = 0; inc < 11; inc++) {
        int s = abs(l0 - (int)imR[(long long)(yl + y0) * pitchR + xr0 + (inc - 5) + x0]);
        if (p1 < 121) s += abs(l1 - (int)imR[(long long)(yl + y1) * pitchR + xr0 + (inc - 5) + x1]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        sadv[inc] = s;
      }
      int bestSad = 0x7FFFFFFF, bestinc = 0;
#pragma unroll
      for (int inc = 0; inc < 11; inc++)
        if (sadv[inc] < bestSad) {
          bestSad = sadv[inc];
          bestinc = inc - 5;
        }
      if (bestinc != -5 && bestinc != 5) {
        float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int inc = 1; inc < 10; inc++)
          if (inc - 5 == bestinc) {
            d1 = (float)sadv[inc - 1];
            d2 = (float)sadv[inc];
            d3 = (float)sadv[inc + 1];
          }
        const float den = __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2)));
        const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), den);
        if (!(deltaR < -1 || deltaR > 1)) {
          float bestuR = __fmul_rn(g.lv[levelL].scale, __fadd_rn(__fadd_rn(sr, (float)bestinc), deltaR));
          float disparity = __fsub_rn(uL, bestuR);
          if (disparity >= 0 && disparity < maxD) {
            if (disparity <= 0) {
              disparity = 0.01f;
              bestuR = (float)__dsub_rn((double)uL, 0.01);
            }
            depth_out = __fdiv_rn(a.bf, disparity);
            uR_out = bestuR;
            sad_out = bestSad;
          }
        }
      }
    }
  }
  if (lane == 0) {
    a.uRight[oL] = uR_out;
    a.depth[oL] = depth_out;
    a.sad[oL] = sad_out;
  }
}

// Median-of-SAD outlier cut (:1072-1083): median = element size/2 of the ascending (SAD, iL) list, i.e. the
// (size/2)-th smallest SAD; matches with SAD >= 1.5*1.4*median are dropped.  One block per pair; the order
// statistic is found exactly with a two-level LDS histogram (SAD <= 121*255 < 2^15: high 8 bits, low 7 bits).
__global__ __launch_bounds__(256) void k_stereo_filter(StereoArgs a) {
  __shared__ int hist[256];
  __shared__ int s_v[4];
  const int tid = threadIdx.x, pair = blockIdx.x;
  const int nL = a.nL[a.firstL + pair];
  const int* sad = a.sad + (long long)pair * a.capL;
  hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < nL; i += 256) {
    const int s = sad[i];
    if (s >= 0) atomicAdd(&hist[min(s >> 7, 255)], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int m = 0;
    for (int k = 0; k < 256; k++) m += hist[k];
    int target = m / 2, k = 0, cum = 0;  // 0-based rank of the median
    if (m > 0) {
      while (cum + hist[k] <= target) cum += hist[k++];
    }
    s_v[0] = m;
    s_v[1] = k;
    s_v[2] = target - cum;  // rank inside the bucket
  }
  __syncthreads();
  const int m = s_v[0];
  if (m == 0) return;  // the reference reads vDistIdx[0] of an empty vector here (UB) — guarded
  const int bucket = s_v[1];
  __syncthreads();
  hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < nL; i += 256) {
    const int s = sad[i];
    if (s >= 0 && min(s >> 7, 255) == bucket) atomicAdd(&hist[s & 127], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int k = 0, cum = 0;
    while (cum + hist[k] <= s_v[2]) cum += hist[k++];
    s_v[3] = (bucket << 7) | k;
  }
  __syncthreads();
  const float median = (float)s_v[3];
  const float th = __fmul_rn(1.5f * 1.4f, median);
  for (int i = tid; i < nL; i += 256) {
    const int s = sad[i];
    if (s >= 0 && !((float)s < th)) {
      a.uRight[(long long)pair * a.capL + i] = -1.f;
      a.depth[(long long)pair * a.capL + i] = -1.f;
    }
  }
}

hipError_t launch_stereo_match(const Geom& g, const Pyr& pl, const Pyr& pr, const StereoArgs& a, int npairs,
                               hipStream_t s) {
  hipLaunchKernelGGL(k_stereo_match, dim3((a.capL + 3) / 4, npairs), dim3(256), 0, s, g, pl, pr, a);
  return hipGetLastError();
}
hipError_t launch_stereo_filter(const StereoArgs& a, int npairs, hipStream_t s) {
  hipLaunchKernelGGL(k_stereo_filter, dim3(npairs), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ================================================================================================ bf knn2
// Brute-force Hamming 2-NN, stable w.r.t. the train index.  Thread per query, 256 train rows
// staged per LDS tile.
__global__ __launch_bounds__(256) void k_bf_knn2(const uint8_t* __restrict__ dQ, int nQ,
                                                 const uint8_t* __restrict__ dT, int nT, int* __restrict__ idx2,
                                                 int* __restrict__ dist2, uint8_t* __restrict__ ok) {
  __shared__ uint32_t tile[256 * 9];  // 8 words + 1 pad per row: conflict-free broadcast reads
  const int q = blockIdx.x * 256 + threadIdx.x;
  uint32_t dq[8];
  if (q < nQ) {
#pragma unroll
    for (int i = 0; i < 8; i++) dq[i] = reinterpret_cast<const uint32_t*>(dQ)[(long long)q * 8 + i];
  }
  int b0 = 0x7FFFFFFF, b1 = 0x7FFFFFFF, i0 = -1, i1 = -1;
  for (int t0 = 0; t0 < nT; t0 += 256) {
    const int nt = min(256, nT - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < nt * 8; i += 256)
      tile[(i >> 3) * 9 + (i & 7)] = reinterpret_cast<const uint32_t*>(dT)[(long long)t0 * 8 + i];
    __syncthreads();
    if (q < nQ) {
      for (int t = 0; t < nt; t++) {
        const int d = hamming256(dq, tile + t * 9);
        if (d < b0) {
          b1 = b0;
          i1 = i0;
          b0 = d;
          i0 = t0 + t;
        } else if (d < b1) {
          b1 = d;
          i1 = t0 + t;
        }
      }
    }
  }
  if (q < nQ) {
    idx2[2 * q] = i0;
    idx2[2 * q + 1] = i1;
    dist2[2 * q] = i0 >= 0 ? b0 : -1;
    dist2[2 * q + 1] = i1 >= 0 ? b1 : -1;
    ok[q] = (i0 >= 0 && i1 >= 0 && (double)(float)b0 < __dmul_rn((double)(float)b1, 0.7)) ? 1 : 0;
  }
}

hipError_t launch_bf_knn2(const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, int* idx2, int* dist2,
                          uint8_t* ok, hipStream_t s) {
  if (nQ <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_bf_knn2, dim3((nQ + 255) / 256), dim3(256), 0, s, dQ, nQ, dT, nT, idx2, dist2, ok);
  return hipGetLastError();
}

// ================================================================================================ fisheye stereo
// Tail of Frame::ComputeStereoFishEyeMatches (src/Frame.cc:1298-1330): one thread per lapping-area left keypoint whose
// 2-NN passed the Lowe test runs KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:341-432).
// This is the floating-point corner of the path: float expressions in the reference's order (this TU is compiled
// with -ffp-contract=off), device libm for atan2f / tanf / cosf / sinf, and the null vector of the 4x4 system from a
// one-sided Jacobi SVD in double instead of Eigen::JacobiSVD<Matrix4f> -- parity is to float rounding, not bit-exact.
struct KB8Cam {
  float p[8];
  float precision;
};

__device__ __forceinline__ void kb8_project(const KB8Cam& c, const float X[3], float uv[2]) {  // :67-86
  const float x2_plus_y2 = X[0] * X[0] + X[1] * X[1];
  const float theta = atan2f(sqrtf(x2_plus_y2), X[2]);
  const float psi = atan2f(X[1], X[0]);
  const float theta2 = theta * theta;
  const float theta3 = theta * theta2;
  const float theta5 = theta3 * theta2;
  const float theta7 = theta5 * theta2;
  const float theta9 = theta7 * theta2;
  const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
  uv[0] = c.p[0] * r * cosf(psi) + c.p[2];
  uv[1] = c.p[1] * r * sinf(psi) + c.p[3];
}

__device__ __forceinline__ void kb8_unproject(const KB8Cam& c, float u, float v, float ray[3]) {  // :116-147
  const float pwx = (u - c.p[2]) / c.p[0], pwy = (v - c.p[3]) / c.p[1];
  float scale = 1.f;
  float theta_d = sqrtf(pwx * pwx + pwy * pwy);
  const float halfPi = (float)(3.1415926535897932384626433832795 / 2.0);
  theta_d = fminf(fmaxf(-halfPi, theta_d), halfPi);
  if ((double)theta_d > 1e-8) {
    float theta = theta_d;
    for (int j = 0; j < 10; j++) {  // Newton on theta (1 + k0 theta^2 + ...) = theta_d
      const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
      const float k0_theta2 = c.p[4] * theta2, k1_theta4 = c.p[5] * theta4;
      const float k2_theta6 = c.p[6] * theta6, k3_theta8 = c.p[7] * theta8;
      const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                              (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
      theta = theta - theta_fix;
      if (fabsf(theta_fix) < c.precision) break;
    }
    scale = tanf(theta) / theta_d;
  }
  ray[0] = pwx * scale;
  ray[1] = pwy * scale;
  ray[2] = 1.f;
}

// Right singular vector of the smallest singular value (= JacobiSVD::matrixV().col(3), :429-431) of a row-major 4x4.
__device__ void null_vector4(const float A[16], float v[4]) {
  double U[4][4], V[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      U[i][j] = (double)A[4 * i + j];
      V[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          al += U[i][p] * U[i][p];
          be += U[i][q] * U[i][q];
          ga += U[i][p] * U[i][q];
        }
        if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
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
  double n[4];
#pragma unroll
  for (int j = 0; j < 4; j++) n[j] = U[0][j] * U[0][j] + U[1][j] * U[1][j] + U[2][j] * U[2][j] + U[3][j] * U[3][j];
  int best = 0;
#pragma unroll
  for (int j = 1; j < 4; j++)
    if (n[j] < n[best]) best = j;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    double x = V[i][0];
    x = best == 1 ? V[i][1] : x;
    x = best == 2 ? V[i][2] : x;
    x = best == 3 ? V[i][3] : x;
    v[i] = (float)x;
  }
}

__device__ float kb8_triangulate_matches(const KB8Cam& c1, const KB8Cam& c2, float u1, float v1, float u2, float v2,
                                         const float* R12, const float* t12, float sigmaLevel, float unc, float p3D[3]) {
  float r1[3], r2[3], r21[3];
  kb8_unproject(c1, u1, v1, r1);
  kb8_unproject(c2, u2, v2, r2);
#pragma unroll
  for (int i = 0; i < 3; i++) r21[i] = R12[3 * i] * r2[0] + R12[3 * i + 1] * r2[1] + R12[3 * i + 2] * r2[2];
  const float dot = r1[0] * r21[0] + r1[1] * r21[1] + r1[2] * r21[2];
  const float n1 = sqrtf(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  const float n2 = sqrtf(r21[0] * r21[0] + r21[1] * r21[1] + r21[2] * r21[2]);
  const float cosParallaxRays = dot / (n1 * n2);
  if ((double)cosParallaxRays > 0.9998) return -1;  // :356
  float T2[3][4];  // Tcw2 = [R21 | -R21 t12]; Tcw1 = [I | 0] is folded into the rows of A below (:369-376)
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int j = 0; j < 3; j++) T2[i][j] = R12[3 * j + i];
    T2[i][3] = (-T2[i][0]) * t12[0] + (-T2[i][1]) * t12[1] + (-T2[i][2]) * t12[2];
  }
  float A[16];  // Triangulate, :420-427
  A[0] = -1.f; A[1] = 0.f; A[2] = r1[0]; A[3] = 0.f;
  A[4] = 0.f; A[5] = -1.f; A[6] = r1[1]; A[7] = 0.f;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    A[8 + j] = r2[0] * T2[2][j] - T2[0][j];
    A[12 + j] = r2[1] * T2[2][j] - T2[1][j];
  }
  float xh[4];
  null_vector4(A, xh);
  const float x3D[3] = {xh[0] / xh[3], xh[1] / xh[3], xh[2] / xh[3]};
  const float z1 = x3D[2];
  if (!(z1 > 0)) return -2;
  const float z2 = T2[2][0] * x3D[0] + T2[2][1] * x3D[1] + T2[2][2] * x3D[2] + T2[2][3];
  if (!(z2 > 0)) return -3;
  float uv1[2];
  kb8_project(c1, x3D, uv1);
  const float errX1 = uv1[0] - u1, errY1 = uv1[1] - v1;
  if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel) return -4;
  float x3D2[3];
#pragma unroll
  for (int i = 0; i < 3; i++) x3D2[i] = T2[i][0] * x3D[0] + T2[i][1] * x3D[1] + T2[i][2] * x3D[2] + T2[i][3];
  float uv2[2];
  kb8_project(c2, x3D2, uv2);
  const float errX2 = uv2[0] - u2, errY2 = uv2[1] - v2;
  if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)unc) return -5;
  p3D[0] = x3D[0];
  p3D[1] = x3D[1];
  p3D[2] = x3D[2];
  return z1;
}

__global__ __launch_bounds__(64) void k_fisheye_triangulate(FisheyeArgs a) {
  const int q = blockIdx.x * 64 + threadIdx.x;
  const int nQ = a.nL - a.monoL;
  bool desc = false, matched = false;
  if (q < nQ && a.ratioOk[q]) {
    desc = true;
    const int iL = q + a.monoL, iR = a.idx2[2 * q] + a.monoR;
    const orbx_keypoint k1 = a.kL[iL], k2 = a.kR[iR];
    KB8Cam c1, c2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      c1.p[i] = a.rig.cam1[i];
      c2.p[i] = a.rig.cam2[i];
    }
    c1.precision = c2.precision = a.rig.precision;
    const float sigma1 = a.sigma2[min(max(k1.octave, 0), a.nLevels - 1)];
    const float sigma2 = a.sigma2[min(max(k2.octave, 0), a.nLevels - 1)];
    float P[3] = {0.f, 0.f, 0.f};
    const float d = kb8_triangulate_matches(c1, c2, k1.x, k1.y, k2.x, k2.y, a.rig.R12, a.rig.t12, sigma1, sigma2, P);
    if (d > 0.0001f) {  // src/Frame.cc:1319
      matched = true;
      a.leftToRight[iL] = iR;
      atomicMax(a.rightToLeft + iR, iL);  // serial loop: the later left keypoint overwrites (:1322-1323)
      a.p3D[3 * iL] = P[0];
      a.p3D[3 * iL + 1] = P[1];
      a.p3D[3 * iL + 2] = P[2];
      a.depth[iL] = d;
    }
  }
  const uint64_t mm = __ballot(matched), md = __ballot(desc);
  if (threadIdx.x == 0) {
    if (mm) atomicAdd(a.counters, __popcll(mm));
    if (md) atomicAdd(a.counters + 1, __popcll(md));
  }
}

// Batched variant on the extractors' device-resident results: thread per lapping-area left keypoint of pair
// blockIdx.y does the 2-NN over the pair's right lapping rows (256-row LDS tiles, as k_bf_knn2), the Lowe test and the
// triangulation in one go.
__global__ __launch_bounds__(256) void k_fisheye_batch(FisheyeBatchArgs a) {
  // 64 queries per block; wave w scans the train rows t = w (mod 4) of every 256-row LDS tile (all lanes read the same
  // row: broadcast, 2 x ds_read_b128), the four partial (distance, index) top-2 lists are merged lexicographically --
  // exactly the stable first-minimum order of the serial scan -- and wave 0 triangulates.
  __shared__ uint4 tile[256 * 2];
  __shared__ uint32_t part[3][64][2];  // waves 1..3: packed (distance << 16 | index) best / second
  const int pr = blockIdx.y;
  const int imL = a.firstL + pr, imR = a.firstR + pr;
  const int nL = min(a.nL[imL], a.capL), nR = min(a.nR[imR], a.capR);
  const int monoL = min(max(a.monoL[imL], 0), nL), monoR = min(max(a.monoR[imR], 0), nR);
  const int nQ = nL - monoL, nT = nR - monoR;
  if ((int)blockIdx.x * 64 >= nQ) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + lane;
  const uint4* dQ = reinterpret_cast<const uint4*>(a.dL + ((long long)imL * a.capL + monoL) * 32);
  const uint4* dT = reinterpret_cast<const uint4*>(a.dR + ((long long)imR * a.capR + monoR) * 32);
  uint4 qa = {0, 0, 0, 0}, qb = {0, 0, 0, 0};
  if (q < nQ) {
    qa = dQ[(long long)q * 2];
    qb = dQ[(long long)q * 2 + 1];
  }
  uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;  // (distance << 16 | train index): lexicographic order, nT < 65536
  for (int t0 = 0; t0 < nT; t0 += 256) {
    const int nt = min(256, nT - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < nt * 2; i += 256) tile[i] = dT[(long long)t0 * 2 + i];
    __syncthreads();
    for (int t = w; t < nt; t += 4) {
      const uint4 ta = tile[2 * t], tb = tile[2 * t + 1];
      const int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                    __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
      const uint32_t key = ((uint32_t)d << 16) | (uint32_t)(t0 + t);
      const uint32_t lo = min(k0, key);
      k1 = min(k1, max(k0, key));
      k0 = lo;
    }
  }
  if (w > 0) {
    part[w - 1][lane][0] = k0;
    part[w - 1][lane][1] = k1;
  }
  __syncthreads();
  bool desc = false, matched = false;
  if (w == 0) {
#pragma unroll
    for (int o = 0; o < 3; o++) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const uint32_t key = part[o][lane][e];
        const uint32_t lo = min(k0, key);
        k1 = min(k1, max(k0, key));
        k0 = lo;
      }
    }
    const int b0 = (int)(k0 >> 16), b1 = (int)(k1 >> 16), i0 = (int)(k0 & 0xFFFF);
    if (q < nQ && k1 != 0xFFFFFFFFu && (double)(float)b0 < __dmul_rn((double)(float)b1, 0.7)) {  // src/Frame.cc:1302
      desc = true;
      const int iL = q + monoL, iR = i0 + monoR;
      const orbx_keypoint kp1 = a.kL[(long long)imL * a.capL + iL], kp2 = a.kR[(long long)imR * a.capR + iR];
      KB8Cam c1, c2;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        c1.p[i] = a.rig.cam1[i];
        c2.p[i] = a.rig.cam2[i];
      }
      c1.precision = c2.precision = a.rig.precision;
      const float sigma1 = a.sigma2[min(max(kp1.octave, 0), a.nLevels - 1)];
      const float sigma2 = a.sigma2[min(max(kp2.octave, 0), a.nLevels - 1)];
      float P[3] = {0.f, 0.f, 0.f};
      const float d = kb8_triangulate_matches(c1, c2, kp1.x, kp1.y, kp2.x, kp2.y, a.rig.R12, a.rig.t12, sigma1, sigma2, P);
      if (d > 0.0001f) {
        matched = true;
        const long long o = (long long)pr * a.capL + iL;
        a.leftToRight[o] = iR;
        atomicMax(a.rightToLeft + (long long)pr * a.capR + iR, iL);
        a.p3D[3 * o] = P[0];
        a.p3D[3 * o + 1] = P[1];
        a.p3D[3 * o + 2] = P[2];
        a.depth[o] = d;
      }
    }
    const uint64_t mm = __ballot(matched), md = __ballot(desc);
    if (lane == 0) {
      if (mm) atomicAdd(a.counters + 2 * pr, __popcll(mm));
      if (md) atomicAdd(a.counters + 2 * pr + 1, __popcll(md));
    }
  }
}

// One launch presets every output of the batch: -1 matches / depths, zero points and counters.
__global__ __launch_bounds__(256) void k_fisheye_init(FisheyeBatchArgs a, int npairs) {
  const long long nl = (long long)npairs * a.capL, nr = (long long)npairs * a.capR;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nl * 3; i += (long long)gridDim.x * 256) {
    a.p3D[i] = 0.f;
    if (i < nl) {
      a.leftToRight[i] = -1;
      a.depth[i] = -1.0f;
    }
    if (i < nr) a.rightToLeft[i] = -1;
    if (i < 2 * npairs) a.counters[i] = 0;
  }
}

hipError_t launch_fisheye_batch(const FisheyeBatchArgs& a, int npairs, hipStream_t s) {
  const long long work = (long long)npairs * (a.capL > a.capR ? a.capL : a.capR) * 3;
  hipLaunchKernelGGL(k_fisheye_init, dim3((unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048)), dim3(256), 0, s, a,
                     npairs);
  hipLaunchKernelGGL(k_fisheye_batch, dim3((a.capL + 63) / 64, npairs), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_fisheye_triangulate(const FisheyeArgs& a, hipStream_t s) {
  const int nQ = a.nL - a.monoL;
  if (nQ <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fisheye_triangulate, dim3((nQ + 63) / 64), dim3(64), 0, s, a);
  return hipGetLastError();
}

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

hipError_t launch_cvt_gray(const uint8_t* src, int w, int h, long long sp, long long sip, int cn, int rgb, uint8_t* dst,
                           long long dp, long long dip, int nimg, hipStream_t s) {
  hipLaunchKernelGGL(k_cvt_gray, dim3((w + 255) / 256, h, nimg), dim3(256), 0, s, src, w, h, sp, sip, cn, rgb, dst, dp, dip);
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
__global__ __launch_bounds__(256) void k_remap1(RemapArgs a, int nimg) {
  const int x0 = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x0 >= a.dw || y >= a.dh) return;
  const int m = blockIdx.z % a.nMaps, grp = blockIdx.z / a.nMaps;
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
  uint32_t selTop[4], selBot[4];  // v_perm selectors: byte d -> bits 0..7, byte d + 1 -> bits 16..23 of the 8-byte window
  int e0[4], e1[4];               // window row of the top / bottom pair
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int d = sxc[k] - bx;
    e0[k] = syc0[k] - by;
    e1[k] = syc1[k] - by;
    fast = fast && d <= 6 && e0[k] <= 1 && e1[k] <= 2;
    selTop[k] = selBot[k] = (uint32_t)d | 0x0c000c00u | ((uint32_t)(d + 1) << 16);
  }
  const int wo0 = by * pitch + bx, wo1 = min(by + 1, a.sh - 1) * pitch + bx, wo2 = min(by + 2, a.sh - 1) * pitch + bx;
  const uint8_t* __restrict__ src = a.src;
  uint8_t* __restrict__ dst = a.dst;
  const int first = grp * kRemapGroup, perMap = (nimg - m + a.nMaps - 1) / a.nMaps;
  const int count = min(kRemapGroup, perMap - first);
  for (int g = 0; g < count; g++) {
    const int img = m + a.nMaps * (first + g);
    const uint8_t* S = src + (long long)img * a.srcImgPitch;
    uint2 r0, r1, r2;
    __builtin_memcpy(&r0, S + wo0, 8);
    __builtin_memcpy(&r1, S + wo1, 8);
    __builtin_memcpy(&r2, S + wo2, 8);
    uint32_t packed = 0;
    if (fast) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t tlo = e0[k] ? r1.x : r0.x, thi = e0[k] ? r1.y : r0.y;
        const uint32_t blo = e1[k] == 0 ? r0.x : (e1[k] == 1 ? r1.x : r2.x), bhi = e1[k] == 0 ? r0.y : (e1[k] == 1 ? r1.y : r2.y);
        // two v_dot2_u32_u16: (p00, p01) . (w0, w1) + (p10, p11) . (w2, w3); every weight is below 2^15
        uint32_t acc = udot2_u16(__builtin_amdgcn_perm(thi, tlo, selTop[k]), wlo[k], 16384u);
        acc = udot2_u16(__builtin_amdgcn_perm(bhi, blo, selBot[k]), whi[k], acc);
        packed |= (acc >> 15) << (8 * k);
      }
    } else {
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
    }
    uint8_t* D = dst + (long long)img * a.dstImgPitch + (long long)y * a.dstPitch + x0;
    if (full && a.dstVec4) {
      *reinterpret_cast<uint32_t*>(D) = packed;
    } else {
      for (int k = 0; k < 4 && x0 + k < a.dw; k++) D[k] = (uint8_t)(packed >> (8 * k));
    }
  }
}
hipError_t launch_remap(const RemapArgs& a, int nimg, hipStream_t s) {
  if (a.cn == 1 && a.sw >= 8) {
    const int perMap = (nimg + a.nMaps - 1) / a.nMaps, groups = (perMap + kRemapGroup - 1) / kRemapGroup;
    hipLaunchKernelGGL(k_remap1, dim3((a.dw + 255) / 256, (a.dh + 3) / 4, a.nMaps * groups), dim3(256), 0, s, a, nimg);
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
  for (int k = tid; k < 16 * 257; k += 256) hist[k] = 0;
  __syncthreads();
  const uint8_t* S = a.src + (long long)img * a.srcImgPitch;
  // thread per 4 pixels of a tile row (sub-dword loads run at a fraction of the dword rate); quads that touch the tile's right
  // edge or the reflected extension go pixel by pixel
  const int qpr = (a.tw + 3) >> 2, nquads = qpr * a.th;
  const float inv_qpr = 1.0f / (float)qpr;
  const int hcopy = (lane & 15) * 257;
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
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o);
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
  int sum = v;  // inclusive prefix sum over the 256 bins
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(sum, o);
    if (lane >= o) sum += t;
  }
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
size_t clahe_cells_bytes(const ClaheArgs& a, int nimg) {
  return (size_t)nimg * (a.tilesY + 1) * (a.tilesX + 1) * 256 * sizeof(uint32_t);
}
hipError_t launch_clahe(const ClaheArgs& a, int nimg, uint32_t* cells, hipStream_t s) {
  hipLaunchKernelGGL(k_clahe_lut, dim3(a.tilesX * a.tilesY, nimg), dim3(256), 0, s, a);
  const dim3 grid((a.w + 255) / 256, (a.h + 3) / 4, nimg);
  if (cells) {
    hipLaunchKernelGGL(k_clahe_pack, dim3((a.tilesX + 1) * (a.tilesY + 1), nimg), dim3(256), 0, s, a, cells);
    hipLaunchKernelGGL(k_clahe_apply4, grid, dim3(256), 0, s, a, cells);
  } else {
    hipLaunchKernelGGL(k_clahe_apply, grid, dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// ================================================================================================ bag of words
// TemplatedVocabulary::transform(feature, word, weight, nid, levelsup) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1202-1250):
// 16 lanes per feature, lane c scores child c of the current node (ORB vocabularies have k = 10), the first minimum by
// child order is a min-reduction over (distance << 16 | child position).  L dependent gathers of k x 32 B per feature.
__global__ __launch_bounds__(256) void k_bow_descend(BowArgs a) {
  const int img = blockIdx.y, sub = threadIdx.x & 15;
  const int f = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int nf = a.counts ? a.counts[img] : a.n;
  if (f >= nf) return;  // whole 16-lane groups leave together
  const uint32_t* D = reinterpret_cast<const uint32_t*>(a.desc + (long long)img * a.descImgPitch) + (long long)f * 8;
  uint32_t d[8];
#pragma unroll
  for (int i = 0; i < 8; i++) d[i] = D[i];
  const BowVoc& v = a.voc;
  const int nidLevel = v.L - a.levelsup;
  int cur = 0, level = 0, nid = 0;
  bool nidSet = nidLevel <= 0;
  for (;;) {
    const int c0 = v.childStart[cur], c1 = v.childStart[cur + 1];
    if (c0 == c1) break;  // leaf (the root of a non-empty vocabulary has children)
    uint32_t best = 0xffffffffu;
    for (int c = c0 + sub; c < c1; c += 16) {
      const uint32_t* nd = v.desc + (long long)v.children[c] * 8;
      best = min(best, ((uint32_t)hamming256(d, nd) << 16) | (uint32_t)(c - c0));
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 16));
    cur = v.children[c0 + (int)(best & 0xffffu)];
    if (++level == nidLevel) { nid = cur; nidSet = true; }
  }
  if (!nidSet) nid = cur;  // a leaf above level L - levelsup: the reference leaves *nid unset
  if (sub == 0) {
    const long long o = (long long)img * a.cap + f;
    a.word[o] = v.wordId[cur];
    a.weight[o] = v.weight[cur];
    a.node[o] = nid;
  }
}

// Sort of the (key << 16 | index) words: bitonic network over P = 2^k slots in LDS, kBowThreads threads.  A compare-exchange
// at distance j < 64 stays inside an aligned group of 64 slots, and thread t always owns the pairs of the same groups, so
// those stages need no workgroup barrier (the wave's own LDS operations are ordered); only the 15 of 66 stages (P = 2048)
// with j >= 64 do -- the kernel is one workgroup per image and pure latency.
constexpr int kBowThreads = 1024;
__device__ __forceinline__ void bow_sort(uint64_t* key, int P) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int q = tid; q < (P >> 1); q += kBowThreads) {
        // pair q of this stage: t = q with a zero bit inserted at position log2(j).  For j < 64 pair q and slot t share their
        // aligned group of 32 pairs / 64 slots, i.e. the wave that owned the group in the previous stage owns it again.
        const int t = ((q & ~(j - 1)) << 1) | (q & (j - 1)), u = t | j;
        const uint64_t ka = key[t], kb = key[u];
        if ((ka > kb) == ((t & k) == 0)) { key[t] = kb; key[u] = ka; }
      }
      if (j >= 64 || j == 1) __syncthreads();
      else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    }
}
// rank[t] = number of set flags before slot t (exclusive), returns the total; flag / rank share one LDS int array.
__device__ __forceinline__ int bow_rank_heads(int* fr, int P, int* wsum) {
  const int per = max(P / kBowThreads, 1), b = threadIdx.x * per;
  int s = 0;
  if (b < P)
    for (int i = 0; i < per; i++) s += fr[b + i];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = s;
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int w = 0; w < kBowThreads / 64; w++) {
    if (w < wv) base += wsum[w];
    total += wsum[w];
  }
  int run = base + incl - s;
  if (b < P)
    for (int i = 0; i < per; i++) {
      const int t = fr[b + i];
      fr[b + i] = run;
      run += t;
    }
  __syncthreads();
  return total;
}

// TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup) (:1125-1188) after the descents: one block
// per image.  The std::map semantics become a sort: (word, feature index) pairs ascending give the BowVector's key order and,
// per word, the reference's additions in feature order (value = w + w + ... sequentially -- every addend of a word is
// the same idf weight); the L1 / L2 norm is accumulated sequentially in ascending word order like BowVector::normalize,
// by one thread (values staged in LDS), so the doubles come out bit-identical.  (node, feature index) pairs give the
// FeatureVector as CSR.
__global__ __launch_bounds__(kBowThreads) void k_bow_assemble(BowArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t bow_smem[];
  __shared__ int wsum[kBowThreads / 64];
  __shared__ double normShared;
  const int img = blockIdx.x, tid = threadIdx.x;
  const int nf = a.counts ? a.counts[img] : a.n;
  int P = 128;
  while (P < nf) P <<= 1;
  uint64_t* key = reinterpret_cast<uint64_t*>(bow_smem);
  double* lval = reinterpret_cast<double*>(bow_smem);  // the values of the unique words, once the keys are consumed
  int* fr = reinterpret_cast<int*>(key + P);
  const long long o = (long long)img * a.cap;
  const int* word = a.word + o;
  const double* wt = a.weight + o;
  const int* node = a.node + o;
  uint32_t* words = a.words + o;
  double* values = a.values + o;
  uint32_t* nodes = a.nodes + o;
  int* nodeStart = a.nodeStart + (long long)img * (a.cap + 1);
  uint32_t* feats = a.feats + o;
  const bool additive = a.voc.weighting == 0 || a.voc.weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
  const bool must = a.voc.scoring != 5, l2 = a.voc.scoring == 1;       // mustNormalize (ScoringObject.h:73-90)
  constexpr uint64_t kNone = ~0ull;

  // ---- BowVector
  for (int t = tid; t < P; t += kBowThreads)
    key[t] = (t < nf && wt[t] > 0) ? ((uint64_t)(uint32_t)word[t] << 16) | (uint64_t)t : kNone;  // "w > 0: not stopped"
  __syncthreads();
  bow_sort(key, P);
  for (int t = tid; t < P; t += kBowThreads) fr[t] = key[t] != kNone && (t == 0 || (key[t] >> 16) != (key[t - 1] >> 16));
  __syncthreads();
  const int U = bow_rank_heads(fr, P, wsum);
  double myV[8];  // values of the heads this thread owns (P / kBowThreads <= 8 slots per thread)
  int myU[8], nMine = 0;
  for (int t = tid; t < P; t += kBowThreads) {
    if (key[t] == kNone || (t > 0 && (key[t] >> 16) == (key[t - 1] >> 16))) continue;
    const double w = wt[key[t] & 0xffff];  // the first feature of the word in feature order
    double v = w;
    if (additive)
      for (int r = t + 1; r < P && (key[r] >> 16) == (key[t] >> 16); r++) v += w;
    words[fr[t]] = (uint32_t)(key[t] >> 16);
    myU[nMine] = fr[t];
    myV[nMine++] = v;
  }
  __syncthreads();  // every key has been read: the array now holds the values
  for (int i = 0; i < nMine; i++) lval[myU[i]] = myV[i];
  __syncthreads();
  double scale = 1.0;
  bool divide = false;
  if (additive && U > 0 && !must) {
    scale = (double)U;
    divide = true;
  }
  if (must) {
    if (tid == 0) {
      double norm = 0.0;
      if (!l2) {
        for (int u = 0; u < U; u++) norm += fabs(lval[u]);
      } else {
        for (int u = 0; u < U; u++) norm += lval[u] * lval[u];
        norm = sqrt(norm);
      }
      normShared = norm;
    }
    __syncthreads();
    scale = normShared;
    divide = scale > 0.0;
  }
  for (int u = tid; u < U; u += kBowThreads) values[u] = divide ? lval[u] / scale : lval[u];
  __syncthreads();

  // ---- FeatureVector
  for (int t = tid; t < P; t += kBowThreads)
    key[t] = (t < nf && wt[t] > 0) ? ((uint64_t)(uint32_t)node[t] << 16) | (uint64_t)t : kNone;
  __syncthreads();
  bow_sort(key, P);
  int nUsed = 0;
  for (int t = tid; t < P; t += kBowThreads) {
    fr[t] = key[t] != kNone && (t == 0 || (key[t] >> 16) != (key[t - 1] >> 16));
    nUsed += key[t] != kNone;
  }
  __syncthreads();
  const int V = bow_rank_heads(fr, P, wsum);
  for (int t = tid; t < P; t += kBowThreads) {
    if (key[t] == kNone) continue;
    feats[t] = (uint32_t)(key[t] & 0xffff);
    if (t == 0 || (key[t] >> 16) != (key[t - 1] >> 16)) {
      nodes[fr[t]] = (uint32_t)(key[t] >> 16);
      nodeStart[fr[t]] = t;
    }
  }
  for (int off = 32; off > 0; off >>= 1) nUsed += __shfl_xor(nUsed, off);
  __syncthreads();
  if ((tid & 63) == 0) wsum[tid >> 6] = nUsed;
  __syncthreads();
  if (tid == 0) {
    int used = 0;
    for (int w = 0; w < kBowThreads / 64; w++) used += wsum[w];
    nodeStart[V] = used;
    a.outCounts[img * 3 + 0] = U;
    a.outCounts[img * 3 + 1] = V;
    a.outCounts[img * 3 + 2] = used;
  }
}

hipError_t launch_bow_transform(const BowArgs& a, int nimg, hipStream_t s) {
  if (a.n <= 0 || nimg <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_bow_descend, dim3((a.n + 15) / 16, nimg), dim3(256), 0, s, a);
  int P = 128;
  while (P < a.n) P <<= 1;
  const size_t lds = (size_t)P * 12;
  if (lds > 48 * 1024) {  // up to 96 KB for 8192 features: above the default dynamic-LDS limit
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_bow_assemble), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_bow_assemble, dim3(nimg), dim3(kBowThreads), lds, s, a);
  return hipGetLastError();
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:230-404).  Features are only compared inside a shared
// vocabulary node and a frame feature belongs to one node, so the nodes are independent: one wave per keyframe node finds
// its partner in the frame's node list (binary search) and walks the node's keyframe features in order, like the reference
// (the "already matched" gate makes that walk order dependent); its lanes score the node's frame features, best / second
// by the serial rule (first minimum; an equal later distance becomes the second).  The right-eye branch keeps the
// reference's "|| true" (:363-365): no ratio test, and only inside "bestDist1 <= TH_LOW".
constexpr int kBowNodeCap = 4096;  // frame features of one node tracked in LDS (a node above this: serial fallback on lane 0)
__global__ __launch_bounds__(64) void k_bow_match(BowMatchArgs a) {
  __shared__ uint8_t taken[kBowNodeCap];
  const int lane = threadIdx.x, ia = blockIdx.x;
  const uint32_t node = a.kfNodes[ia];
  int lo = 0, hi = a.nFNodes;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.fNodes[mid] < node) lo = mid + 1; else hi = mid;
  }
  if (lo >= a.nFNodes || a.fNodes[lo] != node) return;
  const int f0 = a.fStart[lo], nfl = a.fStart[lo + 1] - f0;
  const int k0 = a.kfStart[ia], k1 = a.kfStart[ia + 1];
  if (nfl > kBowNodeCap) {  // never with a real vocabulary (levelsup 4 of 6: 100 nodes); keep the exact semantics anyway
    if (lane == 0) a.flags[32] = 1;
    return;
  }
  for (int i = lane; i < nfl; i += 64) taken[i] = 0;
  __syncthreads();
  // the node's first 64 frame features stay in registers for the whole walk (a node of a real vocabulary holds ~15)
  int iF0 = -1;
  uint32_t fD0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < nfl) {
    iF0 = (int)a.fFeat[f0 + lane];
    const uint32_t* dF = a.fDesc + (long long)iF0 * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) fD0[i] = dF[i];
  }
  const bool twoEyes = a.nLeftF != -1;
  int made = 0;
  for (int kc = k0; kc < k1; kc += 64) {  // keyframe features of the node: 64 at a time into registers, then walked in order
    int myKF = -1, myValid = 0;
    uint32_t myD[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kc + lane < k1) {
      myKF = (int)a.kfFeat[kc + lane];
      myValid = a.kfValid[myKF];
      if (myValid) {
        const uint32_t* dK = a.kfDesc + (long long)myKF * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) myD[i] = dK[i];
      }
    }
    const int cnt = min(64, k1 - kc);
    for (int c = 0; c < cnt; c++) {
      if (!__builtin_amdgcn_readlane(myValid, c)) continue;
      const int iKF = __builtin_amdgcn_readlane(myKF, c);
      uint32_t d[8];
#pragma unroll
      for (int i = 0; i < 8; i++) d[i] = (uint32_t)__builtin_amdgcn_readlane((int)myD[i], c);
      // running best / second of the left and the right eye, merged chunk by chunk in list order
      int b1 = 256, bi = -1, b2 = 256, b1r = 256, bir = -1, b2r = 256;
      for (int q0 = 0; q0 < nfl; q0 += 64) {
        const int q = q0 + lane;
        int dist = 0x7fff, iF = -1;
        bool right = false;
        if (q < nfl && !taken[q]) {
          if (q0 == 0) {
            iF = iF0;
            dist = hamming256(d, fD0);
          } else {
            iF = (int)a.fFeat[f0 + q];
            dist = hamming256(d, a.fDesc + (long long)iF * 8);
          }
          right = twoEyes && iF >= a.nLeftF;
        }
        for (int side = 0; side < (twoEyes ? 2 : 1); side++) {
          const bool mine = iF >= 0 && right == (side == 1);
          uint32_t k1st = mine ? ((uint32_t)dist << 8) | (uint32_t)lane : 0xffffffffu;  // first minimum: lower lane = earlier
          for (int o = 32; o > 0; o >>= 1) k1st = min(k1st, (uint32_t)__shfl_xor((int)k1st, o));
          uint32_t k2nd = (mine && (k1st & 255u) != (uint32_t)lane) ? (uint32_t)dist : 0xffffffffu;
          for (int o = 32; o > 0; o >>= 1) k2nd = min(k2nd, (uint32_t)__shfl_xor((int)k2nd, o));
          if (k1st != 0xffffffffu) {
            const int c1 = (int)(k1st >> 8), cl = (int)(k1st & 255u), c2 = k2nd == 0xffffffffu ? 256 : (int)k2nd;
            const int ci = q0 == 0 ? __builtin_amdgcn_readlane(iF0, cl) : (int)a.fFeat[f0 + q0 + cl];
            int& B1 = side ? b1r : b1; int& BI = side ? bir : bi; int& B2 = side ? b2r : b2;
            if (c1 < B1) { B2 = min(B1, c2); B1 = c1; BI = ci; }
            else { B2 = min(B2, c1); }
          }
        }
      }
      if (b1 <= 50) {  // TH_LOW
        const bool leftOk = (float)b1 < __fmul_rn(a.nnratio, (float)b2), rightOk = b1r <= 50;
        if (lane == 0) {
          for (int side = 0; side < 2; side++) {
            if (!(side ? rightOk : leftOk)) continue;
            const int iF = side ? bir : bi;
            a.match[iF] = iKF;
            if (a.checkOri) {
              float rot = __fsub_rn(a.kfKps[iKF].angle, a.fKps[iF].angle);
              if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
              int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
              if (bin == 30) bin = 0;
              a.bin[iF] = bin;
              atomicAdd(&a.flags[2 + bin], 1);
            }
          }
        }
        made += (leftOk ? 1 : 0) + (rightOk ? 1 : 0);
        if (leftOk || rightOk) {  // mark the taken frame features of this node (positions in the node's list)
          for (int q = lane; q < nfl; q += 64) {
            const int iF = q < 64 ? iF0 : (int)a.fFeat[f0 + q];
            if ((leftOk && iF == bi) || (rightOk && iF == bir)) taken[q] = 1;
          }
          __syncthreads();
        }
      }
    }
  }
  if (lane == 0 && made) atomicAdd(&a.flags[0], made);
}

__global__ __launch_bounds__(256) void k_bow_cull(BowMatchArgs a) {  // :384-401 with ComputeThreeMaxima :1920-1955
  int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < 30; i++) {
    const int s = a.flags[2 + i];
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s; ind3 = ind2; ind2 = i;
    } else if (s > max3) {
      max3 = s; ind3 = i;
    }
  }
  if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
    ind2 = -1;
    ind3 = -1;
  } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
    ind3 = -1;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool rem = false;
  if (i < a.nF && a.match[i] >= 0) {
    const int bin = a.bin[i];
    if (bin != ind1 && bin != ind2 && bin != ind3) {
      a.match[i] = -1;
      rem = true;
    }
  }
  const uint64_t m = __ballot(rem);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.flags[1], __popcll(m));
}
__global__ void k_bow_result(BowMatchArgs a) { a.result[0] = a.flags[32] ? -1 : a.flags[0] - a.flags[1]; }
__global__ __launch_bounds__(256) void k_bow_match_reset(BowMatchArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.nF) a.match[i] = -1;
  if (i < 33) a.flags[i] = 0;
}

hipError_t launch_bow_match(const BowMatchArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_bow_match_reset, dim3((max(a.nF, 33) + 255) / 256), dim3(256), 0, s, a);
  if (a.nKfNodes > 0 && a.nFNodes > 0) hipLaunchKernelGGL(k_bow_match, dim3(a.nKfNodes), dim3(64), 0, s, a);
  if (a.checkOri && a.nF > 0) hipLaunchKernelGGL(k_bow_cull, dim3((a.nF + 255) / 256), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_bow_result, dim3(1), dim3(1), 0, s, a);
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

// ================================================================================================ search init
// Frame grid (64 x 48, PosInGrid rounds to the nearest cell, src/Frame.cc:833-844) as CSR lists with
// ascending keypoint indices.
__device__ __forceinline__ int grid_cell(const orbx_keypoint& k, const InitArgs& a) {
  const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, a.minX), a.invW));
  const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, a.minY), a.invH));
  if (px < 0 || px >= 64 || py < 0 || py >= 48) return -1;
  return px * 48 + py;
}

__global__ __launch_bounds__(256) void k_init_grid(InitArgs a) {  // single block
  // Counting sort of the keypoints by grid cell, ascending keypoint index inside a cell (mGrid[i][j].push_back order,
  // src/Frame.cc:536-546): count -> block scan -> unordered atomic fill -> per-cell insertion sort of the short lists.
  __shared__ int cnt[64 * 48];
  __shared__ int wsum[4];
  constexpr int kCells = 64 * 48, kPer = kCells / 256;  // 12 consecutive cells per thread
  const int tid = threadIdx.x, lane = tid & 63;
  for (int c = tid; c < kCells; c += 256) cnt[c] = 0;
  __syncthreads();
  for (int i = tid; i < a.n2; i += 256) {
    const int c = grid_cell(a.k2[i], a);
    if (c >= 0) atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  int local[kPer], sum = 0;
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    local[k] = cnt[tid * kPer + k];
    sum += local[k];
  }
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < (tid >> 6); w++) run += wsum[w];
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    a.cellStart[tid * kPer + k] = run;
    cnt[tid * kPer + k] = run;  // becomes the fill cursor of the cell
    run += local[k];
  }
  if (tid == 255) a.cellStart[kCells] = run;
  __syncthreads();
  for (int i = tid; i < a.n2; i += 256) {
    const int c = grid_cell(a.k2[i], a);
    if (c >= 0) a.cellItems[atomicAdd(&cnt[c], 1)] = i;
  }
  __threadfence_block();
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPer; k++) {  // cells hold a handful of keypoints: insertion sort, thread per cell
    const int c = tid * kPer + k;
    const int b = cnt[c] - local[k];
    for (int i = 1; i < local[k]; i++) {
      const int v = a.cellItems[b + i];
      int j = i - 1;
      while (j >= 0 && a.cellItems[b + j] > v) {
        a.cellItems[b + j + 1] = a.cellItems[b + j];
        j--;
      }
      a.cellItems[b + j + 1] = v;
    }
  }
  for (int i = tid; i < a.n2; i += 256) {
    a.matchedDist[i] = 0x7FFFFFFF;
    a.matches21[i] = -1;
  }
  for (int i = tid; i < a.n1; i += 256) a.matches12[i] = -1;
  if (tid == 0) {
    a.result[0] = 0;
    a.result[1] = 0;
  }
}

// GetFeaturesInArea(x, y, r, 0, 0) (src/Frame.cc:765-831) for one level-0 keypoint of F1 per wave, in the
// reference's candidate order (ix outer, iy inner, in-cell order).  pass 0 counts, pass 1 writes (i2, dist).
__global__ __launch_bounds__(256) void k_init_cands(InitArgs a, int pass) {
  const int lane = threadIdx.x & 63;
  const int i1 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i1 >= a.n1) return;
  const orbx_keypoint k1 = a.k1[i1];
  int total = 0;
  if (k1.octave <= 0) {
    const float x = a.prev[2 * i1], y = a.prev[2 * i1 + 1], r = (float)a.window;
    const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.minX), r), a.invW)));
    const int cx1 = min(63, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.minX), r), a.invW)));
    const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.minY), r), a.invH)));
    const int cy1 = min(47, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.minY), r), a.invH)));
    if (cx0 < 64 && cx1 >= 0 && cy0 < 48 && cy1 >= 0) {
      uint32_t d1[8];
#pragma unroll
      for (int i = 0; i < 8; i++) d1[i] = reinterpret_cast<const uint32_t*>(a.d1)[(long long)i1 * 8 + i];
      const int wbase = pass ? a.candOff[i1] : 0;
      for (int ix = cx0; ix <= cx1; ix++)
        for (int iy = cy0; iy <= cy1; iy++) {
          const int b = a.cellStart[ix * 48 + iy], e = a.cellStart[ix * 48 + iy + 1];
          for (int base = b; base < e; base += 64) {
            const int j = base + lane;
            bool ok = false;
            int i2 = 0;
            if (j < e) {
              i2 = a.cellItems[j];
              const orbx_keypoint k2 = a.k2[i2];
              ok = k2.octave == 0 && fabsf(__fsub_rn(k2.x, x)) < r && fabsf(__fsub_rn(k2.y, y)) < r;
            }
            const uint64_t m = __ballot(ok);
            if (pass && ok) {
              const int o = wbase + total + prefix_count(m);
              if (o < a.candCap) {
                a.candIdx[o] = i2;
                a.candDist[o] = hamming256(d1, reinterpret_cast<const uint32_t*>(a.d2) + (long long)i2 * 8);
              }
            }
            total += __popcll(m);
          }
        }
    }
  }
  if (!pass && lane == 0) a.candOff[i1] = total;
}

__global__ __launch_bounds__(256) void k_init_scan(InitArgs a) {  // single block: exclusive scan of candOff
  __shared__ int tsum[256];
  const int tid = threadIdx.x, n = a.n1;
  const int per = (n + 255) >> 8, b = min(tid * per, n), e = min(b + per, n);
  int s = 0;
  for (int i = b; i < e; i++) s += a.candOff[i];
  tsum[tid] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int t = tid >= d ? tsum[tid - d] : 0;
    __syncthreads();
    tsum[tid] += t;
    __syncthreads();
  }
  int run = tid ? tsum[tid - 1] : 0;
  for (int i = b; i < e; i++) {
    const int t = a.candOff[i];
    a.candOff[i] = run;
    run += t;
  }
  if (tid == 255) {
    a.candOff[n] = tsum[255];
    if (tsum[255] > a.candCap) a.result[1] = tsum[255];
  }
}

// The greedy bookkeeping (vMatchedDistance gate, match stealing, rotation histogram) is order dependent:
// one wave walks i1 in serial order, the lanes reduce each keypoint's candidate list.
__global__ __launch_bounds__(64) void k_init_resolve(InitArgs a) {
  __shared__ int hist[30];
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int8_t* bins = reinterpret_cast<int8_t*>(smem);  // n1 entries: histogram bin of i1 or -1
  const int lane = threadIdx.x;
  for (int i = lane; i < 30; i += 64) hist[i] = 0;
  for (int i = lane; i < a.n1; i += 64) bins[i] = -1;
  __syncthreads();
  int nmatches = 0;
  for (int i1 = 0; i1 < a.n1; i1++) {
    const int b = a.candOff[i1], e = a.candOff[i1 + 1];
    if (e <= b) continue;
    // best = first strict minimum in list order; second = second order statistic (strict updates)
    uint64_t best = ~0ull;  // (dist << 32 | position)
    for (int j = b + lane; j < e; j += 64) {
      const int i2 = a.candIdx[j], d = a.candDist[j];
      if (a.matchedDist[i2] <= d) continue;
      const uint64_t v = ((uint64_t)(uint32_t)d << 32) | (uint32_t)(j - b);
      best = v < best ? v : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t t = __shfl_xor((unsigned long long)best, o);
      best = t < best ? t : best;
    }
    if (best == ~0ull) continue;
    const int bestDist = (int)(best >> 32), bestPos = (int)(best & 0xFFFFFFFFu);
    int second = 0x7FFFFFFF;
    for (int j = b + lane; j < e; j += 64) {
      if (j - b == bestPos) continue;
      const int i2 = a.candIdx[j], d = a.candDist[j];
      if (a.matchedDist[i2] <= d) continue;
      second = min(second, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) second = min(second, __shfl_xor(second, o));
    if (bestDist <= 50 && (float)bestDist < __fmul_rn((float)second, a.nnratio)) {
      if (lane == 0) {
        const int bestIdx2 = a.candIdx[b + bestPos];
        const int owner = a.matches21[bestIdx2];
        if (owner >= 0) {
          a.matches12[owner] = -1;
          nmatches--;
        }
        a.matches12[i1] = bestIdx2;
        a.matches21[bestIdx2] = i1;
        a.matchedDist[bestIdx2] = bestDist;
        nmatches++;
        if (a.checkOri) {
          float rot = __fsub_rn(a.k1[i1].angle, a.k2[bestIdx2].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
          if (bin == 30) bin = 0;
          bins[i1] = (int8_t)bin;
          hist[bin]++;
        }
      }
      __threadfence_block();
    }
    __syncthreads();
  }
  nmatches = __shfl(nmatches, 0);
  if (a.checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < 30; i++) {  // ComputeThreeMaxima, src/ORBmatcher.cc:1920-1955
      const int s = hist[i];
      if (s > max1) {
        max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i;
      } else if (s > max2) {
        max3 = max2; max2 = s; ind3 = ind2; ind2 = i;
      } else if (s > max3) {
        max3 = s; ind3 = i;
      }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
      ind2 = -1;
      ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
      ind3 = -1;
    }
    int removed = 0;
    for (int i = lane; i < a.n1; i += 64) {
      const int bn = bins[i];
      if (bn >= 0 && bn != ind1 && bn != ind2 && bn != ind3 && a.matches12[i] >= 0) {
        a.matches12[i] = -1;
        removed++;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
    nmatches -= removed;
  }
  __syncthreads();
  for (int i = lane; i < a.n1; i += 64) {
    const int m = a.matches12[i];
    if (m >= 0) {
      a.prev[2 * i] = a.k2[m].x;
      a.prev[2 * i + 1] = a.k2[m].y;
    }
  }
  if (lane == 0) a.result[0] = nmatches;
}

// Frame::GetFeaturesInArea (src/Frame.cc:765-831) for a batch of queries (x, y, r, minLevel, maxLevel): one wave
// per query walks the cells in the reference's order (ix outer, iy inner, in-cell order).  pass 0 counts,
// pass 1 writes the indices at qOff[q].
__global__ __launch_bounds__(256) void k_area_query(InitArgs a, const float* __restrict__ q, int nq,
                                                    int* __restrict__ qOff, int* __restrict__ out, int pass) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= nq) return;
  const float x = q[5 * qi], y = q[5 * qi + 1], r = q[5 * qi + 2];
  const int minLevel = (int)q[5 * qi + 3], maxLevel = (int)q[5 * qi + 4];
  const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
  int total = 0;
  const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.minX), r), a.invW)));
  const int cx1 = min(63, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.minX), r), a.invW)));
  const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.minY), r), a.invH)));
  const int cy1 = min(47, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.minY), r), a.invH)));
  if (cx0 < 64 && cx1 >= 0 && cy0 < 48 && cy1 >= 0) {
    const int wbase = pass ? qOff[qi] : 0;
    for (int ix = cx0; ix <= cx1; ix++)
      for (int iy = cy0; iy <= cy1; iy++) {
        const int b = a.cellStart[ix * 48 + iy], e = a.cellStart[ix * 48 + iy + 1];
        for (int base = b; base < e; base += 64) {
          const int j = base + lane;
          bool ok = false;
          int i2 = 0;
          if (j < e) {
            i2 = a.cellItems[j];
            const orbx_keypoint k2 = a.k2[i2];
            ok = !(checkLevels && (k2.octave < minLevel || (maxLevel >= 0 && k2.octave > maxLevel))) &&
                 fabsf(__fsub_rn(k2.x, x)) < r && fabsf(__fsub_rn(k2.y, y)) < r;
          }
          const uint64_t m = __ballot(ok);
          if (pass && ok) out[wbase + total + prefix_count(m)] = i2;
          total += __popcll(m);
        }
      }
  }
  if (!pass && lane == 0) qOff[qi] = total;
}

hipError_t launch_grid_build(const InitArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_grid, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_area_query(const InitArgs& a, const float* q, int nq, int* qOff, int* out, int pass, hipStream_t s) {
  if (nq > 0) hipLaunchKernelGGL(k_area_query, dim3((nq + 3) / 4), dim3(256), 0, s, a, q, nq, qOff, out, pass);
  return hipGetLastError();
}
hipError_t launch_scan_offsets(const InitArgs& a, hipStream_t s) {  // exclusive scan of a.candOff[0..n1] (n1 = #queries)
  hipLaunchKernelGGL(k_init_scan, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_search_init(const InitArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_grid, dim3(1), dim3(256), 0, s, a);
  if (a.n1 > 0) {
    hipLaunchKernelGGL(k_init_cands, dim3((a.n1 + 3) / 4), dim3(256), 0, s, a, 0);
    hipLaunchKernelGGL(k_init_scan, dim3(1), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}
// ---- SearchForInitialization: the greedy walk as a parallel fixed-point iteration ------------------------------------------
// vMatchedDistance[i2] seen by keypoint i1 = the distance of the LAST claim on i2 by a keypoint < i1 (claims on one i2
// strictly decrease, :665), so the walk is the unique fixed point of "claim[i1] = best candidate under the gates given
// the claims of all i1' < i1".  Rounds re-evaluate every i1 against the previous round's claims (per i2 the list of
// claimers, at most kFeWriters) until a round changes nothing; overflow or no convergence -> k_init_resolve.
__device__ __forceinline__ int init_matched_dist(const InitArgs& a, int prev, int round_no, int i2, int i1) {
  int lw = -1, ld = 0x7FFFFFFF;
  if (round_no > 0) {
    const int c = min(a.nclaimers[prev][i2], kFeWriters);
    for (int e = 0; e < c; e++) {
      const int2 w = a.claimers[prev][i2 * kFeWriters + e];
      if (w.x < i1 && w.x > lw) {
        lw = w.x;
        ld = w.y;
      }
    }
  }
  return ld;
}

__global__ __launch_bounds__(256) void k_init_round(InitArgs a, int prev, int round_no) {
  const int lane = threadIdx.x & 63;
  const int i1 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i1 >= a.n1) return;
  const int b = a.candOff[i1], e = a.candOff[i1 + 1];
  int2 cl = {-1, 0};
  if (e > b) {
    uint64_t best = ~0ull;  // (dist << 32 | position)
    for (int j = b + lane; j < e; j += 64) {
      const int i2 = a.candIdx[j], d = a.candDist[j];
      if (init_matched_dist(a, prev, round_no, i2, i1) <= d) continue;
      const uint64_t v = ((uint64_t)(uint32_t)d << 32) | (uint32_t)(j - b);
      best = v < best ? v : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t t = __shfl_xor((unsigned long long)best, o);
      best = t < best ? t : best;
    }
    if (best != ~0ull) {
      const int bestDist = (int)(best >> 32), bestPos = (int)(best & 0xFFFFFFFFu);
      int second = 0x7FFFFFFF;
      for (int j = b + lane; j < e; j += 64) {
        if (j - b == bestPos) continue;
        const int i2 = a.candIdx[j], d = a.candDist[j];
        if (init_matched_dist(a, prev, round_no, i2, i1) <= d) continue;
        second = min(second, d);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) second = min(second, __shfl_xor(second, o));
      if (bestDist <= 50 && (float)bestDist < __fmul_rn((float)second, a.nnratio)) {
        cl.x = a.candIdx[b + bestPos];
        cl.y = bestDist;
      }
    }
  }
  if (lane == 0) {
    const int2 o = a.claim[prev][i1];
    if (round_no == 0 || o.x != cl.x || o.y != cl.y) a.flags[0] = 1;
    a.claim[prev ^ 1][i1] = cl;
    if (cl.x >= 0) {
      const int pos = atomicAdd(&a.nclaimers[prev ^ 1][cl.x], 1);
      if (pos < kFeWriters) a.claimers[prev ^ 1][cl.x * kFeWriters + pos] = make_int2(i1, cl.y);
      else a.flags[1] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void k_init_reset(InitArgs a, int which, int first) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n2; i += gridDim.x * 256) {
    a.nclaimers[which][i] = 0;
    if (first) a.matches21[i] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x < 34) {
    if (threadIdx.x == 0) a.flags[0] = 0;
    else if (first) a.flags[threadIdx.x] = 0;
  }
}

__device__ __forceinline__ int init_bin(const InitArgs& a, int i1, int i2) {
  float rot = __fsub_rn(a.k1[i1].angle, a.k2[i2].angle);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
  if (bin == 30) bin = 0;
  return bin;
}

__global__ __launch_bounds__(256) void k_init_owner(InitArgs a, int last) {  // vnMatches21 = the last claimer; votes
  const int i1 = blockIdx.x * 256 + threadIdx.x;
  if (i1 >= a.n1) return;
  const int2 cl = a.claim[last][i1];
  if (cl.x < 0) return;
  atomicMax(&a.matches21[cl.x], i1);
  if (a.checkOri) atomicAdd(&a.flags[4 + init_bin(a, i1, cl.x)], 1);  // stolen matches stay in rotHist (:712-719)
}

__global__ __launch_bounds__(256) void k_init_finish(InitArgs a, int last) {
  int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
  if (a.checkOri) {
    for (int i = 0; i < 30; i++) {
      const int s = a.flags[4 + i];
      if (s > max1) {
        max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i;
      } else if (s > max2) {
        max3 = max2; max2 = s; ind3 = ind2; ind2 = i;
      } else if (s > max3) {
        max3 = s; ind3 = i;
      }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
      ind2 = -1;
      ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
      ind3 = -1;
    }
  }
  const int i1 = blockIdx.x * 256 + threadIdx.x;
  int kept = 0;
  if (i1 < a.n1) {
    const int2 cl = a.claim[last][i1];
    int m = -1;
    if (cl.x >= 0 && a.matches21[cl.x] == i1) {  // not stolen by a later keypoint
      m = cl.x;
      if (a.checkOri) {
        const int bin = init_bin(a, i1, cl.x);
        if (bin != ind1 && bin != ind2 && bin != ind3) m = -1;
      }
    }
    a.matches12[i1] = m;
    if (m >= 0) {
      kept = 1;
      a.prev[2 * i1] = a.k2[m].x;
      a.prev[2 * i1 + 1] = a.k2[m].y;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
  if ((threadIdx.x & 63) == 0 && kept) atomicAdd(&a.flags[2], kept);
}

__global__ void k_init_result(InitArgs a) { a.result[0] = a.flags[2]; }

hipError_t launch_search_init_cands_fill(const InitArgs& a, hipStream_t s) {
  if (a.n1 > 0) hipLaunchKernelGGL(k_init_cands, dim3((a.n1 + 3) / 4), dim3(256), 0, s, a, 1);
  return hipGetLastError();
}
hipError_t launch_search_init_rounds(const InitArgs& a, int first_round, int rounds, hipStream_t s) {
  const int gb = (a.n2 + 255) / 256 > 0 ? (a.n2 + 255) / 256 : 1;
  for (int r = first_round; r < first_round + rounds; r++) {
    const int prev = r & 1;
    hipLaunchKernelGGL(k_init_reset, dim3(gb), dim3(256), 0, s, a, prev ^ 1, r == 0 ? 1 : 0);
    hipLaunchKernelGGL(k_init_round, dim3((a.n1 + 3) / 4), dim3(256), 0, s, a, prev, r);
  }
  return hipGetLastError();
}
hipError_t launch_search_init_finish(const InitArgs& a, int last_round, hipStream_t s) {
  const int last = (last_round & 1) ^ 1;
  hipLaunchKernelGGL(k_init_owner, dim3((a.n1 + 255) / 256), dim3(256), 0, s, a, last);
  hipLaunchKernelGGL(k_init_finish, dim3((a.n1 + 255) / 256), dim3(256), 0, s, a, last);
  hipLaunchKernelGGL(k_init_result, dim3(1), dim3(1), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_search_init_resolve_serial(const InitArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_resolve, dim3(1), dim3(64), (size_t)((a.n1 + 15) & ~15) + 16, s, a);
  return hipGetLastError();
}

hipError_t launch_search_init_fill(const InitArgs& a, hipStream_t s) {
  if (a.n1 > 0) hipLaunchKernelGGL(k_init_cands, dim3((a.n1 + 3) / 4), dim3(256), 0, s, a, 1);
  hipLaunchKernelGGL(k_init_resolve, dim3(1), dim3(64), (size_t)((a.n1 + 15) & ~15) + 16, s, a);
  return hipGetLastError();
}

// ================================================================================================ projection
// ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&) (src/ORBmatcher.cc:41-221), pinhole case.
// Per map point the candidate list (GetFeaturesInArea order, level filter, stereo-consistency filter) and the
// Hamming distances do not depend on the evolving F.mvpMapPoints, so they are produced in parallel (one wave per
// map point); only the occupancy gate + best / second-best + assignment is walked serially in iMP order.
__device__ __forceinline__ bool proj_active(const orbx_map_point_view& mp, const ProjArgs& a, float& radius) {
  if (!mp.in_view) return false;
  if (a.far && mp.track_depth > a.thFar) return false;
  if (mp.bad) return false;
  float r = ((double)mp.view_cos > 0.998) ? 2.5f : 4.0f;  // RadiusByViewingCos, :223-228
  if ((double)a.th != 1.0) r = __fmul_rn(r, a.th);
  radius = __fmul_rn(r, a.scale[mp.predicted_level]);
  return true;
}

// One query per point, common to both SearchByProjection flavours.
struct ProjQuery {
  float x, y, ur, r;
  int minLevel, maxLevel;
  const uint8_t* desc;
};
__device__ __forceinline__ bool proj_query(const ProjArgs& a, int im, ProjQuery& q) {
  if (a.mode == 0) {
    const orbx_map_point_view& mp = a.mps[im];
    if (!proj_active(mp, a, q.r)) return false;
    q.x = mp.proj_x;
    q.y = mp.proj_y;
    q.ur = mp.proj_xr;
    q.minLevel = mp.predicted_level - 1;
    q.maxLevel = mp.predicted_level;
    q.desc = mp.desc;
    return true;
  }
  const orbx_projected_point& p = a.pts[im];
  if (!p.valid) return false;
  q.x = p.u;
  q.y = p.v;
  q.ur = p.ur;
  q.r = p.radius;
  q.minLevel = p.min_level;
  q.maxLevel = p.max_level;
  q.desc = p.desc;
  return true;
}

__global__ __launch_bounds__(256) void k_proj_cands(ProjArgs a, int pass) {
  const int lane = threadIdx.x & 63;
  const int im = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (im >= a.nmp) return;
  const InitArgs& g = a.grid;
  ProjQuery q;
  int total = 0;
  if (proj_query(a, im, q)) {
    const float x = q.x, y = q.y, r = q.r;
    const int minLevel = q.minLevel, maxLevel = q.maxLevel;
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int cx1 = min(63, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH)));
    const int cy1 = min(47, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH)));
    if (cx0 < 64 && cx1 >= 0 && cy0 < 48 && cy1 >= 0) {
      uint32_t d1[8];
#pragma unroll
      for (int i = 0; i < 8; i++) d1[i] = reinterpret_cast<const uint32_t*>(q.desc)[i];
      const int wbase = pass ? a.candOff[im] : 0;
      for (int ix = cx0; ix <= cx1; ix++)
        for (int iy = cy0; iy <= cy1; iy++) {
          const int b = g.cellStart[ix * 48 + iy], e = g.cellStart[ix * 48 + iy + 1];
          for (int base = b; base < e; base += 64) {
            const int j = base + lane;
            bool ok = false;
            int i2 = 0, oct = 0;
            if (j < e) {
              i2 = g.cellItems[j];
              const orbx_keypoint k2 = g.k2[i2];
              oct = k2.octave;
              ok = !(checkLevels && (oct < minLevel || (maxLevel >= 0 && oct > maxLevel))) &&
                   fabsf(__fsub_rn(k2.x, x)) < r && fabsf(__fsub_rn(k2.y, y)) < r;
              if (ok && a.uRight) {  // stereo consistency, :97-100 / :1666-1670
                const float ur = a.uRight[i2];
                if (ur > 0 && fabsf(__fsub_rn(q.ur, ur)) > r) ok = false;
              }
            }
            const uint64_t m = __ballot(ok);
            if (pass && ok) {
              const int o = wbase + total + prefix_count(m);
              if (o < a.candCap) {
                a.candIdx[o] = i2;
                a.candDist[o] = (hamming256(d1, reinterpret_cast<const uint32_t*>(a.desc) + (long long)i2 * 8) << 8) | oct;
              }
            }
            total += __popcll(m);
          }
        }
    }
  }
  if (!pass && lane == 0) a.candOff[im] = total;
}

__global__ __launch_bounds__(64) void k_proj_resolve(ProjArgs a) {
  __shared__ int hist[30];
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int* binIdx = reinterpret_cast<int*>(smem);  // mode 1: (bin << 24 | keypoint index) per accepted match, in order
  const int lane = threadIdx.x;
  for (int i = lane; i < a.grid.n2; i += 64) a.match[i] = -1;
  for (int i = lane; i < 30; i += 64) hist[i] = 0;
  __syncthreads();
  int nmatches = 0, nBin = 0;
  for (int im = 0; im < a.nmp; im++) {
    const int b = a.candOff[im], e = a.candOff[im + 1];
    if (e <= b) continue;
    // two smallest (dist, position) among the candidates whose keypoint is still free == the reference's
    // best / second-best tracking with strict '<' updates
    uint64_t best = ~0ull, second = ~0ull;  // (dist << 40) | (position << 8) | octave
    for (int j = b + lane; j < e; j += 64) {
      if (a.occupied[a.candIdx[j]]) continue;
      const int dv = a.candDist[j];
      const uint64_t v = ((uint64_t)(uint32_t)(dv >> 8) << 40) | ((uint64_t)(uint32_t)(j - b) << 8) | (uint32_t)(dv & 0xFF);
      if (v < best) {
        second = best;
        best = v;
      } else if (v < second) {
        second = v;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t ob = __shfl_xor((unsigned long long)best, o), os = __shfl_xor((unsigned long long)second, o);
      // merge two sorted pairs (best <= second, ob <= os): new best = min, new second = second smallest of the four
      const uint64_t nb = best < ob ? best : ob;
      const uint64_t mx = best < ob ? ob : best;
      const uint64_t ms = second < os ? second : os;
      second = mx < ms ? mx : ms;
      best = nb;
    }
    if (best == ~0ull) continue;
    const int bestDist = (int)(best >> 40), bestPos = (int)((best >> 8) & 0xFFFFFFFFu), bestLevel = (int)(best & 0xFF);
    bool accept = false;
    if (bestDist <= 100) {  // TH_HIGH
      if (a.mode == 0) {
        const int bestDist2 = second == ~0ull ? 256 : (int)(second >> 40);
        const int bestLevel2 = second == ~0ull ? -1 : (int)(second & 0xFF);
        const float lim = __fmul_rn(a.nnratio, (float)bestDist2);
        const bool reject = bestLevel == bestLevel2 && (float)bestDist > lim;
        accept = !reject && (bestLevel != bestLevel2 || (float)bestDist <= lim);
      } else {
        accept = true;
      }
    }
    if (accept) {
      if (lane == 0) {
        const int bestIdx = a.candIdx[b + bestPos];
        a.match[bestIdx] = im;
        a.occupied[bestIdx] = a.mode == 0 ? a.mps[im].has_observations : a.pts[im].has_observations;
        if (a.mode == 1 && a.checkOri) {
          float rot = __fsub_rn(a.pts[im].angle, a.grid.k2[bestIdx].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
          if (bin == 30) bin = 0;
          binIdx[nBin] = (bin << 24) | bestIdx;
          hist[bin]++;
        }
      }
      nBin++;
      nmatches++;
      __threadfence_block();
    }
    __syncthreads();
  }
  if (a.mode == 1 && a.checkOri) {  // rotation-consistency cull, :1780-1800 (+ ComputeThreeMaxima :1920-1955)
    __syncthreads();
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < 30; i++) {
      const int s = hist[i];
      if (s > max1) {
        max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i;
      } else if (s > max2) {
        max3 = max2; max2 = s; ind3 = ind2; ind2 = i;
      } else if (s > max3) {
        max3 = s; ind3 = i;
      }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
      ind2 = -1;
      ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
      ind3 = -1;
    }
    int removed = 0;
    for (int i = lane; i < nBin; i += 64) {
      const int bn = binIdx[i] >> 24, idx = binIdx[i] & 0xFFFFFF;
      if (bn != ind1 && bn != ind2 && bn != ind3) {
        a.match[idx] = -1;  // CurrentFrame.mvpMapPoints[idx] = NULL (even if a later point re-took the slot)
        removed++;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
    nmatches -= removed;
  }
  if (lane == 0) a.result[0] = nmatches;
}

// ---- parallel resolve ------------------------------------------------------------------------------------------------
// The serial walk (point im sees the keypoints claimed by points < im) is the unique fixed point of
//   choice[im] = best candidate among keypoints k with !occupied0[k] and no accepted, observed point im' < im with
//                choice[im'] == k.
// Round r evaluates every point in parallel against the claims of round r - 1 (taker[k] = smallest claiming point
// index).  By induction point t is final after round t + 1, and a round that changes nothing has reached the fixed
// point, which is the serial result; on real inputs a handful of rounds suffice (a claim only matters when two points
// compete for one keypoint).  Wave per point.
__global__ __launch_bounds__(256) void k_proj_round(ProjArgs a, int prev, int round_no) {
  const int lane = threadIdx.x & 63;
  const int im = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (im >= a.nmp) return;
  const int* takerPrev = a.taker[prev];
  int* takerNew = a.taker[prev ^ 1];
  const int b = a.candOff[im], e = a.candOff[im + 1];
  uint64_t best = ~0ull, second = ~0ull;  // (dist << 40) | (position << 8) | octave
  for (int j = b + lane; j < e; j += 64) {
    const int idx = a.candIdx[j];
    if (a.occupied[idx] || (round_no > 0 && takerPrev[idx] < im)) continue;
    const int dv = a.candDist[j];
    const uint64_t v = ((uint64_t)(uint32_t)(dv >> 8) << 40) | ((uint64_t)(uint32_t)(j - b) << 8) | (uint32_t)(dv & 0xFF);
    if (v < best) {
      second = best;
      best = v;
    } else if (v < second) {
      second = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t ob = __shfl_xor((unsigned long long)best, o), os = __shfl_xor((unsigned long long)second, o);
    const uint64_t nb = best < ob ? best : ob;
    const uint64_t mx = best < ob ? ob : best;
    const uint64_t ms = second < os ? second : os;
    second = mx < ms ? mx : ms;
    best = nb;
  }
  int chosen = -1;
  if (best != ~0ull) {
    const int bestDist = (int)(best >> 40), bestPos = (int)((best >> 8) & 0xFFFFFFFFu), bestLevel = (int)(best & 0xFF);
    bool accept = false;
    if (bestDist <= 100) {  // TH_HIGH
      if (a.mode == 0) {
        const int bestDist2 = second == ~0ull ? 256 : (int)(second >> 40);
        const int bestLevel2 = second == ~0ull ? -1 : (int)(second & 0xFF);
        const float lim = __fmul_rn(a.nnratio, (float)bestDist2);
        const bool reject = bestLevel == bestLevel2 && (float)bestDist > lim;
        accept = !reject && (bestLevel != bestLevel2 || (float)bestDist <= lim);
      } else {
        accept = true;
      }
    }
    if (accept) chosen = a.candIdx[b + bestPos];
  }
  if (lane == 0) {
    if (round_no == 0 || a.choice[im] != chosen) a.flags[0] = 1;
    a.choice[im] = chosen;
    if (chosen >= 0 && (a.mode == 0 ? a.mps[im].has_observations : a.pts[im].has_observations))
      atomicMin(&takerNew[chosen], im);
  }
}

__global__ __launch_bounds__(256) void k_proj_reset(ProjArgs a, int which, int first) {  // taker[which] = +inf
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.grid.n2; i += gridDim.x * 256) {
    a.taker[which][i] = 0x7FFFFFFF;
    if (first) a.match[i] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x < 33) {
    if (threadIdx.x == 0) a.flags[0] = 0;                 // "changed in this round"
    else if (first) a.flags[threadIdx.x] = 0;             // accepted, removed, histogram
  }
}

// After convergence: match[k] = the LAST point that chose k (later assignments overwrite), occupied[k] = that point's
// observation flag, orientation histogram of the accepted pairs (mode 1).
__global__ __launch_bounds__(256) void k_proj_assign(ProjArgs a) {
  const int im = blockIdx.x * 256 + threadIdx.x;
  bool acc = false;
  if (im < a.nmp) {
    const int k = a.choice[im];
    if (k >= 0) {
      acc = true;
      atomicMax(&a.match[k], im);
      if (a.mode == 1 && a.checkOri) {
        float rot = __fsub_rn(a.pts[im].angle, a.grid.k2[k].angle);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
        if (bin == 30) bin = 0;
        atomicAdd(&a.flags[3 + bin], 1);
      }
    }
  }
  const uint64_t m = __ballot(acc);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.flags[1], __popcll(m));
}

__global__ __launch_bounds__(256) void k_proj_cull(ProjArgs a) {
  // occupied: set by the last chooser (a keypoint whose first chooser has observations has no later chooser)
  for (int k = blockIdx.x * 256 + threadIdx.x; k < a.grid.n2; k += gridDim.x * 256) {
    const int im = a.match[k];
    if (im >= 0) a.occupied[k] = a.mode == 0 ? a.mps[im].has_observations : a.pts[im].has_observations;
  }
}

__global__ __launch_bounds__(256) void k_proj_cull2(ProjArgs a) {  // rotation-consistency cull (:1780-1800, :1920-1955)
  int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < 30; i++) {
    const int s = a.flags[3 + i];
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s; ind3 = ind2; ind2 = i;
    } else if (s > max3) {
      max3 = s; ind3 = i;
    }
  }
  if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
    ind2 = -1;
    ind3 = -1;
  } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
    ind3 = -1;
  }
  const int im = blockIdx.x * 256 + threadIdx.x;
  bool rem = false;
  if (im < a.nmp) {
    const int k = a.choice[im];
    if (k >= 0) {
      float rot = __fsub_rn(a.pts[im].angle, a.grid.k2[k].angle);
      if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
      int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
      if (bin == 30) bin = 0;
      if (bin != ind1 && bin != ind2 && bin != ind3) {
        a.match[k] = -1;  // CurrentFrame.mvpMapPoints[idx] = NULL, even if a later point re-took the slot
        rem = true;
      }
    }
  }
  const uint64_t m = __ballot(rem);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.flags[2], __popcll(m));
}

__global__ void k_proj_result(ProjArgs a) { a.result[0] = a.flags[1] - a.flags[2]; }

hipError_t launch_proj_cands_fill(const ProjArgs& a, hipStream_t s) {
  if (a.nmp > 0) hipLaunchKernelGGL(k_proj_cands, dim3((a.nmp + 3) / 4), dim3(256), 0, s, a, 1);
  return hipGetLastError();
}
hipError_t launch_proj_rounds(const ProjArgs& a, int first_round, int rounds, hipStream_t s) {
  if (a.nmp <= 0) return hipSuccess;
  const int gb = (a.grid.n2 + 255) / 256;
  for (int r = first_round; r < first_round + rounds; r++) {
    const int prev = r & 1;  // round r reads taker[r & 1] (claims of round r - 1) and writes taker[(r & 1) ^ 1]
    hipLaunchKernelGGL(k_proj_reset, dim3(gb), dim3(256), 0, s, a, prev ^ 1, r == 0 ? 1 : 0);
    hipLaunchKernelGGL(k_proj_round, dim3((a.nmp + 3) / 4), dim3(256), 0, s, a, prev, r);
  }
  return hipGetLastError();
}
hipError_t launch_proj_finish(const ProjArgs& a, int last_round, hipStream_t s) {
  (void)last_round;
  if (a.nmp > 0) {
    hipLaunchKernelGGL(k_proj_assign, dim3((a.nmp + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_proj_cull, dim3((a.grid.n2 + 255) / 256), dim3(256), 0, s, a);
    if (a.mode == 1 && a.checkOri) hipLaunchKernelGGL(k_proj_cull2, dim3((a.nmp + 255) / 256), dim3(256), 0, s, a);
  }
  hipLaunchKernelGGL(k_proj_result, dim3(1), dim3(1), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_proj_resolve_serial(const ProjArgs& a, hipStream_t s) {
  const size_t lds = (a.mode == 1 && a.checkOri) ? (size_t)(a.nmp + 4) * 4 : 16;
  hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(64), lds, s, a);
  return hipGetLastError();
}

// ---- stereo-fisheye resolve (F.Nleft != -1) ------------------------------------------------------------------------------
// best / second-best (dist << 40 | position << 8 | octave) over the still-free candidates of one point, all lanes.
__device__ __forceinline__ void proj_best2(const int* off, const int* idx, const int* dist, const uint8_t* occ, int im, int lane,
                                           uint64_t& best, uint64_t& second, int& b) {
  b = off[im];
  const int e = off[im + 1];
  best = ~0ull;
  second = ~0ull;
  for (int j = b + lane; j < e; j += 64) {
    if (occ[idx[j]]) continue;
    const int dv = dist[j];
    const uint64_t v = ((uint64_t)(uint32_t)(dv >> 8) << 40) | ((uint64_t)(uint32_t)(j - b) << 8) | (uint32_t)(dv & 0xFF);
    if (v < best) {
      second = best;
      best = v;
    } else if (v < second) {
      second = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t ob = __shfl_xor((unsigned long long)best, o), os = __shfl_xor((unsigned long long)second, o);
    const uint64_t nb = best < ob ? best : ob;
    const uint64_t mx = best < ob ? ob : best;
    const uint64_t ms = second < os ? second : os;
    second = mx < ms ? mx : ms;
    best = nb;
  }
}

__global__ __launch_bounds__(64) void k_proj_resolve_fe(ProjFeArgs a) {
  __shared__ int hist[30];
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int* binIdx = reinterpret_cast<int*>(smem);  // mode 1: (bin << 24 | slot) per accepted match
  const int lane = threadIdx.x;
  for (int i = lane; i < a.n; i += 64) a.match[i] = -1;
  for (int i = lane; i < 30; i += 64) hist[i] = 0;
  __syncthreads();
  int nmatches = 0, nBin = 0;
  const uint8_t* occL = a.occupied;
  const uint8_t* occR = a.occupied + a.nLeft;
  for (int im = 0; im < a.nmp; im++) {
    const uint8_t obs = a.mode == 0 ? a.mps[im].has_observations : a.pts[im].has_observations;
    auto assign = [&](int slot) {  // F.mvpMapPoints[slot] = pMP (lane 0 writes; the barrier below publishes it)
      if (lane == 0) {
        a.match[slot] = im;
        a.occupied[slot] = obs;
      }
    };
    auto vote = [&](int slot) {
      if (a.mode == 1 && a.checkOri) {
        if (lane == 0) {
          float rot = __fsub_rn(a.pts[im].angle, a.kps[slot].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
          if (bin == 30) bin = 0;
          binIdx[nBin] = (bin << 24) | slot;
          hist[bin]++;
        }
        nBin++;
      }
    };
    bool skipRight = false;
    // ---- left camera (:60-138 / :1639-1701)
    if (a.offL[im + 1] > a.offL[im]) {
      uint64_t best, second;
      int b;
      proj_best2(a.offL, a.idxL, a.distL, occL, im, lane, best, second, b);
      if (best != ~0ull && (int)(best >> 40) <= 100) {
        const int bestDist = (int)(best >> 40), bestIdx = a.idxL[b + (int)((best >> 8) & 0xFFFFFFFFu)];
        if (a.mode == 0) {
          const int bestLevel = (int)(best & 0xFF);
          const int bestDist2 = second == ~0ull ? 256 : (int)(second >> 40);
          const int bestLevel2 = second == ~0ull ? -1 : (int)(second & 0xFF);
          const float lim = __fmul_rn(a.nnratio, (float)bestDist2);
          if (bestLevel == bestLevel2 && (float)bestDist > lim) {
            skipRight = true;  // `continue`, :120
          } else if (bestLevel != bestLevel2 || (float)bestDist <= lim) {
            assign(bestIdx);
            nmatches++;
            const int partner = a.l2r[bestIdx];
            if (partner != -1) {
              assign(partner + a.nLeft);
              nmatches++;
            }
          }
        } else {
          assign(bestIdx);
          nmatches++;
          vote(bestIdx);
        }
      }
    } else if (a.mode == 1) {
      skipRight = true;  // `if (vIndices2.empty()) continue;`, :1651
    }
    __threadfence_block();
    __syncthreads();
    // ---- right camera (:141-213 / :1703-1775)
    if (!skipRight && a.offR[im + 1] > a.offR[im]) {
      uint64_t best, second;
      int b;
      proj_best2(a.offR, a.idxR, a.distR, occR, im, lane, best, second, b);
      if (best != ~0ull && (int)(best >> 40) <= 100) {
        const int bestDist = (int)(best >> 40), bestIdx = a.idxR[b + (int)((best >> 8) & 0xFFFFFFFFu)];
        bool accept = true;
        if (a.mode == 0) {
          const int bestLevel = (int)(best & 0xFF);
          const int bestDist2 = second == ~0ull ? 256 : (int)(second >> 40);
          const int bestLevel2 = second == ~0ull ? -1 : (int)(second & 0xFF);
          accept = !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(a.nnratio, (float)bestDist2));
        }
        if (accept) {
          if (a.mode == 0) {
            const int partner = a.r2l[bestIdx];
            if (partner != -1) {
              assign(partner);
              nmatches++;
            }
          }
          assign(bestIdx + a.nLeft);
          nmatches++;
          if (a.mode == 1) vote(bestIdx + a.nLeft);
        }
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  if (a.mode == 1 && a.checkOri) {
    __syncthreads();
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < 30; i++) {
      const int s = hist[i];
      if (s > max1) {
        max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i;
      } else if (s > max2) {
        max3 = max2; max2 = s; ind3 = ind2; ind2 = i;
      } else if (s > max3) {
        max3 = s; ind3 = i;
      }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
      ind2 = -1;
      ind3 = -1;
    } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
      ind3 = -1;
    }
    int removed = 0;
    for (int i = lane; i < nBin; i += 64) {
      const int bn = binIdx[i] >> 24, slot = binIdx[i] & 0xFFFFFF;
      if (bn != ind1 && bn != ind2 && bn != ind3) {
        a.match[slot] = -1;
        removed++;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
    nmatches -= removed;
  }
  if (lane == 0) a.result[0] = nmatches;
}

// ---- stereo-fisheye resolve as a parallel fixed-point iteration ---------------------------------------------------------
// Slot occupancy is a last-writer relation here (the stereo-partner assignments overwrite unconditionally):
//   occupied(s, im) = has_observations[last point < im that wrote s], or the initial flag if there is none.
// Round r evaluates every point in parallel against the writes of round r - 1 (every slot keeps the list of points that
// wrote it, at most kFeWriters; an overflow sends the call to the serial walk).  Point t is final after round t + 1 and a
// round that reproduces the previous writes is the serial result, exactly as in k_proj_round.
__device__ __forceinline__ bool fe_occupied(const ProjFeArgs& a, int prev, int round_no, int s, int im) {
  int lw = -1;
  if (round_no > 0) {
    const int c = min(a.nwriters[prev][s], kFeWriters);
    for (int e = 0; e < c; e++) {
      const int w = a.writers[prev][s * kFeWriters + e];
      if (w < im && w > lw) lw = w;
    }
  }
  if (lw < 0) return a.occupied[s] != 0;
  return (a.mode == 0 ? a.mps[lw].has_observations : a.pts[lw].has_observations) != 0;
}

__device__ __forceinline__ void fe_best2(const ProjFeArgs& a, int prev, int round_no, const int* off, const int* idx,
                                         const int* dist, int slot0, int im, int lane, uint64_t& best, uint64_t& second,
                                         int& b) {
  b = off[im];
  const int e = off[im + 1];
  best = ~0ull;
  second = ~0ull;
  for (int j = b + lane; j < e; j += 64) {
    if (fe_occupied(a, prev, round_no, slot0 + idx[j], im)) continue;
    const int dv = dist[j];
    const uint64_t v = ((uint64_t)(uint32_t)(dv >> 8) << 40) | ((uint64_t)(uint32_t)(j - b) << 8) | (uint32_t)(dv & 0xFF);
    if (v < best) {
      second = best;
      best = v;
    } else if (v < second) {
      second = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t ob = __shfl_xor((unsigned long long)best, o), os = __shfl_xor((unsigned long long)second, o);
    const uint64_t nb = best < ob ? best : ob;
    const uint64_t mx = best < ob ? ob : best;
    const uint64_t ms = second < os ? second : os;
    second = mx < ms ? mx : ms;
    best = nb;
  }
}

__global__ __launch_bounds__(256) void k_proj_round_fe(ProjFeArgs a, int prev, int round_no) {
  const int lane = threadIdx.x & 63;
  const int im = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (im >= a.nmp) return;
  int4 w = {-1, -1, -1, -1};
  bool skipRight = false;
  if (a.offL[im + 1] > a.offL[im]) {
    uint64_t best, second;
    int b;
    fe_best2(a, prev, round_no, a.offL, a.idxL, a.distL, 0, im, lane, best, second, b);
    if (best != ~0ull && (int)(best >> 40) <= 100) {
      const int bestDist = (int)(best >> 40), bestIdx = a.idxL[b + (int)((best >> 8) & 0xFFFFFFFFu)];
      if (a.mode == 0) {
        const int bestLevel = (int)(best & 0xFF);
        const int bestDist2 = second == ~0ull ? 256 : (int)(second >> 40);
        const int bestLevel2 = second == ~0ull ? -1 : (int)(second & 0xFF);
        const float lim = __fmul_rn(a.nnratio, (float)bestDist2);
        if (bestLevel == bestLevel2 && (float)bestDist > lim) {
          skipRight = true;
        } else if (bestLevel != bestLevel2 || (float)bestDist <= lim) {
          w.x = bestIdx;
          const int partner = a.l2r[bestIdx];
          if (partner != -1) w.y = partner + a.nLeft;
        }
      } else {
        w.x = bestIdx;
      }
    }
  } else if (a.mode == 1) {
    skipRight = true;
  }
  if (!skipRight && a.offR[im + 1] > a.offR[im]) {
    // the right search of point im sees im's own left-camera writes only through slots it cannot select (a left slot,
    // or the partner of the left best: that right slot now holds im itself, i.e. occupied iff im has observations)
    uint64_t best, second;
    int b;
    const int bb = a.offR[im], ee = a.offR[im + 1];
    best = ~0ull;
    second = ~0ull;
    const bool selfObs = (a.mode == 0 ? a.mps[im].has_observations : a.pts[im].has_observations) != 0;
    b = bb;
    for (int j = bb + lane; j < ee; j += 64) {
      const int s = a.nLeft + a.idxR[j];
      const bool occ = (s == w.y) ? selfObs : fe_occupied(a, prev, round_no, s, im);
      if (occ) continue;
      const int dv = a.distR[j];
      const uint64_t v = ((uint64_t)(uint32_t)(dv >> 8) << 40) | ((uint64_t)(uint32_t)(j - bb) << 8) | (uint32_t)(dv & 0xFF);
      if (v < best) {
        second = best;
        best = v;
      } else if (v < second) {
        second = v;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t ob = __shfl_xor((unsigned long long)best, o), os = __shfl_xor((unsigned long long)second, o);
      const uint64_t nb = best < ob ? best : ob;
      const uint64_t mx = best < ob ? ob : best;
      const uint64_t ms = second < os ? second : os;
      second = mx < ms ? mx : ms;
      best = nb;
    }
    if (best != ~0ull && (int)(best >> 40) <= 100) {
      const int bestDist = (int)(best >> 40), bestIdx = a.idxR[b + (int)((best >> 8) & 0xFFFFFFFFu)];
      bool accept = true;
      if (a.mode == 0) {
        const int bestLevel = (int)(best & 0xFF);
        const int bestDist2 = second == ~0ull ? 256 : (int)(second >> 40);
        const int bestLevel2 = second == ~0ull ? -1 : (int)(second & 0xFF);
        accept = !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(a.nnratio, (float)bestDist2));
      }
      if (accept) {
        w.z = bestIdx + a.nLeft;
        if (a.mode == 0) {
          const int partner = a.r2l[bestIdx];
          if (partner != -1) w.w = partner;
        }
      }
    }
  }
  if (lane == 0) {
    const int4 o = a.writes[prev][im];
    if (round_no == 0 || o.x != w.x || o.y != w.y || o.z != w.z || o.w != w.w) a.flags[0] = 1;
    a.writes[prev ^ 1][im] = w;
    const int ws[4] = {w.x, w.y, w.z, w.w};
    for (int t = 0; t < 4; t++) {
      if (ws[t] < 0) continue;
      bool dup = false;
      for (int u = 0; u < t; u++) dup = dup || ws[u] == ws[t];
      if (dup) continue;
      const int pos = atomicAdd(&a.nwriters[prev ^ 1][ws[t]], 1);
      if (pos < kFeWriters) a.writers[prev ^ 1][ws[t] * kFeWriters + pos] = im;
      else a.flags[1] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void k_proj_reset_fe(ProjFeArgs a, int which, int first) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    a.nwriters[which][i] = 0;
    if (first) a.match[i] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x < 34) {
    if (threadIdx.x == 0) a.flags[0] = 0;
    else if (first) a.flags[threadIdx.x] = 0;
  }
}

__device__ __forceinline__ int fe_bin(const ProjFeArgs& a, int im, int slot) {
  float rot = __fsub_rn(a.pts[im].angle, a.kps[slot].angle);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
  if (bin == 30) bin = 0;
  return bin;
}

__global__ __launch_bounds__(256) void k_proj_assign_fe(ProjFeArgs a, int last) {  // last writer wins every slot
  const int im = blockIdx.x * 256 + threadIdx.x;
  int nw = 0;
  if (im < a.nmp) {
    const int4 w = a.writes[last][im];
    // order of the serial writes of one point: left best, its partner, [right partner], right best -- a later write of
    // the same point to the same slot changes nothing (same point index), so only the count matters
    const int ws[4] = {w.x, w.y, w.w, w.z};
    for (int t = 0; t < 4; t++)
      if (ws[t] >= 0) {
        atomicMax(&a.match[ws[t]], im);
        nw++;
      }
    if (a.mode == 1 && a.checkOri) {
      if (w.x >= 0) atomicAdd(&a.flags[4 + fe_bin(a, im, w.x)], 1);
      if (w.z >= 0) atomicAdd(&a.flags[4 + fe_bin(a, im, w.z)], 1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nw += __shfl_xor(nw, o);
  if ((threadIdx.x & 63) == 0 && nw) atomicAdd(&a.flags[2], nw);
}

__global__ __launch_bounds__(256) void k_proj_occ_fe(ProjFeArgs a) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < a.n; k += gridDim.x * 256) {
    const int im = a.match[k];
    if (im >= 0) a.occupied[k] = a.mode == 0 ? a.mps[im].has_observations : a.pts[im].has_observations;
  }
}

__global__ __launch_bounds__(256) void k_proj_cull_fe(ProjFeArgs a, int last) {
  int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < 30; i++) {
    const int s = a.flags[4 + i];
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s; ind3 = ind2; ind2 = i;
    } else if (s > max3) {
      max3 = s; ind3 = i;
    }
  }
  if ((float)max2 < __fmul_rn(0.1f, (float)max1)) {
    ind2 = -1;
    ind3 = -1;
  } else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) {
    ind3 = -1;
  }
  const int im = blockIdx.x * 256 + threadIdx.x;
  int rem = 0;
  if (im < a.nmp) {
    const int4 w = a.writes[last][im];
    const int ws[2] = {w.x, w.z};
    for (int t = 0; t < 2; t++)
      if (ws[t] >= 0) {
        const int bin = fe_bin(a, im, ws[t]);
        if (bin != ind1 && bin != ind2 && bin != ind3) {
          a.match[ws[t]] = -1;
          rem++;
        }
      }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rem += __shfl_xor(rem, o);
  if ((threadIdx.x & 63) == 0 && rem) atomicAdd(&a.flags[3], rem);
}

__global__ void k_proj_result_fe(ProjFeArgs a) { a.result[0] = a.flags[2] - a.flags[3]; }

hipError_t launch_proj_rounds_fisheye(const ProjFeArgs& a, int first_round, int rounds, hipStream_t s) {
  if (a.nmp <= 0) return hipSuccess;
  const int gb = (a.n + 255) / 256;
  for (int r = first_round; r < first_round + rounds; r++) {
    const int prev = r & 1;
    hipLaunchKernelGGL(k_proj_reset_fe, dim3(gb), dim3(256), 0, s, a, prev ^ 1, r == 0 ? 1 : 0);
    hipLaunchKernelGGL(k_proj_round_fe, dim3((a.nmp + 3) / 4), dim3(256), 0, s, a, prev, r);
  }
  return hipGetLastError();
}
hipError_t launch_proj_finish_fisheye(const ProjFeArgs& a, int last_round, hipStream_t s) {
  const int last = (last_round & 1) ^ 1;  // round r wrote writes[(r & 1) ^ 1]
  hipLaunchKernelGGL(k_proj_assign_fe, dim3((a.nmp + 255) / 256), dim3(256), 0, s, a, last);
  hipLaunchKernelGGL(k_proj_occ_fe, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  if (a.mode == 1 && a.checkOri) hipLaunchKernelGGL(k_proj_cull_fe, dim3((a.nmp + 255) / 256), dim3(256), 0, s, a, last);
  hipLaunchKernelGGL(k_proj_result_fe, dim3(1), dim3(1), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_proj_resolve_fisheye(const ProjFeArgs& a, hipStream_t s) {
  const size_t lds = (a.mode == 1 && a.checkOri) ? (size_t)(2 * a.nmp + 4) * 4 : 16;  // up to two votes per point
  hipLaunchKernelGGL(k_proj_resolve_fe, dim3(1), dim3(64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_proj_count(const ProjArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_grid, dim3(1), dim3(256), 0, s, a.grid);
  if (a.nmp > 0) {
    hipLaunchKernelGGL(k_proj_cands, dim3((a.nmp + 3) / 4), dim3(256), 0, s, a, 0);
    InitArgs sc = a.grid;  // k_init_scan scans candOff[0 .. n1]
    sc.candOff = a.candOff;
    sc.n1 = a.nmp;
    sc.candCap = 1 << 30;
    hipLaunchKernelGGL(k_init_scan, dim3(1), dim3(256), 0, s, sc);
  }
  return hipGetLastError();
}
hipError_t launch_proj_fill(const ProjArgs& a, hipStream_t s) {
  if (a.nmp > 0) hipLaunchKernelGGL(k_proj_cands, dim3((a.nmp + 3) / 4), dim3(256), 0, s, a, 1);
  const size_t lds = (a.mode == 1 && a.checkOri) ? (size_t)(a.nmp + 4) * 4 : 16;  // one int per accepted match
  hipLaunchKernelGGL(k_proj_resolve, dim3(1), dim3(64), lds, s, a);
  return hipGetLastError();
}

hipError_t prepare_kernels(const Geom& g) {
  const size_t lds_oct = octree_lds_bytes(g);
  const size_t lds_det = (size_t)g.tileP * g.tileH + (size_t)g.scoreP * g.scoreH + 2 * (kSurvCap + kCornerCap) + 16;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_octree),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_oct);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_detect), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds_det);
}

// Host-callable check of the introsort replica (tests compare with std::sort).
void debug_introsort_host(uint64_t* v, int n) { introsort<uint64_t, KeyLess>(v, n, KeyLess()); }

}  // namespace orbx
