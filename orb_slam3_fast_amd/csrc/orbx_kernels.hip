// orbx_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) of the ORB front-end: the extraction pipeline.
//
// One kernel per stage of ORBextractor::operator() (src/ORBextractor.cc:1015-1106 of the reference); every
// kernel covers all images of the batch (and all pyramid levels where the stage allows) in one launch:
//   k_resize    ComputePyramid               :1108-1145  (cv::resize INTER_LINEAR 8U)
//   k_detect    per-cell cv::FAST + NMS + ini/min threshold fallback :892-971, one wave per cell
//   k_octree    DistributeOctTree            :557-757    one workgroup per (image, level), node tables in LDS
//   k_blur      GaussianBlur 7x7 sigma 2     :1074-1076  
//   k_slots     mono/lapping slot assignment :1062-1099  (serial-order semantics)
//   k_describe  IC_Angle + computeOrbDescriptor + output :75-147, one wave per keypoint, ballot-packed bits
// (the association kernels live in orbx_stereo.hip, the grid-guided matchers in orbx_guided.hip, pre-processing in
// orbx_preproc.hip, bag of words in orbx_bow.hip)
//
// Integer/bitwise work; the one contraction on the path -- k_describe's separable 7x7 Gaussian, two banded integer GEMMs per
// keypoint window -- runs on v_mfma_i32_16x16x64_i8 (orbx_blur_mfma.h, exact in i32).  Float steps that decide bits (fastAtan2, the rotated sampling
// coordinates, sub-pixel disparity) use IEEE ops without contraction (-ffp-contract=off for this TU) and
// round-half-even conversions, mirroring the x86-64 baseline (no FMA) build of the reference.
#include <type_traits>
#include <map>
#include <mutex>

#include "orbx_device.h"
#include "orbx_introsort.h"
#include "orbx_sincos.h"
#include "orbx_blur_mfma.h"

namespace orbx {

__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

// ================================================================================================ resize
// cv::resize INTER_LINEAR 8U, level l from level l-1.  Coefficient tables (sx, a0/a1 ; sy, b0/b1)
// are built on the host exactly as OpenCV builds them and uploaded once per image size.
// Block = 256 dst columns x RS_DR dst rows.  The source footprint is staged in LDS with coalesced dword loads;
// the horizontal pass (one thread per dst column, walking the footprint rows) leaves (S0*a0 + S1*a1) >> 4 as u16
// in LDS; the vertical pass emits 4 pixels per thread with one u32 store.
#define RS_DW 256
#define RS_DR 16
#define RS_HROWS 6  // footprint rows per wave in the unrolled part of the horizontal pass (4 waves: 24 rows)
// NDW: compile-time footprint row pitch in dwords (0 = run time).  With a constant pitch a wave's six footprint rows are
// one LDS address per output column plus ds_read2 immediates.
// xtab: one uint4 per dst column {sx, sx & ~3, v_perm selector, a0 | a1 << 16}, each level padded to a multiple of
// RS_DW entries with copies of its last column (a block reads its 256 entries unconditionally).
// The level pair of one launch, by value: the kernel's first scalar loads are its only kernel-argument loads (indexing Geom::lv with
// the run-time level cost a second, dependent trip and ~40 scalar instructions of address arithmetic per wave).
struct ResizeLv {
  struct { int w, h, pitch, xcoef, ycoef; } D;
  struct { int w, h; } S;
  int sp, l;                      // source row pitch; destination level (1 = the source is the caller's level 0)
  int nbx, xcdRun;                // tiles per row of tiles; run length of the XCD-aware tile order (<= 1: plain order)
  const uint8_t* src; long long srcImg;   // level l-1 of image 0, bytes between images
  uint8_t* dst; long long dstImg;         // level l of image 0
};
// Vertical blend of cv::resize for four pixels: (((b0 * u0) >> 16) + ((b1 * u1) >> 16) + 2) >> 2 with u0 / u1 = the u16 halves of
// t0 / t1 (horizontal results >> 4, < 2^15) and bb = b0 | b1 << 16 (wave-uniform, an SGPR).  Hand-assembled (the compiler spent
// 24 instructions per four pixels on it: unpacking shifts, per-pixel shift-and-insert with materialised constants): the products
// are v_mad_u32_u16 on the halves picked by op_sel, the rounding 2 rides in the first product as 2 << 16 (exact: it is added below
// the truncation), the two high words are summed and laid down as u16 pairs by SDWA selects (sum <= 1022), a packed shift by 2 and
// one v_perm collect the four bytes: 15 instructions.
__device__ __forceinline__ uint32_t vblend4(uint2 t0, uint2 t1, uint32_t bb, uint32_t kRound) {
  uint32_t p0, p1, p2, p3, q0, q1, q2, q3, w01, w23;
  asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(p0) : "v"(t0.x), "s"(bb), "v"(kRound));
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(p1) : "v"(t0.x), "s"(bb), "v"(kRound));
  asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(p2) : "v"(t0.y), "s"(bb), "v"(kRound));
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(p3) : "v"(t0.y), "s"(bb), "v"(kRound));
  asm("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[0,1,0,0]" : "=v"(q0) : "v"(t1.x), "s"(bb));
  asm("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[1,1,0,0]" : "=v"(q1) : "v"(t1.x), "s"(bb));
  asm("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[0,1,0,0]" : "=v"(q2) : "v"(t1.y), "s"(bb));
  asm("v_mad_u32_u16 %0, %1, %2, 0 op_sel:[1,1,0,0]" : "=v"(q3) : "v"(t1.y), "s"(bb));
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(w01) : "v"(p0), "v"(q0));
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(w01) : "v"(p1), "v"(q1));
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(w23) : "v"(p2), "v"(q2));
  asm("v_add_u32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(w23) : "v"(p3), "v"(q3));
  asm("v_pk_lshrrev_b16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(w01) : "v"(w01));
  asm("v_pk_lshrrev_b16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(w23) : "v"(w23));
  return __builtin_amdgcn_perm(w23, w01, 0x06040200u);
}
template <int NDW>
__global__ __launch_bounds__(256) void k_resize(ResizeLv rl, const uint4* __restrict__ xtab,
                                                const uint32_t* __restrict__ yrow, const short* __restrict__ yab,
                                                int srcRowsMax, int srcDwMaxRt) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int SDW = NDW ? NDW : srcDwMaxRt;
  const auto D = rl.D;
  const auto S = rl.S;
  const int l = rl.l;
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  // Tile order: the grid is flat over (tile row, tile column) and runs of xcdRun consecutive tiles -- x-adjacent first -- share
  // an XCD, i.e. an L2: the 307-byte source-row segments of x-adjacent blocks meet inside 128-byte lines, and in the plain
  // (x, y, z) grid order those neighbours sat on different XCDs and each fetched the shared line from HBM (k_resize fetched
  // 1.74 x its algorithmic read bytes, profiles/r4e_pmc_traffic.json; k_detect has had the same remap since round 2).
  const int tIdx = xcd_run_remap_rt((int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z, rl.xcdRun);
  const int tby = __builtin_amdgcn_readfirstlane((int)(((float)tIdx + 0.5f) * __builtin_amdgcn_rcpf((float)rl.nbx)));
  const int tbx = tIdx - tby * rl.nbx;
  const int x0 = tbx * RS_DW, y0 = tby * RS_DR;
  const int x1 = min(x0 + RS_DW, D.w) - 1, y1 = min(y0 + RS_DR, D.h) - 1;  // last dst column / row of the block
  const int sp = rl.sp;
  const uint8_t* src = rl.src + (long long)img * rl.srcImg;
  uint32_t* st = reinterpret_cast<uint32_t*>(smem);                        // [srcRowsMax][SDW] dwords
  uint8_t* ht8 = smem + (size_t)srcRowsMax * SDW * 4;                      // [srcRowsMax][RS_DW] u16
#ifdef RS_PROF
  long long tq0 = wall_clock64();
#endif
  // yrow[dy] = the two source rows of destination row dy, already clamped to the level (low / high half; built on the host):
  // no s_max / s_min chains per row on the scalar ALU, this kernel's busiest pipe (round 5)
  const int rb = (int)(yrow[D.ycoef + y0] & 0xFFFFu);                           // first source row needed
  const int re = (int)(yrow[D.ycoef + y1] >> 16);                               // last source row needed
  const int nrows = re - rb + 1;
  const int cb = (int)xtab[D.xcoef + x0].y;                                     // first source byte (dword aligned)
  const int ce = min((int)xtab[D.xcoef + x1].x + 1, S.w - 1);
  const int ndw = ((ce - cb) >> 2) + 1;
  // Every coefficient the two passes need is fetched HERE, together with the footprint: a block is a chain of dependent
  // memory latencies (tile bounds -> footprint -> column coefficients -> row coefficients), and the kernel's time is that
  // chain times the number of block generations, not bandwidth or issue slots.
  const int qc = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t sel[4], coef[4];
  int ba[4];  // byte offset, inside a footprint row, of the aligned dword pair that holds S[sx], S[sx + 1]
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint4 e = xtab[D.xcoef + x0 + 4 * qc + j];  // S[sx + 1] is only weighted by a1 != 0 when it exists (build_coefs)
    ba[j] = (int)e.y - cb;
    sel[j] = e.z;
    coef[j] = e.w;
  }
  uint32_t vsy[RS_DR / 4];
  uint32_t vbb[RS_DR / 4];
#pragma unroll
  for (int k = 0; k < RS_DR / 4; k++) {  // rows w, w + 4, ... of the block belong to this wave (wave-uniform: scalar loads)
    const int dy = min(y0 + w + 4 * k, D.h - 1);
    vsy[k] = yrow[D.ycoef + dy];
    vbb[k] = reinterpret_cast<const uint32_t*>(yab)[D.ycoef + dy];
  }
  {
    // Footprint -> LDS.  Thread = (row phase r0, dword column c): rpp = 256 / ndw source rows per pass, so a trip is an
    // offset increment and a load (no per-item index division); all of a thread's global loads are issued before its
    // first LDS store (a plain copy loop serialised one HBM latency per trip).  Rows past the footprint are clamped to its
    // last row on both sides (the same bytes land on the same LDS dword again): no predicates in the loop.
    // Only a level-0 source (the caller's buffer) may end with its last pixel: then the last dword is read byte-wise.
    const bool whole = l > 1 || cb + 4 * ndw <= S.w;
    constexpr int kFastTrips = 8;
    if (NDW != 0 && whole && ndw > 64 && ndw <= 85 && nrows <= 3 * kFastTrips && rb + 3 * kFastTrips + 4 <= S.h) {
      // The usual block (1.2 pyramid; not the level's last rows or a narrow last column block): 65 .. 85 dwords per footprint row,
      // i.e. three rows per trip of the 256 threads, so the trips are SCALAR base increments of the loads and immediate offsets of
      // the LDS stores -- no per-trip address arithmetic, no clamps, and straight-line code (inside the general loops below the
      // compiler drains the coefficient loads before it issues the first footprint load).  Rows past the footprint (up to row
      // 3 + 21 + 3: threads beyond 3 * ndw duplicate the next trip's rows) are real image rows (tested above) and land behind the
      // footprint area, in the horizontal pass's output buffer, which is only written after the barrier.
      const float inv_nd = __builtin_amdgcn_rcpf((float)ndw);
      const int r0 = (int)(((float)tid + 0.5f) * inv_nd);      // tid / ndw (0 .. 3)
      const int c = tid - __mul24(r0, ndw);
      const uint8_t* fb = src + (long long)rb * sp + cb;       // wave-uniform base, 32-bit lane offset
      const uint32_t off0 = (uint32_t)(__mul24(r0, sp) + 4 * c);
      uint8_t* sdst = smem + 4 * (__mul24(r0, NDW) + c);
      uint32_t v[kFastTrips];
      const uint8_t* fbk[kFastTrips];
#pragma unroll
      for (int k = 0; k < kFastTrips; k++) fbk[k] = fb + (size_t)(3 * k) * (size_t)sp;   // scalar
      static_assert(kFastTrips == 8, "the asm below issues eight loads");
      // ONE asm statement: the eight loads and the wait for them.  The compiler does not know that the destination registers are
      // written asynchronously; as separate statements (round 4) nothing kept it from copying or spilling a v[k] between the
      // load and the wait.  Early-clobber outputs: no destination may share a register with the lane offset.
      asm volatile(
          "global_load_dword %0, %8, %9\n\t"
          "global_load_dword %1, %8, %10\n\t"
          "global_load_dword %2, %8, %11\n\t"
          "global_load_dword %3, %8, %12\n\t"
          "global_load_dword %4, %8, %13\n\t"
          "global_load_dword %5, %8, %14\n\t"
          "global_load_dword %6, %8, %15\n\t"
          "global_load_dword %7, %8, %16\n\t"
          "s_waitcnt vmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
          : "v"(off0), "s"(fbk[0]), "s"(fbk[1]), "s"(fbk[2]), "s"(fbk[3]), "s"(fbk[4]), "s"(fbk[5]), "s"(fbk[6]), "s"(fbk[7])
          : "memory");
#pragma unroll
      for (int k = 0; k < kFastTrips; k++) *reinterpret_cast<uint32_t*>(sdst + k * (12 * NDW)) = v[k];
    } else
    for (int cbase = 0; cbase < ndw; cbase += 256) {          // one trip unless the scale factor exceeds ~3.9
    const int nd = min(ndw - cbase, 256);
    // (v_rcp_f32 is good to 1 ulp; the quotients below stay >= 0.5 / nd away from the next integer)
    const float inv_nd = __builtin_amdgcn_rcpf((float)nd);
    const int rpp = __builtin_amdgcn_readfirstlane((int)(256.5f * inv_nd));  // rows per pass (3 at scale 1.2: 78 dwords per row)
    const int r0 = (int)(((float)tid + 0.5f) * inv_nd);                      // tid / nd, once per thread
    const int c = cbase + tid - r0 * nd;
    constexpr int kTrips = 8;  // one batch covers the footprint at scale 1.2 (21 rows / 3 per pass); larger scales loop
    if (whole) {
      const uint8_t* fb = src + (long long)rb * sp + cb;       // wave-uniform base, 32-bit lane offsets
      const uint32_t offLast = (uint32_t)(__mul24(nrows - 1, sp) + 4 * c), offStep = (uint32_t)(rpp * sp);
      const int idxLast = 4 * (__mul24(nrows - 1, SDW) + c), idxStep = 4 * rpp * SDW;
      for (int rbase = 0; rbase < nrows; rbase += rpp * kTrips) {
        const uint32_t off0 = (uint32_t)(__mul24(rbase + r0, sp) + 4 * c);
        const int idx0 = 4 * (__mul24(rbase + r0, SDW) + c);
        uint32_t v[kTrips];
#pragma unroll
        for (int k = 0; k < kTrips; k++) v[k] = *reinterpret_cast<const uint32_t*>(fb + min(off0 + (uint32_t)k * offStep, offLast));
#pragma unroll
        for (int k = 0; k < kTrips; k++) *reinterpret_cast<uint32_t*>(smem + min(idx0 + k * idxStep, idxLast)) = v[k];
      }
    } else {
      const bool lanes = r0 < rpp;                               // threads beyond rpp * nd idle
      const int gx = cb + 4 * c;
      const bool wide = gx + 4 <= S.w;
      for (int rbase = 0; rbase < nrows; rbase += rpp * kTrips) {
        uint32_t v[kTrips];
        const uint8_t* q = src + (long long)(rb + rbase + r0) * sp + gx;
#pragma unroll
        for (int k = 0; k < kTrips; k++) {
          v[k] = 0;
          if (lanes && rbase + r0 + k * rpp < nrows) {
            if (wide) {
              v[k] = *reinterpret_cast<const uint32_t*>(q);
            } else {
              for (int bI = 0; bI < 4; bI++)
                if (gx + bI < S.w) v[k] |= (uint32_t)q[bI] << (8 * bI);
            }
          }
          q += (long long)rpp * sp;
        }
#pragma unroll
        for (int k = 0; k < kTrips; k++)
          if (lanes && rbase + r0 + k * rpp < nrows) st[(rbase + r0 + k * rpp) * SDW + c] = v[k];
      }
    }
    }
  }
#ifdef RS_PROF
  long long tq1 = wall_clock64();
#endif
  __syncthreads();
#ifdef RS_PROF
  long long tq2 = wall_clock64();
#endif
  {  // horizontal pass: lane = quad of 4 dst columns; wave w takes footprint rows 6w .. 6w+5 (consecutive rows: their LDS
     // offsets fit the ds_read2 immediates), rows from 24 on (scale factors above 1.3) go round-robin.  Per output one
     // ds_read2_b32 (the aligned dword pair holding S[sx], S[sx+1]), one v_perm with a per-lane selector that spreads
     // the two bytes into u16 halves, one v_dot2_u32_u16 against (a0, a1), one shift; four results leave as one b64.
    auto hrow = [&](const uint8_t* rowp, uint8_t* outp) {
      uint32_t t[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t* pr = reinterpret_cast<const uint32_t*>(rowp + ba[j]);
        t[j] = udot2_u16(__builtin_amdgcn_perm(pr[1], pr[0], sel[j]), coef[j], 0u);
      }
      uint2 pk;  // (t >> 4) as u16 pairs: the even result by a plain shift (t < 2^19: the high half comes out zero), the odd
      pk.x = t[0] >> 4;  // one shifted INTO the high half by an SDWA destination select (2 instead of 3 instructions per pair)
      pk.y = t[2] >> 4;
      asm("v_lshrrev_b32_sdwa %0, 4, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(pk.x) : "v"(t[1]));
      asm("v_lshrrev_b32_sdwa %0, 4, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(pk.y) : "v"(t[3]));
      *reinterpret_cast<uint2*>(outp) = pk;
    };
    const int rw0 = RS_HROWS * w;
    const uint8_t* rowp = smem + rw0 * SDW * 4;
    uint8_t* outp = ht8 + rw0 * (RS_DW * 2) + 8 * qc;
#pragma unroll
    for (int i = 0; i < RS_HROWS; i++) {
      if (rw0 + i >= nrows) break;
      hrow(rowp + i * SDW * 4, outp + i * (RS_DW * 2));
    }
    for (int r = 4 * RS_HROWS + w; r < nrows; r += 4) hrow(smem + r * SDW * 4, ht8 + r * (RS_DW * 2) + 8 * qc);
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
    uint32_t kRound = 0x20000u;                // the "+ 2" of the vertical blend, added below the first product's truncation
    asm volatile("" : "+v"(kRound));           // (a VGPR: the products' one scalar operand is the row's coefficient pair)
    if (dx < D.w) {                            // lane-invariant over the rows: tested once, the row loop only has uniform exits
      const uint8_t* htq = ht8 + 8 * qx;
      // wave-uniform: the stores take row base + 32-bit lane offset; the row base advances by four rows per trip
      uint8_t* dstRow = rl.dst + (long long)img * rl.dstImg + (long long)(y0 + w) * D.pitch;
      const long long dstStep = 4ll * D.pitch;
#pragma unroll
      for (int k = 0; k < RS_DR / 4; k++) {
        const int dy = y0 + w + 4 * k;
        if (dy >= D.h) break;
        const uint32_t bb = vbb[k];
        const int r0 = (int)(vsy[k] & 0xFFFFu) - rb, r1 = (int)(vsy[k] >> 16) - rb;
        const uint2 t0 = *reinterpret_cast<const uint2*>(htq + r0 * (RS_DW * 2));
        const uint2 t1 = *reinterpret_cast<const uint2*>(htq + r1 * (RS_DW * 2));
        const uint32_t outw = vblend4(t0, t1, bb, kRound);
        // scalar base + 32-bit lane offset (left to itself the compiler folds dx into the pointer and multiplies dy * pitch
        // per lane in 64 bits)
        asm volatile("global_store_dword %0, %1, %2" : : "v"((uint32_t)dx), "v"(outw), "s"(dstRow) : "memory");
        dstRow += dstStep;
      }
    }
  }
#ifdef RS_PROF
  if (tid == 0 && blockIdx.z == 7 && tbx == 1 && (tby % 9) == 3)
    printf("L%d by %d: load %d wait %d horiz %d wait %d vert %d (x10ns)\n", l, tby, (int)(tq1 - tq0), (int)(tq2 - tq1),
           (int)(tq3 - tq2), (int)(tq4 - tq3), (int)(wall_clock64() - tq4));
#endif
}

// footprint bounds of a 256 x 16 dst block for this level's scale (+ slack for the floor/ceil of the taps)
static void resize_footprint(const Geom& g, int level, int& srcRowsMax, int& srcDwMax, size_t& lds) {
  const LevelDev& D = g.lv[level];
  const LevelDev& S = g.lv[level - 1];
  srcRowsMax = (int)((double)RS_DR * S.h / D.h) + 4;
  srcDwMax = ((int)((double)RS_DW * S.w / D.w) + 12) / 4 + 1;
  lds = (size_t)srcRowsMax * srcDwMax * 4 + (size_t)srcRowsMax * RS_DW * 2;
}
size_t resize_lds_bytes(const Geom& g) {
  size_t m = 0;
  for (int l = 1; l < g.nlevels; l++) {
    int a, b;
    size_t n;
    resize_footprint(g, l, a, b, n);
    m = n > m ? n : m;
  }
  return m;
}

constexpr int kResizeNdw = 80;  // footprint pitch of the compile-time instantiation (every level of a 1.2 pyramid)
hipError_t launch_resize(const Geom& g, const Pyr& p, int nimg, int level, const uint4* xtab, const uint32_t* yofs,
                         const short* yab, hipStream_t s) {
  const LevelDev& D = g.lv[level];
  int srcRowsMax, srcDwMax;
  size_t lds;
  resize_footprint(g, level, srcRowsMax, srcDwMax, lds);
  const int nbx = (D.w + RS_DW - 1) / RS_DW, nby = (D.h + RS_DR - 1) / RS_DR;
  dim3 grid(nbx * nby, 1, nimg);
  const LevelDev& S = g.lv[level - 1];
  static const int xcdRun = getenv("ORBX_RESIZE_XCD_RUN") ? atoi(getenv("ORBX_RESIZE_XCD_RUN")) : 8;   // A/B aid (1 = plain order)
  ResizeLv rl;
  rl.nbx = nbx;
  rl.xcdRun = xcdRun;
  rl.D.w = D.w; rl.D.h = D.h; rl.D.pitch = D.pitch; rl.D.xcoef = D.xcoef; rl.D.ycoef = D.ycoef;
  rl.S.w = S.w; rl.S.h = S.h;
  rl.l = level;
  if (level == 1) {
    rl.sp = (int)p.l0Row; rl.src = p.l0; rl.srcImg = p.l0Img;
  } else {
    rl.sp = S.pitch; rl.src = p.pyr + S.off; rl.srcImg = g.pyrImg;
  }
  rl.dst = p.pyr + D.off; rl.dstImg = g.pyrImg;
  if (srcDwMax == kResizeNdw)
    hipLaunchKernelGGL(k_resize<kResizeNdw>, grid, dim3(256), lds, s, rl, xtab, yofs, yab, srcRowsMax, srcDwMax);
  else
    hipLaunchKernelGGL(k_resize<0>, grid, dim3(256), lds, s, rl, xtab, yofs, yab, srcRowsMax, srcDwMax);
  return hipGetLastError();
}


// cv::resize(INTER_LINEAR) of whole single-channel frames -- System::TrackStereo's input resize (src/System.cc:297-298,
// settings_->needToResize()) -- with the pyramid's kernel: the arithmetic is the same cv::resize, so the pre-processing
// plans build the level tables for the pair (source size, output size) and launch k_resize on it (round 6; the per-pixel
// k_resize_generic ran 64 x 752x480 -> 600x350 in 73 us).  A two-level geometry by value: level 0 = the caller's frames, level 1 =
// the plan's output buffer.
static Geom resize_plain_geom(int sw, int sh, int dw, int dh, long long dp, long long dip) {
  Geom g{};
  g.nlevels = 2;
  g.lv[0].w = sw; g.lv[0].h = sh;
  g.lv[1].w = dw; g.lv[1].h = dh; g.lv[1].pitch = (int)dp;
  g.lv[1].xcoef = 0; g.lv[1].ycoef = 0; g.lv[1].off = 0;
  g.pyrImg = dip;
  return g;
}
size_t resize_plain_lds(int sw, int sh, int dw, int dh) { return resize_lds_bytes(resize_plain_geom(sw, sh, dw, dh, 0, 0)); }
hipError_t prepare_resize_plain(int sw, int sh, int dw, int dh) {
  const size_t lds = std::max<size_t>(resize_plain_lds(sw, sh, dw, dh), 1024);
  hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(k_resize<0>), lds);
  if (e == hipSuccess) e = raise_dynamic_lds(reinterpret_cast<const void*>(k_resize<kResizeNdw>), lds);
  return e;
}
hipError_t launch_resize_plain(const uint8_t* src, int sw, int sh, long long sp, long long sip, uint8_t* dst, int dw, int dh,
                               long long dp, long long dip, const uint4* xtab, const uint32_t* yrow, const short* yab, int nimg,
                               hipStream_t s) {
  const Geom g = resize_plain_geom(sw, sh, dw, dh, dp, dip);
  Pyr p{};
  p.l0 = src; p.l0Row = sp; p.l0Img = sip; p.pyr = dst;
  return launch_resize(g, p, nimg, 1, xtab, yrow, yab, s);
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

// __introsort_loop of a segment of <= 64 elements carried through ALL of its partitions in registers by one wave: lane i
// holds element first + i, a partition is ballots + six cross-lane moves (the stopper lists of partition_wave become "lane k
// receives the k-th stopper" by ds_permute with the stopper's rank as destination; lane 63 never is a destination -- at most 63
// stoppers exist, the pivot slot is excluded -- and takes the pushes of the non-stoppers), the sub-segments are walked depth
// first with a three-entry stack of uniform values.  No LDS traffic, no barrier: the 128-node sort of a level-0 quadtree took
// 10 us as five workgroup-wide rounds of LDS partitions (profiles/r5a_octree_sections.txt).
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}
__device__ void introsort_seg_reg(uint64_t* a, int first, int last, int depth0, int lane) {
  const KeyLess less;
  const int n = last - first;  // 17 .. 64 (uniform)
  uint64_t v = a[first + min(lane, n - 1)];
  uint32_t stk[3];  // f | l << 8 | depth << 16 of the pending right-hand segments (> 16 elements each: at most three exist)
  int sp = 0;
  int f = 0, l = n, d = depth0;
  for (;;) {
    while (l - f > 16) {
      if (d == 0) {  // depth budget exhausted (adversarial inputs only): __partial_sort on this range, serial, through LDS
        if (lane < n) a[first + lane] = v;
        wsync();
        if (lane == 0) is_heapsort<uint64_t, KeyLess>(a, first + f, first + l, less);
        wsync();
        v = a[first + min(lane, n - 1)];
        break;
      }
      --d;
      const int mid = f + (l - f) / 2;
      const uint64_t va = readlane_u64(v, f + 1), vb = readlane_u64(v, mid), vc = readlane_u64(v, l - 1);
      int sel;
      if (less(va, vb)) sel = less(vb, vc) ? mid : (less(va, vc) ? l - 1 : f + 1);
      else if (less(va, vc)) sel = f + 1;
      else sel = less(vb, vc) ? l - 1 : mid;
      const uint64_t vf = readlane_u64(v, f), pivot = readlane_u64(v, sel);
      if (lane == f) v = pivot;
      if (lane == sel) v = vf;
      const bool in = lane > f && lane < l;
      const bool stL = in && !less(v, pivot), stR = in && !less(pivot, v);
      const uint64_t mL = __ballot(stL), mR = __ballot(stR);
      const int nL = __popcll(mL), nR = __popcll(mR), nmin = min(nL, nR);
      const int rankL = __popcll(mL & lanemask_lt()), rankR = __popcll(mR & ~(lanemask_lt() | (1ull << lane)));
      // lane k <- k-th left stopper (ascending) / k-th right stopper (descending)
      const int LiK = __builtin_amdgcn_ds_permute(4 * (stL ? rankL : 63), lane);
      const int RiK = __builtin_amdgcn_ds_permute(4 * (stR ? rankR : 63), lane);
      const int K = __popcll(__ballot(lane < nmin && LiK < RiK));  // (monotone: true for k < K)
      const int INF = 1 << 30;
      const int lk = K < nL ? __builtin_amdgcn_readlane(LiK, K) : INF;
      const int rk = K > 0 ? __builtin_amdgcn_readlane(RiK, K - 1) : INF;
      const int cut = min(lk, rk);
      // pairs k < K swap: a left stopper of rank k takes the element of Ri[k] and vice versa (no lane is both: see DESIGN)
      const int pL = __builtin_amdgcn_ds_bpermute(4 * rankL, RiK), pR = __builtin_amdgcn_ds_bpermute(4 * rankR, LiK);
      const int src = (stL && rankL < K) ? pL : ((stR && rankR < K) ? pR : lane);
      const uint32_t nlo = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * src, (int)(uint32_t)v);
      const uint32_t nhi = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * src, (int)(uint32_t)(v >> 32));
      v = ((uint64_t)nhi << 32) | nlo;
      if (l - cut > 16) {
        const uint32_t e = (uint32_t)cut | ((uint32_t)l << 8) | ((uint32_t)d << 16);
        if (sp == 0) stk[0] = e; else if (sp == 1) stk[1] = e; else stk[2] = e;
        ++sp;
      }
      l = cut;
    }
    if (sp == 0) break;
    --sp;
    const uint32_t e = sp == 0 ? stk[0] : (sp == 1 ? stk[1] : stk[2]);
    f = (int)(e & 0xFF);
    l = (int)((e >> 8) & 0xFF);
    d = (int)(e >> 16);
  }
  if (lane < n) a[first + lane] = v;
}

// One __unguarded_partition_pivot of a segment of 65 .. 128 elements with the segment in REGISTERS (two elements per lane: lane i holds
// first + i and first + 64 + i), the twin of partition_wave for the sizes introsort_seg_reg cannot take.  partition_wave builds the two
// stopper lists in LDS, counts the crossing point K over them and swaps pairs through them -- six dependent LDS round trips and three
// wave syncs, 1.5 - 1.8 us for the 128 expandable nodes of a level-0 quadtree.  Here nothing is listed: with L ascending and R
// descending, "L[k] < R[k]" is monotone in k, so a left stopper at index i with rank r (left stoppers below it) swaps iff MORE than r
// right stoppers lie above i, a right stopper at j with rank s (right stoppers above it) iff more than s left stoppers lie below j --
// prefix popcounts of four ballots.  The partners meet through a mailbox (the swapping left stopper of rank r posts to box[r] and
// collects box[half + r], its partner the other way round; K <= (n - 1) / 2 pairs, so both boxes fit the segment's own n entries of
// the scratch array): ONE LDS round trip.  cut = min(first non-swapping left stopper, last swapping right stopper) as before.
__device__ int partition_reg2(uint64_t* a, int first, int last, uint64_t* box, int lane) {
  const KeyLess less;
  const int n = last - first;  // 65 .. 128 (uniform)
  uint64_t v0 = a[first + lane], v1 = a[first + min(64 + lane, n - 1)];
  const bool in0 = lane >= 1, in1 = 64 + lane < n;   // (index 0 is the pivot's slot)
  auto get = [&](int i) { return i < 64 ? readlane_u64(v0, i) : readlane_u64(v1, i - 64); };  // i uniform, relative to first
  const int mid = n / 2;
  const uint64_t va = get(1), vb = get(mid), vc = get(n - 1);
  int sel;
  if (less(va, vb)) sel = less(vb, vc) ? mid : (less(va, vc) ? n - 1 : 1);
  else if (less(va, vc)) sel = 1;
  else sel = less(vb, vc) ? n - 1 : mid;
  const uint64_t vf = get(0), pivot = get(sel);
  if (lane == 0) v0 = pivot;
  if (sel < 64) {
    if (lane == sel) v0 = vf;
  } else if (lane == sel - 64) {
    v1 = vf;
  }
  const bool stL0 = in0 && !less(v0, pivot), stL1 = in1 && !less(v1, pivot);
  const bool stR0 = in0 && !less(pivot, v0), stR1 = in1 && !less(pivot, v1);
  const uint64_t mL0 = __ballot(stL0), mL1 = __ballot(stL1), mR0 = __ballot(stR0), mR1 = __ballot(stR1);
  const uint64_t lt = lanemask_lt(), gt = ~(lt | (1ull << lane));
  const int belowL0 = __popcll(mL0 & lt), belowL1 = __popcll(mL0) + __popcll(mL1 & lt);   // left stoppers below this element
  const int aboveR0 = __popcll(mR0 & gt) + __popcll(mR1), aboveR1 = __popcll(mR1 & gt);   // right stoppers above it
  const bool swL0 = stL0 && aboveR0 > belowL0, swL1 = stL1 && aboveR1 > belowL1;
  const bool swR0 = stR0 && belowL0 > aboveR0, swR1 = stR1 && belowL1 > aboveR1;
  const uint64_t sL0 = __ballot(swL0), sL1 = __ballot(swL1), sR0 = __ballot(swR0), sR1 = __ballot(swR1);
  const uint64_t nsL0 = mL0 & ~sL0, nsL1 = mL1 & ~sL1;
  const int INF = 1 << 30;
  const int lk = nsL0 ? (int)__builtin_ctzll(nsL0) : (nsL1 ? 64 + (int)__builtin_ctzll(nsL1) : INF);
  const int rk = sR0 ? (int)__builtin_ctzll(sR0) : (sR1 ? 64 + (int)__builtin_ctzll(sR1) : INF);
  const int half = n / 2;
  if (swL0) box[belowL0] = v0;
  if (swL1) box[belowL1] = v1;
  if (swR0) box[half + aboveR0] = v0;
  if (swR1) box[half + aboveR1] = v1;
  wsync();
  if (swL0) v0 = box[half + belowL0];
  if (swL1) v1 = box[half + belowL1];
  if (swR0) v0 = box[aboveR0];
  if (swR1) v1 = box[aboveR1];
  a[first + lane] = v0;
  if (in1) a[first + 64 + lane] = v1;
  wsync();
  return first + min(lk, rk);
}

// The same std::sort replica run by a WHOLE workgroup.  __introsort_loop is a binary tree of partitions: a segment
// (first, last, depth) is partitioned, and both halves continue with depth - 1 -- nothing else is shared between them.
// So the tree is walked breadth first: every wave takes segments of the current level (partition_wave on disjoint ranges
// of `a`, stopper lists at the segment's own offset of Li / Ri), children longer than 16 go to the next level's list.
// Identical result, log(n / 16) rounds instead of n / 16 sequential partitions (17 -> ~5 us for the ~130 expandable
// nodes of a level-0 quadtree).  segs: 2 x 256 packed segments + 2 counters (u32).
__device__ __forceinline__ void introsort_block(uint64_t* a, int n, uint64_t* tmp, uint16_t* Li, uint16_t* Ri, uint32_t* segs) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
  const KeyLess less;
#ifdef OCT_PROF
  long long smk[24]; int nsmk = 0;
#define SMK() do { if (nsmk < 24) smk[nsmk++] = wall_clock64(); } while (0)
#else
#define SMK() do {} while (0)
#endif
  SMK();
  if (n > 16) {  // (uniform)
    uint32_t* cnt = segs + 512;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    if (tid == 0) {
      segs[0] = (uint32_t)n << 13 | (uint32_t)(2 * lg) << 26;  // first : 13 | last : 13 | depth : 6
      cnt[0] = 1;
      cnt[1] = 0;
    }
    __syncthreads();
    int cur = 0;
    for (;;) {
      const int ns = (int)cnt[cur];
      if (ns == 0) break;
      for (int k = wv; k < ns; k += nw) {
        const uint32_t sg = segs[cur * 256 + k];
        const int first = (int)(sg & 0x1FFF), last = (int)((sg >> 13) & 0x1FFF), depth = (int)(sg >> 26);
        if (last - first <= 64) {
          introsort_seg_reg(a, first, last, depth, lane);  // the whole sub-tree of this segment, in registers
        } else if (depth == 0) {
          if (lane == 0) is_heapsort<uint64_t, KeyLess>(a, first, last, less);
        } else {
          const int cut = last - first <= 128 ? partition_reg2(a, first, last, tmp + first, lane)
                                              : partition_wave(a, first, last, Li + first, Ri + first, lane);
          if (lane == 0) {
            const uint32_t d = (uint32_t)(depth - 1) << 26;
            if (cut - first > 16) segs[(cur ^ 1) * 256 + atomicAdd(&cnt[cur ^ 1], 1u)] = (uint32_t)first | (uint32_t)cut << 13 | d;
            if (last - cut > 16) segs[(cur ^ 1) * 256 + atomicAdd(&cnt[cur ^ 1], 1u)] = (uint32_t)cut | (uint32_t)last << 13 | d;
          }
        }
      }
      SMK();
      __syncthreads();
      if (tid == 0) cnt[cur] = 0;
      cur ^= 1;
      __syncthreads();
      SMK();
    }
  }
  SMK();
  // __final_insertion_sort == stable rank inside a +-16 window
  // (these node-list steps are latency chains of one wave's instructions: the window's 33 comparisons are dealt to 4 / 2
  // lanes per element when the workgroup has lanes to spare, partial counts meet by DPP quad / pair exchanges)
  {
    const int per = (4 * n <= nt) ? 4 : ((2 * n <= nt) ? 2 : 1);  // lanes per element (uniform)
    const int chunk = (33 + per - 1) / per;                        // 9, 17 or 33 window positions per lane
    for (int i0 = 0; i0 < n * per; i0 += nt) {
      const int t = i0 + tid, i = min(t / per, n - 1), sub = t % per;
      const uint64_t vi = a[i];
      int c = 0;
      const int jb = i - 16 + sub * chunk, je = min(jb + chunk, i + 17);
      for (int j = jb; j < je; j++) {
        const uint64_t vj = a[min(max(j, 0), n - 1)];
        const bool inw = j >= 0 && j < n;
        c += (inw && (less(vj, vi) || (!less(vi, vj) && j < i))) ? 1 : 0;
      }
      if (per >= 2) c += __shfl_xor(c, 1);
      if (per >= 4) c += __shfl_xor(c, 2);
      if (t < n * per && sub == 0) tmp[max(0, i - 16) + c] = vi;
    }
  }
  SMK();
  __syncthreads();
  for (int i = tid; i < n; i += nt) a[i] = tmp[i];
  __syncthreads();
  SMK();
#if defined(OCT_PROF) && !defined(OCT_NO_SORT_PRINT)
  if (tid == 0 && blockIdx.x == 0 && gridDim.x > 1) {
    printf("sort n=%d:", n);
    for (int i = 1; i < nsmk; i++) printf(" %d", (int)(smk[i] - smk[i - 1]));
    printf("  (x10 ns: init | {work, barriers} per round | - | final rank | copy back)\n");
  }
#endif
#undef SMK
}

// Test entry: sort n elements (one block, dynamic LDS): 512 threads = the workgroup version the quadtree runs,
// 64 threads = the single-wave version.
__global__ __launch_bounds__(512) void k_debug_sort(uint64_t* v, int n) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);
  uint64_t* tmp = a + n;
  uint32_t* segs = reinterpret_cast<uint32_t*>(tmp + n);  // 514 u32, also the single-wave version's stack
  uint16_t* Li = reinterpret_cast<uint16_t*>(segs + 516);
  uint16_t* Ri = Li + n + 4;
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += blockDim.x) a[i] = v[i];
  __syncthreads();
  if (blockDim.x == 64) introsort_wave(a, n, tmp, Li, Ri, reinterpret_cast<int*>(segs), tid);
  else introsort_block(a, n, tmp, Li, Ri, segs);
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) v[i] = a[i];
}
hipError_t launch_debug_sort(uint64_t* d_v, int n, hipStream_t s) {
  const size_t lds = (size_t)n * 16 + 516 * 4 + (size_t)(2 * n + 16) * 2 + 64;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_debug_sort),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  static int flip = 0;  // alternate the two versions: the test calls this many times
  hipLaunchKernelGGL(k_debug_sort, dim3(1), dim3((flip++ & 1) ? 64 : 512), lds, s, d_v, n);
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
  // (the floors are the histogram variant's tables for <= 2 roots: code -> node map in cnt4, histogram + leaf table in cntr)
  o.cnt4 = take(maxn * 16 > 4096 ? maxn * 16 : 4096);
  o.cntr = take(maxn * 16 * rep > 16400 ? maxn * 16 * rep : 16400);  // quadrant counters, `rep` replicas each (see OctCtx::cntr)
  o.scan = take(maxn * 8);  // u64 scan values; reused as best[] at the end
  o.cpos = take(maxn * 8);  // [cpos, tsum) is also the histogram variant's coordinate-table region (OctCtx::tab): keep contiguous
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
// Counter replicas: 2 (1 for very large per-level quotas).  Four replicas (rounds 1 - 4) cost 56 KB = 44 of a CU's 128 LDS
// granules per workgroup; with two it is 50 KB = 40 granules, the workgroup itself is 1.5 us faster (fewer replicas to sum) and the
// kernels that run beside it have room for one more workgroup per CU (+ 0.4 % pairs/s with three handles, A/B on one box).
__host__ __device__ inline int oct_rep(const Geom& g) {
  const int maxn = oct_maxn(g), maxcells = oct_maxcells(g);
  if (oct_layout(maxn, maxcells, 2).total <= 78 * 1024) return 2;
  return 1;
}
size_t octree_lds_bytes(const Geom& g) { return (size_t)oct_layout(oct_maxn(g), oct_maxcells(g), oct_rep(g)).total; }

constexpr int OCT_NT = 512;  // threads per quadtree block: halves the key-loop trip counts vs 256, 2 blocks/CU stay resident
// Exclusive scan of n u64 values in LDS (in place) by an OCT_NT-thread block; returns the total.
// Packed fields must not overflow into each other (callers keep each field < 2^21).
// Per-thread chunk sums are scanned inside each wave with DPP row shifts / broadcasts (no barriers); only the
// wave totals go through LDS: 2 barriers per call instead of the 18 of a Hillis-Steele scan over 256 threads.
// the same scan by wave 0 alone (n <= 64, called by its 64 lanes only): exclusive prefix in place, returns the total
__device__ __forceinline__ uint64_t wave0_scan_u64(uint64_t* v, int n) {
  const int lane = threadIdx.x;
  const uint64_t s = lane < n ? v[lane] : 0;
  const uint64_t incl = wave_scan_dpp_u64(s);
  if (lane < n) v[lane] = incl - s;
  return readlane_u64(incl, 63);
}
__device__ __forceinline__ uint64_t block_scan_u64(uint64_t* v, int n, uint64_t* tsum) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int per = (n + OCT_NT - 1) / OCT_NT;
  const int b = min(tid * per, n), e = min(b + per, n);
  uint64_t s = 0;
  for (int i = b; i < e; i++) s += v[i];
  const uint64_t incl = wave_scan_dpp_u64(s);
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
        if (j0 * OCT_NT >= n) continue;  // whole group past the last candidate (uniform): a level-4 workgroup fills 9 of 32 slots
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
  uint16_t* owner;  // gather only: candidate k -> FAST cell, aliasing every table in front of tsum (ownerBytes)
  int ownerBytes;
  uint8_t* tab;     // histogram variant: per-coordinate tables (path codes for sweep 1, canonical ranks for the winner sweeps),
  int tabBytes;     // aliasing cpos / e / mark / bestk at times those are idle
};

// Dense candidate k -> (cell, index in cell) for the gather.  Every cell writes its index over its range of an LDS map
// (a few fire-and-forget ds_write per thread), so a candidate's slot address is two LDS reads; the binary search over the
// cell prefix it replaces was ten DEPENDENT reads per candidate, 12 of a level-0 workgroup's 46 us.  The map aliases the
// node tables, which nothing uses yet; levels with more candidates than it holds keep the search.
template <class CD>
__device__ __forceinline__ void octree_gather(CD& cd, const OctCtx& c, const LevelDev& L, int n, int cells,
                                              const uint32_t* __restrict__ sparse) {
  if (2 * n <= c.ownerBytes) {  // (uniform)
    for (int cell = threadIdx.x; cell < cells; cell += OCT_NT) {
      const int b = c.cellpre[cell], e = min(c.cellpre[cell + 1], n);
      for (int i = b; i < e; i++) c.owner[i] = (uint16_t)cell;
    }
    __syncthreads();
    cd.fill([&](int k) {
      const int cell = c.owner[k];
      return sparse[(long long)cell * L.cellCap + (k - c.cellpre[cell])];
    });
    __syncthreads();  // the map's bytes become node tables again
  } else {
    // dense index k -> (cell, i) by binary search over the LDS-resident prefix: balanced, no per-cell loops
    cd.fill([&](int k) {
      int lo = 0, hi = cells;  // largest cell with cellpre[cell] <= k
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (c.cellpre[mid] <= k) lo = mid; else hi = mid;
      }
      return sparse[(long long)lo * L.cellCap + (k - c.cellpre[lo])];
    });
  }
}

template <bool REG>
__device__ __forceinline__ void octree_body(const Geom& g, const LevelDev& L, const OctCtx& c, int n, int cells,
                                            const uint32_t* __restrict__ sparse, uint32_t* __restrict__ keys,
                                            uint16_t* __restrict__ kn, uint32_t* __restrict__ out,
                                            int* __restrict__ outCount, int profLevel) {
  const int tid = threadIdx.x;
#ifdef OCT_PROF  // section timing of one block (tools/octree_prof.py; build with -DOCT_PROF, select the level with
                 // ORBX_OCTREE_PROF_LEVEL): thread 0 prints the 10 ns ticks between the MK() markers
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
  octree_gather(cd, c, L, n, cells, sparse);
  MK();
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
  // ---- phase 2: expand the largest nodes first until the quota is reached (:678-735)
  while (!finish) {
    const int prevSize = nA;
    uint64_t* E = c.ebuf + ecur * c.maxn;
    uint64_t* E2 = c.ebuf + (ecur ^ 1) * c.maxn;
    // the whole workgroup sorts; scratch: scan (stopper lists), E2 (rank scatter), tsum (segment lists)
    introsort_block(E, nE, E2, reinterpret_cast<uint16_t*>(scan), reinterpret_cast<uint16_t*>(scan) + c.maxn + 4,
                    reinterpret_cast<uint32_t*>(tsum));
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
  if (tid == 0 && profLevel) {  // (the caller passes 1 for the selected level)
    printf("n=%d nA=%d :", n, nA);
    for (int i = 1; i < nmk; i++) printf(" %d", (int)(tmk[i] - tmk[i - 1]));
    printf("\n");
  }
#endif
#undef MK
}

// ---- quadtree from a path-code histogram ---------------------------------------------------------------------------
// The child a candidate falls into depends only on the node's rectangle (DivideNode :494-495 halves it with ceil), never
// on the other candidates, so every candidate's quadrant sequence down to depth OCT_HD can be computed in ONE sweep
// (registers only) and the number of candidates in ANY node of depth <= OCT_HD is a prefix count: a histogram of the
// depth-OCT_HD path codes plus its 4-ary sums.  All of DistributeOctTree then runs on the node lists alone (<= N + 3
// nodes: the same list layout, phase-1 / phase-2 rules and std::sort replica as octree_body), and the candidates are
// touched twice more at the end (winner per node, output) -- 3 candidate sweeps instead of ~26 with 12 rounds of LDS
// atomics.  A node of depth OCT_HD that must be split (dense clusters) makes the block fall back to octree_body, which
// gives the same result.  LDS: the histogram, the leaf table and the code -> node map reuse octree_body's counter areas.
constexpr int OCT_HD = 5;
__host__ __device__ inline int oct_hbase(int nIni, int d) { return nIni * (((1 << (2 * d)) - 1) / 3); }  // entries above depth d
__host__ __device__ inline bool oct_hist_fits(int nIni) {  // oct_layout reserves the tables of up to 2 roots
  return nIni <= 2 && oct_hbase(nIni, OCT_HD + 1) * 6 <= 16400 && nIni * (1 << (2 * OCT_HD)) * 2 <= 4096;
}

// returns false when a node deeper than the table had to be split (caller falls back to octree_body)
__device__ __forceinline__ bool octree_hist_body(const Geom& g, const LevelDev& L, const OctCtx& c, int n, int cells,
                                                 const uint32_t* __restrict__ sparse, uint32_t* __restrict__ keys,
                                                 uint32_t* __restrict__ out, int* __restrict__ outCount, int prof) {
  const int tid = threadIdx.x;
#ifdef OCT_PROF  // section timing of the selected level's block of image 0 (tools/octree_prof.py)
  long long hmk[48]; int nhmk = 0;
#define HMK() do { if (nhmk < 48) hmk[nhmk++] = wall_clock64(); } while (0)
#else
#define HMK() do {} while (0)
#endif
  HMK();
  OctCands<true> cd;
  cd.keys = keys;
  cd.kn = nullptr;
  cd.n = n;
  octree_gather(cd, c, L, n, cells, sparse);
  HMK();  // gather
  const int N = L.quota;
  struct Buf {
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
  const int W = L.w - 2 * kBorder, H = L.h - 2 * kBorder;
  const int nIni = (int)roundf((float)W / (float)H);  // :566
  const float hX = (float)W / (float)nIni;             // :568
  uint32_t* hist = c.cntr;                                                   // [oct_hbase(nIni, OCT_HD + 1)] counts by (depth, code)
  uint16_t* leafAt = reinterpret_cast<uint16_t*>(hist + oct_hbase(nIni, OCT_HD + 1));  // same shape: final node index or 0xFFFF
  uint16_t* t5 = reinterpret_cast<uint16_t*>(c.cnt4);                        // depth-OCT_HD code -> final node index
  uint16_t* ndcBuf = c.cpos;                                                 // [2][maxn]: depth << 13 | code of every node
  auto ndc = [&](int b) { return ndcBuf + b * c.maxn; };
  uint64_t* scan = c.scan;
  uint16_t* mark = c.mark;
  uint64_t* tsum = c.tsum;
  int* s_i = c.s_i;
  const int nH = oct_hbase(nIni, OCT_HD + 1);
  for (int i = tid; i < nH; i += OCT_NT) hist[i] = 0;
  if (tid == 0) s_i[3] = 0;  // fallback flag
  // The quadrant sequence of a candidate separates: its x bits depend on x alone (root column included), its y bits on y
  // alone -- DivideNode halves a rectangle's sides independently (:494-495).  So the depth-OCT_HD path code is px[x] + py[y]
  // with two small per-coordinate tables (W + H entries) instead of ~90 instructions per candidate (an IEEE division for
  // the root and five rounds of midpoints / selects): sweep 1 was 7.7 of a level-1 workgroup's 46 us
  // (profiles/r5a_octree_sections.txt).  The tables sit in LDS that is idle until the node lists start.
  const bool pathTab = 2 * (W + H) + 8 <= c.tabBytes;  // (uniform; very large levels keep the arithmetic form)
  uint16_t* px = reinterpret_cast<uint16_t*>(c.tab);
  uint16_t* py = px + ((W + 3) & ~3);
  auto x_path = [&](int x) {
    const int r = (int)((float)x / hX);
    int x0 = (int)(hX * (float)r), x1 = (int)(hX * (float)(r + 1));
    uint32_t code = (uint32_t)r;
#pragma unroll
    for (int d = 0; d < OCT_HD; d++) {
      const int mx = x0 + ((x1 - x0 + 1) >> 1);
      const int qx = x >= mx;
      x0 = qx ? mx : x0;
      x1 = qx ? x1 : mx;
      code = code * 4 + (uint32_t)qx;
    }
    return code;
  };
  auto y_path = [&](int y) {
    int y0 = 0, y1 = H;
    uint32_t code = 0;
#pragma unroll
    for (int d = 0; d < OCT_HD; d++) {
      const int my = y0 + ((y1 - y0 + 1) >> 1);
      const int qy = y >= my;
      y0 = qy ? my : y0;
      y1 = qy ? y1 : my;
      code = code * 4 + (uint32_t)(qy << 1);
    }
    return code;
  };
  if (pathTab) {
    for (int i = tid; i < W + H; i += OCT_NT) {
      if (i < W) px[i] = (uint16_t)x_path(i);
      else py[i - W] = (uint16_t)y_path(i - W);
    }
  }
  __syncthreads();
  // ---- sweep 1: path code of every candidate, histogram at depth OCT_HD
  {
    uint32_t* h5 = hist + oct_hbase(nIni, OCT_HD);
    if (pathTab) {
      cd.sweep([&](int, uint32_t key, uint32_t& nd) {
        const uint32_t code = (uint32_t)px[key_x(key)] + (uint32_t)py[key_y(key)];
        nd = code;
        atomicAdd(&h5[code], 1u);
      });
    } else {
      cd.sweep([&](int, uint32_t key, uint32_t& nd) {
        const uint32_t code = x_path(key_x(key)) + y_path(key_y(key));
        nd = code;
        atomicAdd(&h5[code], 1u);
      });
    }
  }
  __syncthreads();
  HMK();  // sweep 1 (path codes + depth-5 histogram)
  for (int d = OCT_HD - 1; d >= 0; d--) {  // 4-ary sums
    const uint32_t* ch = hist + oct_hbase(nIni, d + 1);
    uint32_t* pa = hist + oct_hbase(nIni, d);
    for (int i = tid; i < (nIni << (2 * d)); i += OCT_NT) pa[i] = ch[4 * i] + ch[4 * i + 1] + ch[4 * i + 2] + ch[4 * i + 3];
    __syncthreads();
  }
  auto child_count = [&](uint32_t dc, int q) {  // candidates in child q of node (depth, code); depth < OCT_HD
    const int d = (int)(dc >> 13);
    return hist[oct_hbase(nIni, d + 1) + (int)(dc & 0x1FFF) * 4 + q];
  };
  // ---- roots (:575-601)
  if (tid == 0) {
    int na = 0;
    for (int i = 0; i < nIni; i++) {
      if (hist[i] == 0) continue;
      nx0[0][na] = (int16_t)(int)(hX * (float)i);
      nx1[0][na] = (int16_t)(int)(hX * (float)(i + 1));
      ny0[0][na] = 0;
      ny1[0][na] = (int16_t)H;
      ncnt[0][na] = hist[i];
      ndc(0)[na] = (uint16_t)i;
      na++;
    }
    s_i[0] = na;
  }
  __syncthreads();
  int nA = s_i[0];
  int cur = 0;
  bool finish = false;
  int nE = 0, ecur = 0;
  HMK();  // 4-ary sums + roots
  // ---- phase 1: split every expandable node per pass (:610-677)
  // One pass = flags, scan, scatter of the next list.  While the list fits one wave (the first passes: 1 - 2 roots -> 8 -> 32 nodes)
  // wave 0 runs the passes ALONE, in program order on the LDS lists -- no workgroup barriers (four per pass, ~1 us of a 16-wave
  // barrier chain each pass for a handful of nodes); the other waves wait at one barrier for the result.
  // returns 0 = continue, 1 = finished, 2 = -> phase 2, 3 = a node deeper than the table (caller falls back)
  auto pass = [&](auto waveTag) -> int {
    constexpr bool kWave = decltype(waveTag)::value;
    auto sync = [&]() {
      if (kWave) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
      } else {
        __syncthreads();
      }
    };
    const int prevSize = nA;
      for (int i = tid; i < nA; i += OCT_NT) {
        uint64_t cc = 0, nm = 1, ce = 0;
        if (ncnt[cur][i] > 1) {
          nm = 0;
          const uint32_t dc = ndc(cur)[i];
          if ((dc >> 13) >= (uint32_t)OCT_HD) {
            s_i[3] = 1;
          } else {
            for (int q = 0; q < 4; q++) {
              const uint32_t cq = child_count(dc, q);
              cc += cq > 0;
              ce += cq > 1;
            }
          }
        }
        scan[i] = cc | (nm << 21) | (ce << 42);
      }
      sync();
      if (s_i[3]) return 3;
      const uint64_t tot = kWave ? wave0_scan_u64(scan, nA) : block_scan_u64(scan, nA, tsum);
      const int tc = (int)(tot & 0x1FFFFF), tnm = (int)((tot >> 21) & 0x1FFFFF), tce = (int)(tot >> 42);
      const int nxt = cur ^ 1;
      for (int i = tid; i < nA; i += OCT_NT) {
        const uint64_t pre = scan[i];
        const int pc = (int)(pre & 0x1FFFFF), pnm = (int)((pre >> 21) & 0x1FFFFF), pce = (int)(pre >> 42);
        if (ncnt[cur][i] > 1) {
          const uint32_t dc = ndc(cur)[i];
          const uint32_t cdc = (((dc >> 13) + 1) << 13) | ((dc & 0x1FFF) * 4);
          const int x0 = nx0[cur][i], x1 = nx1[cur][i], y0 = ny0[cur][i], y1 = ny1[cur][i];
          const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
          uint32_t cq4[4];
          int cc = 0;
          for (int q = 0; q < 4; q++) {
            cq4[q] = child_count(dc, q);
            cc += cq4[q] > 0;
          }
          const int base = tc - (pc + cc);
          int after = 0, eb = pce;
          for (int q = 0; q < 4; q++) {  // E entries in creation order n1..n4
            if (cq4[q] > 1) {
              int rank_after = 0;
              for (int q2 = q + 1; q2 < 4; q2++) rank_after += cq4[q2] > 0;
              const int cx0 = (q & 1) ? x0 + hx : x0;
              (c.ebuf + ecur * c.maxn)[eb++] = ((uint64_t)cq4[q] << 28) | ((uint64_t)(uint32_t)cx0 << 16) | (uint64_t)(base + rank_after);
            }
          }
          for (int q = 3; q >= 0; q--) {  // list order n4,n3,n2,n1
            if (cq4[q] == 0) continue;
            const int pos = base + after++;
            nx0[nxt][pos] = (int16_t)((q & 1) ? x0 + hx : x0);
            nx1[nxt][pos] = (int16_t)((q & 1) ? x1 : x0 + hx);
            ny0[nxt][pos] = (int16_t)((q & 2) ? y0 + hy : y0);
            ny1[nxt][pos] = (int16_t)((q & 2) ? y1 : y0 + hy);
            ncnt[nxt][pos] = cq4[q];
            ndc(nxt)[pos] = (uint16_t)(cdc + q);
          }
        } else {
          const int pos = tc + pnm;
          nx0[nxt][pos] = nx0[cur][i];
          nx1[nxt][pos] = nx1[cur][i];
          ny0[nxt][pos] = ny0[cur][i];
          ny1[nxt][pos] = ny1[cur][i];
          ncnt[nxt][pos] = ncnt[cur][i];
          ndc(nxt)[pos] = ndc(cur)[i];
        }
      }
    sync();
    cur = nxt;
    nA = tc + tnm;
    nE = tce;
    if (nA >= N || nA == prevSize) return 1;
    if (nA + 3 * nE > N) return 2;  // -> phase 2
    return 0;
  };
  int state = 0;
  if (nA <= 64) {  // (uniform)
    if (tid < 64) {
      while (state == 0 && nA <= 64) state = pass(std::true_type{});
      if (tid == 0) {
        s_i[4] = state;
        s_i[5] = nA;
        s_i[6] = nE;
        s_i[7] = cur;
      }
    }
    __syncthreads();
    state = s_i[4];
    nA = s_i[5];
    nE = s_i[6];
    cur = s_i[7];
  }
  while (state == 0) state = pass(std::false_type{});
  if (state == 3) return false;
  finish = state == 1;
  HMK();  // phase 1
  // ---- phase 2: expand the largest nodes first until the quota is reached (:678-735)
  while (!finish) {
    const int prevSize = nA;
    uint64_t* E = c.ebuf + ecur * c.maxn;
    uint64_t* E2 = c.ebuf + (ecur ^ 1) * c.maxn;
    // the whole workgroup sorts; scratch: scan (stopper lists), E2 (rank scatter), tsum (segment lists)
    introsort_block(E, nE, E2, reinterpret_cast<uint16_t*>(scan), reinterpret_cast<uint16_t*>(scan) + c.maxn + 4,
                    reinterpret_cast<uint32_t*>(tsum));
    HMK();  // sort
    if (tid == 0) {
      s_i[1] = nE;  // cut (exclusive count of processed) defaults to all
      s_i[2] = 0;   // broke
    }
    for (int i = tid; i < nA; i += OCT_NT) mark[i] = 0;
    __syncthreads();
    // scan over the processing order m (largest first): c (children), ce (expandable children)
    for (int m = tid; m < nE; m += OCT_NT) {
      const int nd = (int)(E[nE - 1 - m] & 0xFFFF);
      mark[nd] = (uint16_t)(m + 1);
      const uint32_t dc = ndc(cur)[nd];
      uint64_t cc = 0, ce = 0;
      if ((dc >> 13) >= (uint32_t)OCT_HD) {
        s_i[3] = 1;
      } else {
        for (int q = 0; q < 4; q++) {
          const uint32_t cq = child_count(dc, q);
          cc += cq > 0;
          ce += cq > 1;
        }
      }
      scan[m] = cc | (ce << 21);
    }
    __syncthreads();
    if (s_i[3]) return false;
    block_scan_u64(scan, nE, tsum);
    // first m at which the list reaches N nodes: size after m+1 expansions = nA + C_incl(m) - (m+1)
    for (int m = tid; m < nE; m += OCT_NT) {
      const int nd = (int)(E[nE - 1 - m] & 0xFFFF);
      const uint32_t dc = ndc(cur)[nd];
      int cc = 0;
      for (int q = 0; q < 4; q++) cc += child_count(dc, q) > 0;
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
      int cc = 0, ce = 0;
      if (nE) {
        const uint32_t dc = ndc(cur)[(int)(E[0] & 0xFFFF)];  // m = nE-1
        for (int q = 0; q < 4; q++) {
          const uint32_t cq = child_count(dc, q);
          cc += cq > 0;
          ce += cq > 1;
        }
      }
      tc = nE ? (int)(scan[nE - 1] & 0x1FFFFF) + cc : 0;
      tce = nE ? (int)(scan[nE - 1] >> 21) + ce : 0;
    }
    const int nxt = cur ^ 1;
    // children of processed nodes: later processed first, each group n4..n1
    for (int m = tid; m < nP; m += OCT_NT) {
      const int nd = (int)(E[nE - 1 - m] & 0xFFFF);
      const uint32_t dc = ndc(cur)[nd];
      const uint32_t cdc = (((dc >> 13) + 1) << 13) | ((dc & 0x1FFF) * 4);
      const int x0 = nx0[cur][nd], x1 = nx1[cur][nd], y0 = ny0[cur][nd], y1 = ny1[cur][nd];
      const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
      uint32_t cq4[4];
      int cc = 0;
      for (int q = 0; q < 4; q++) {
        cq4[q] = child_count(dc, q);
        cc += cq4[q] > 0;
      }
      const int pc = (int)(scan[m] & 0x1FFFFF);
      int eb = (int)(scan[m] >> 21);
      const int base = tc - (pc + cc);
      for (int q = 0; q < 4; q++) {
        if (cq4[q] > 1) {
          int rank_after = 0;
          for (int q2 = q + 1; q2 < 4; q2++) rank_after += cq4[q2] > 0;
          const int cx0 = (q & 1) ? x0 + hx : x0;
          E2[eb++] = ((uint64_t)cq4[q] << 28) | ((uint64_t)(uint32_t)cx0 << 16) | (uint64_t)(base + rank_after);
        }
      }
      int after = 0;
      for (int q = 3; q >= 0; q--) {
        if (cq4[q] == 0) continue;
        const int pos = base + after++;
        nx0[nxt][pos] = (int16_t)((q & 1) ? x0 + hx : x0);
        nx1[nxt][pos] = (int16_t)((q & 1) ? x1 : x0 + hx);
        ny0[nxt][pos] = (int16_t)((q & 2) ? y0 + hy : y0);
        ny1[nxt][pos] = (int16_t)((q & 2) ? y1 : y0 + hy);
        ncnt[nxt][pos] = cq4[q];
        ndc(nxt)[pos] = (uint16_t)(cdc + q);
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
        ndc(nxt)[pos] = ndc(cur)[i];
      }
    }
    __syncthreads();
    cur = nxt;
    nA = tc + nKeep;
    nE = tce;
    ecur ^= 1;
    if (broke || nA == prevSize) finish = true;
    HMK();  // rest of the phase-2 round
  }
  // ---- candidate -> final node: every final node marks its (depth, code); a depth-OCT_HD code belongs to its deepest
  // marked ancestor
  for (int i = tid; i < nH; i += OCT_NT) leafAt[i] = 0xFFFF;
  __syncthreads();
  for (int i = tid; i < nA; i += OCT_NT) {
    const uint32_t dc = ndc(cur)[i];
    leafAt[oct_hbase(nIni, (int)(dc >> 13)) + (int)(dc & 0x1FFF)] = (uint16_t)i;
  }
  __syncthreads();
  for (int c5 = tid; c5 < (nIni << (2 * OCT_HD)); c5 += OCT_NT) {
    uint16_t v = 0xFFFF;
#pragma unroll
    for (int d = OCT_HD; d >= 0; d--) {
      const uint16_t w = leafAt[oct_hbase(nIni, d) + (c5 >> (2 * (OCT_HD - d)))];
      v = v == 0xFFFF ? w : v;
    }
    t5[c5] = v;
  }
  __syncthreads();
  HMK();  // leaf table
  // ---- best response per node, first candidate (reference order) wins ties (:741-754), as in octree_body
  uint64_t* best = scan;
  uint64_t* bestr = reinterpret_cast<uint64_t*>(c.cntr);  // [node][nrepB]; overwrites the histogram (no longer needed)
  const int nrepB = c.nrep >= 2 ? c.nrep * 2 : 1, repB = tid & (nrepB - 1);
  if (nrepB > 1)
    for (int i = tid; i < nA * nrepB; i += OCT_NT) bestr[i] = 0;
  else
    for (int i = tid; i < nA; i += OCT_NT) best[i] = 0;
  __syncthreads();
  const float invH = 1.0f / (float)L.hCell, invW = 1.0f / (float)L.wCell;
  // canonical rank of a candidate = the reference's candidate order (cell row, cell column, y, x).  The winner of a node is
  // then DECODED from the node's maximum (the rank determines the pixel), one thread per node: no second candidate sweep
  // ("am I my node's maximum?", rounds 2 - 4).  (Rank tables per coordinate like the path tables were measured and are
  // slower than the arithmetic here: the sweep waits for its LDS reads, t5[nd] first.)
  auto rank_key = [&](uint32_t key) {
    const int xr = key_x(key) - 3, yr = key_y(key) - 3;  // relative to the first detectable pixel (19,19)
    const int cy = (int)(((float)yr + 0.5f) * invH), cx = (int)(((float)xr + 0.5f) * invW);
    const uint32_t rank = (uint32_t)(((cy * L.nCols + cx) * L.hCell + (yr - cy * L.hCell)) * L.wCell + (xr - cx * L.wCell));
    return ((unsigned long long)key_r(key) << 32) | (0xFFFFFFFFu - rank);
  };
  cd.sweep([&](int, uint32_t key, uint32_t& nd) {
    const uint32_t fn = t5[nd];  // final node index
    if (nrepB > 1) atomicMax((unsigned long long*)&bestr[fn * nrepB + repB], rank_key(key));
    else atomicMax((unsigned long long*)&best[fn], rank_key(key));
  });
  __syncthreads();
  const int nOut = min(nA, L.selCap);
  for (int i = tid; i < nOut; i += OCT_NT) {
    uint64_t v = nrepB > 1 ? 0 : best[i];
    if (nrepB > 1)
      for (int r = 0; r < nrepB; r++) v = bestr[i * nrepB + r] > v ? bestr[i * nrepB + r] : v;
    // (every final node holds >= 1 candidate, so v != 0)
    const uint32_t rank = 0xFFFFFFFFu - (uint32_t)v, resp = (uint32_t)(v >> 32);
    const uint32_t xloc = rank % (uint32_t)L.wCell, t1 = rank / (uint32_t)L.wCell;
    const uint32_t yloc = t1 % (uint32_t)L.hCell, t2 = t1 / (uint32_t)L.hCell;
    const uint32_t cx = t2 % (uint32_t)L.nCols, cy = t2 / (uint32_t)L.nCols;
    out[i] = pack_key((int)(cx * L.wCell + xloc) + 3 + kBorder, (int)(cy * L.hCell + yloc) + 3 + kBorder, (int)resp);
  }
  if (tid == 0) *outCount = nOut;
  HMK();  // best-response sweeps
#ifdef OCT_PROF
  if (tid == 0 && prof) {
    printf("hist n=%d nA=%d :", n, nA);
    for (int i = 1; i < nhmk; i++) printf(" %d", (int)(hmk[i] - hmk[i - 1]));
    printf("  (x10 ns: gather | sweep1 | sums+roots | phase 1 | {sort, rest} per phase-2 round | leaf table | best sweeps)\n");
  }
#endif
#undef HMK
  return true;
}

__global__ __launch_bounds__(OCT_NT, 4) void k_octree(Geom g, const uint32_t* __restrict__ cellCand,
                                                const int* __restrict__ cellCount, int* __restrict__ cellPrefix,
                                                uint32_t* __restrict__ cand, int* __restrict__ candCount,
                                                uint16_t* __restrict__ knode, uint32_t* __restrict__ sel,
                                                int* __restrict__ selCount, int profLevel, int forceGlobal, int level0,
                                                int nlv) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  __shared__ int s_i[8];
  const int tid = threadIdx.x;
#ifdef OCT_WG_PROF  // measurement aid: life of every (image, level) workgroup of one launch, x10 ns (make prof PROF_FLAGS=-DOCT_WG_PROF,
                    // tools/octree_frame_prof.py + tools/octwg_summary.py: resolves 0.1 us where the frame's wall time resolves 1 us)
  struct WgPrint {
    long long t0;
    int l, img, on;
    __device__ ~WgPrint() {
      if (on) printf("octwg level %d img %d start %lld end %lld\n", l, img, t0 % 100000000, wall_clock64() % 100000000);
    }
  } wgPrint{wall_clock64(), 0, 0, (int)(threadIdx.x == 0)};
#endif
  // Level-major block order (all images' level 0 first): the 2-per-CU residency then pairs a heavy level-0 / level-1
  // workgroup with a light level-4+ one instead of with another heavy one.
  const int nimg_ = gridDim.x / nlv;
  const int l = level0 + blockIdx.x / nimg_, img = blockIdx.x % nimg_;
#ifdef OCT_WG_PROF
  wgPrint.l = l;
  wgPrint.img = img;
#endif
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
  c.owner = (uint16_t*)smem;
  c.ownerBytes = o.tsum;
  c.tab = smem + o.cpos;
  c.tabBytes = o.tsum - o.cpos;

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
    const int incl = wave_scan_dpp(sum);
    if ((tid & 63) == 63) c.tsum[tid >> 6] = (uint64_t)incl;
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
  // forceGlobal (test hook): 0 = product path, 1 = global-memory candidates (octree_body<false>), 2 = register-resident
  // per-pass sweeps (octree_body<true>, the fallback of the histogram variant)
  if (n <= OCT_KMAX * OCT_NT && forceGlobal != 1) {
    const int W = L.w - 2 * kBorder, H = L.h - 2 * kBorder;
    const int nIni = (int)roundf((float)W / (float)H);
    bool done = false;
    if (forceGlobal == 0 && oct_hist_fits(nIni)) {
      done = octree_hist_body(g, L, c, n, cells, sparse, keys, out, outCount, profLevel == l && img == 0);
      __syncthreads();
    }
    if (!done) octree_body<true>(g, L, c, n, cells, sparse, keys, kn, out, outCount, profLevel == l && img == 0);
  } else {
    octree_body<false>(g, L, c, n, cells, sparse, keys, kn, out, outCount, profLevel == l && img == 0);
  }
}

static int g_octree_force_global_host = 0;
void debug_set_octree_global(int on) { g_octree_force_global_host = on < 0 ? 0 : (on > 2 ? 2 : on); }

hipError_t launch_octree(const Geom& g, int nimg, const uint32_t* cellCand, const int* cellCount, int* cellPrefix,
                         uint32_t* cand, int* candCount, uint16_t* knode, uint32_t* sel, int* selCount, int level0,
                         int level1, hipStream_t s) {
  if (level1 <= level0) return hipSuccess;
  dim3 grid((level1 - level0) * nimg);
#ifdef OCT_PROF
  static const int profLevel = getenv("ORBX_OCTREE_PROF_LEVEL") ? atoi(getenv("ORBX_OCTREE_PROF_LEVEL")) : 0;
#else
  const int profLevel = -1;
#endif
  hipLaunchKernelGGL(k_octree, grid, dim3(OCT_NT), octree_lds_bytes(g), s, g, cellCand, cellCount, cellPrefix, cand,
                     candCount, knode, sel, selCount, profLevel, g_octree_force_global_host, level0, level1 - level0);
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
// T440: the taps of OpenCV 4.0 .. 4.5.0 {18,34,49,55,49,34,18} (they sum to 257: the result saturates at 255) instead of
// {18,34,48,56,48,34,18} (OpenCV >= 4.5.1) -- orbx_set_opencv_compat, Geom::cv440.
template <bool T440>
__global__ __launch_bounds__(256) void k_blur(Geom g, Pyr p, int level0, int level1, int xcdRun) {
  constexpr uint32_t kT2 = T440 ? 49u : 48u, kT3 = T440 ? 55u : 56u;
  __shared__ uint32_t in[BL_TH + 6][BL_TW / 4 + 2 + 1];    // +1: pad against bank conflicts
  __shared__ uint32_t hp[(BL_TH + 6) / 2][BL_TW + 1];      // [row pair][32*(x%4) + x/4] = H(2j, x) | H(2j+1, x) << 16
                                                           // (quad-transposed columns: both passes bank-conflict free)
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  int tile = xcd_run_remap_rt(blockIdx.x, gridDim.x, blockIdx.z, xcdRun);
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
  // Geom::cv440 = 16 / 32: columns [0, w & ~(vec - 1)) are the vector body of OpenCV 4.0 .. 4.5.0's vertical pass, which FLOORS
  // on the 257-sum taps (oracle/orb_oracle.cpp gaussian_blur7); the scalar tail behind it rounds
  const int bodyEnd = T440 && g.cv440 > 1 ? (L.w & ~(g.cv440 - 1)) : 0;
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
      const uint32_t wA = 18u | (34u << 8) | (kT2 << 16) | (kT3 << 24), wB = kT2 | (34u << 8) | (18u << 16);  // taps -3..0 and +1..+3 (LSB = lowest x)
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
      const uint32_t w01 = 18u | (34u << 16), w23 = kT2 | (kT3 << 16), w45 = kT2 | (34u << 16), w6 = 18u;  // even r
      const uint32_t v0 = 18u << 16, v12 = 34u | (kT2 << 16), v34 = kT3 | (kT2 << 16), v56 = 34u | (18u << 16);  // odd r
      const uint32_t rnd = T440 && x0 + 4 * bc + cI < bodyEnd ? 0u : 32768u;
#pragma unroll
      for (int rr = 0; rr < 4; rr++) {
        const int k0 = rr >> 1;
        uint32_t acc;
        if ((rr & 1) == 0) {  // rows r = 4br + rr (even): pairs k0 .. k0+2, then the low half of pair k0+3
          acc = udot2_u16(pr[k0], w01, rnd);
          acc = udot2_u16(pr[k0 + 1], w23, acc);
          acc = udot2_u16(pr[k0 + 2], w45, acc);
          acc = udot2_u16(pr[k0 + 3], w6, acc);
        } else {              // odd r: high half of pair k0, then pairs k0+1 .. k0+3
          acc = udot2_u16(pr[k0], v0, rnd);
          acc = udot2_u16(pr[k0 + 1], v12, acc);
          acc = udot2_u16(pr[k0 + 2], v34, acc);
          acc = udot2_u16(pr[k0 + 3], v56, acc);
        }
        accs[rr][cI] = T440 ? min(acc, 0x00FFFFFFu) : acc;  // result byte = bits 16..23 (257-sum taps: saturated at 255)
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
  if (g.cv440) hipLaunchKernelGGL(k_blur<true>, dim3(tiles, 1, nimg), dim3(256), 0, s, g, p, level0, level1, 1);
  else hipLaunchKernelGGL(k_blur<false>, dim3(tiles, 1, nimg), dim3(256), 0, s, g, p, level0, level1, 1);  // plain tile order (runs: slower)
  return hipGetLastError();
}

// ================================================================================================ slots
// Output slot of every selected keypoint under the serial semantics of :1062-1099: walk levels then list
// order; keypoints inside the lapping area fill from the back, the others from the front.
__global__ __launch_bounds__(256) void k_slots(Geom g, const uint32_t* __restrict__ sel,
                                               const int* __restrict__ selCount, const int* __restrict__ lap,
                                               int* __restrict__ slot, int* __restrict__ nOut,
                                               int* __restrict__ mono) {
  // Round 6: a chain of latencies, not work (14 us for 16 images x 1500 keypoints: thread 0 walked the level counts one load
  // after the other, every thread fetched its keys one dependent load at a time -- twice --, and the 256-wide scan took 16
  // barriers).  Now: the level counts in parallel + a wave scan, a thread's keys requested together and kept in registers (up to
  // kKeep; longer runs re-read), the lapping counts by a DPP wave scan and one combine across the four waves.
  __shared__ int cum[ORBX_MAX_LEVELS + 1];
  __shared__ int wsum[4];
  const int tid = threadIdx.x, img = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 64) {
    const int c = lane < g.nlevels ? selCount[img * g.nlevels + lane] : 0;
    const int inc = wave_scan_dpp(c);            // inclusive
    if (lane < g.nlevels) cum[lane] = inc - c;
    if (lane == g.nlevels - 1) cum[g.nlevels] = inc;
  }
  __syncthreads();
  const int n = cum[g.nlevels];
  const float lap0 = (float)lap[2 * img], lap1 = (float)lap[2 * img + 1];
  const int per = (n + 255) >> 8;
  const int b = min(tid * per, n), e = min(b + per, n);
  constexpr int kKeep = 8;
  auto locate = [&](int gidx, int& lvl, int& idx) {
    lvl = 0;
    while (gidx >= cum[lvl + 1]) lvl++;
    idx = gidx - cum[lvl];
  };
  auto lapping = [&](uint32_t key, int lvl) {
    float x = (float)key_x(key);
    if (lvl != 0) x = x * g.lv[lvl].scale;
    return x >= lap0 && x <= lap1;
  };
  uint32_t keys[kKeep];
  int lvs[kKeep], adr[kKeep];
  int cnt = 0;
  if (per <= kKeep) {
#pragma unroll
    for (int k = 0; k < kKeep; k++) {
      const int i = b + k;
      lvs[k] = 0;
      adr[k] = 0;
      keys[k] = 0;
      if (i < e) {
        int ix;
        locate(i, lvs[k], ix);
        adr[k] = g.lv[lvs[k]].selOff + ix;
        keys[k] = sel[(long long)img * g.selImg + adr[k]];
      }
    }
#pragma unroll
    for (int k = 0; k < kKeep; k++) cnt += (b + k < e && lapping(keys[k], lvs[k])) ? 1 : 0;
  } else {
    for (int i = b; i < e; i++) {
      int lv, ix;
      locate(i, lv, ix);
      cnt += lapping(sel[(long long)img * g.selImg + g.lv[lv].selOff + ix], lv) ? 1 : 0;
    }
  }
  const int incl = wave_scan_dpp(cnt);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int lapBefore = incl - cnt;
  for (int k = 0; k < wave; k++) lapBefore += wsum[k];
  if (per <= kKeep) {
#pragma unroll
    for (int k = 0; k < kKeep; k++) {
      const int i = b + k;
      if (i < e) {
        const bool il = lapping(keys[k], lvs[k]);
        slot[(long long)img * g.selImg + adr[k]] = il ? (n - 1 - lapBefore) : (i - lapBefore);
        lapBefore += il ? 1 : 0;
      }
    }
  } else {
    for (int i = b; i < e; i++) {
      int lv, ix;
      locate(i, lv, ix);
      const bool il = lapping(sel[(long long)img * g.selImg + g.lv[lv].selOff + ix], lv);
      slot[(long long)img * g.selImg + g.lv[lv].selOff + ix] = il ? (n - 1 - lapBefore) : (i - lapBefore);
      lapBefore += il ? 1 : 0;
    }
  }
  if (tid == 0) {
    nOut[img] = n;
    mono[img] = n - (wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  }
}

hipError_t launch_slots(const Geom& g, int nimg, const uint32_t* sel, const int* selCount, const int* lap,
                        int* slot, int* nOut, int* mono, hipStream_t s) {
  hipLaunchKernelGGL(k_slots, dim3(nimg), dim3(256), 0, s, g, sel, selCount, lap, slot, nOut, mono);
  return hipGetLastError();
}

// ================================================================================================ describe
// (the kernel and its design notes follow the tables)
// IC_Angle weights: for window dword item i = 9 r + c (row r = 0..30 <-> v = r - 15, aligned dword c = 0..8) and
// misalignment m = (X - 15) & 3, byte b of the dword is patch column u = 4c + b - m - 15.  Entry .x = 0x01 per byte inside
// the circle (|u| <= umax[|v|]), .y = (u + 16) per such byte (1..31), so that with wd = the four pixels
//   sum(val) = v_dot4(wd, .x)      sum(u * val) = v_dot4(wd, .y) - 16 * sum(val)
// -- two dot products per dword instead of four masked multiply-adds.
struct IcTable {
  uint2 e[4][5 * 64];
};
// Item (trip t, lane): lanes 0..62 = (row phase r0 = lane / 9, dword c = lane % 9), patch row r = r0 + 7 t; items past row 30
// and lane 63 are zero.  No index division at run time, and a lane's window reads are one address plus immediates.
constexpr IcTable make_ic_table() {
  IcTable t{};
  const int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
  for (int m = 0; m < 4; m++)
    for (int tt = 0; tt < 5; tt++)
      for (int ln = 0; ln < 64; ln++) {
        uint32_t mk = 0, wt = 0;
        const int r = ln / 9 + 7 * tt, c = ln % 9;
        if (ln < 63 && r < 31) {
          const int v = r - 15, lim = um[v < 0 ? -v : v];
          for (int b = 0; b < 4; b++) {
            const int u = 4 * c + b - m - 15;
            if (u >= -lim && u <= lim) {
              mk |= 1u << (8 * b);
              wt |= (uint32_t)(u + 16) << (8 * b);
            }
          }
        }
        t.e[m][tt * 64 + ln].x = mk;
        t.e[m][tt * 64 + ln].y = wt;
      }
  return t;
}
__device__ const IcTable c_ic = make_ic_table();

constexpr int DW_ROWS = 43;   // raw window rows / columns
constexpr int DW_RP = 12;     // raw row pitch in dwords (48 bytes >= 43 + 3 bytes of misalignment)
typedef int orbx_v4i __attribute__((ext_vector_type(4)));
__device__ const BlurMfmaTab c_blur_tab_451 = make_blur_mfma_tab<false>();
__device__ const BlurMfmaTab c_blur_tab_440 = make_blur_mfma_tab<true>();
static_assert(DW_RP * 4 == BM_P && DW_ROWS == BM_ROWS, "window pitch of the loader == pitch of the MFMA operand reads");

// k_describe (round 6 form).  IC_Angle + the 7x7 Gaussian + rBRIEF of src/ORBextractor.cc:75-147,1074-1076 on each selected
// keypoint's own 43x43 window (rows / columns -21..+21: the 37x37 rBRIEF footprint plus the blur's 3-px reach; it contains the
// 31x31 IC patch); the Gaussian is evaluated ONLY where descriptors read it, bit-identical to blurring the level first
// (BORDER_REFLECT_101 is applied while a window that leaves the level is loaded).
// An ablation of the rounds 2 - 5 kernel (one keypoint per wave; profiles/r6_describe_ablation.txt) showed what its 98 us per 64
// images were: 30 the scalar front of a wave, 20 the wait for its window, 16 the serial IC -> atan -> sincos chain, 23 the blur,
// 7 rBRIEF -- a chain of latencies per wave, with the launch bound by how many chains a CU holds.  Hence:
//   * a wave owns NK consecutive selected slots of ONE (image, level): the front -- task decode, level geometry, the image's
//     per-level counts, the wave's keys with one vector load -- is paid once per wave, and the constant MFMA operands of the blur and
//     the rBRIEF pattern (packed as signed bytes: 4 registers) stay in registers across its keypoints;
//   * the raw window of keypoint k + 1 is fetched by LDS-DMA (global_load_lds_dwordx4: 129 16-byte chunks = 3 instructions, no
//     registers, no LDS stores) as soon as the window of keypoint k has been read into registers, so it arrives under keypoint
//     k's blur / angle / rBRIEF; results are stored one keypoint late, under the next one's matrix work;
//   * the blur is two banded integer GEMMs on the matrix pipe (orbx_blur_mfma.h);
//   * fastAtan2 and glibc's sincosf are written branch-free (same operations, selects instead of branches);
//   * one wave per workgroup: a finished wave frees its slot at once.
// NK is a launch parameter (make_desc_plan): 8 for full batches (fewer, longer waves), down to 1 for small batches and the
// single-frame entries (latency: all keypoints at once).
// 108 VGPRs = four waves per SIMD: with the windows prefetched the waves no longer need eight per SIMD to hide their loads
// (alone 102 -> 95 us per 64 images; under three handles 76.4 -> 78.5 k pairs/s, profiles/r6_describe_ab.txt).
struct DescPlan {
  int taskOff[ORBX_MAX_LEVELS];   // first task of level l (INT_MAX past the last level); a task = nk consecutive selected slots
  int tasks;                      // per image
  int nk;                         // keypoints per wave (<= 64: the keys ride in the lanes of one register)
  unsigned magic;                 // j / tasks == umulhi(j, magic) for every flat workgroup index of the launch (host-checked; 0: no XCD remap)
};
struct PatternB {
  uint32_t v[256];   // test i: x0 | y0 << 8 | x1 << 16 | y1 << 24 as signed bytes
};
constexpr PatternB make_pattern_b() {
  constexpr int8_t src[1024] = {
#include "orb_pattern31.inc"
  };
  PatternB t{};
  for (int i = 0; i < 256; i++)
    t.v[i] = (uint32_t)(uint8_t)src[4 * i] | ((uint32_t)(uint8_t)src[4 * i + 1] << 8) | ((uint32_t)(uint8_t)src[4 * i + 2] << 16) |
             ((uint32_t)(uint8_t)src[4 * i + 3] << 24);
  return t;
}
__device__ const PatternB c_pattern_b = make_pattern_b();

// cv::fastAtan2: every product / sum rounded separately; branch-free (the two cases of |x| >= |y| by selects)
__device__ __forceinline__ float fast_atan2_sel(float y, float x) {
  const float s = (float)(180.0 / 3.1415926535897932384626433832795);
  const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s,
              p7 = -0.04432655554792128f * s;
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  const bool xge = ax >= ay;
  const float num = xge ? ay : ax, den = xge ? ax : ay;
  const float c = __fdiv_rn(num, __fadd_rn(den, eps));
  const float c2 = __fmul_rn(c, c);
  const float pa = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  float a = xge ? pa : __fsub_rn(90.f, pa);
  a = x < 0 ? __fsub_rn(180.f, a) : a;
  a = y < 0 ? __fsub_rn(360.f, a) : a;
  return a;
}
// glibc sinf / cosf (orbx_sincos.h, the FMA variant), branch-free: the |y| < pi/4 path is the n = 0 case of the reduction
// (n = round(y * 2/pi) = 0 below 0.75, r = y - 0 * pi/2 = y, sign +1: the same operands reach the same polynomials), the
// |y| < 2^-12 case a select; one sine and one cosine polynomial are evaluated either way, the quadrant picks which is which.
__device__ __forceinline__ void glibc_sincosf_sel(float y, float& s_out, float& c_out) {
#pragma clang fp contract(off)
  const uint32_t top = (__builtin_bit_cast(uint32_t, y) >> 20) & 0x7ffu;
  const double x0 = (double)y;
  const double r = x0 * 0x1.45F306DC9C883p+23;
  const int n = top < 0x3f4u ? 0 : (((int)r + 0x800000) >> 24);
  const double hpi = 0x1.921FB54442D18p0;
  const double x = top < 0x3f4u ? x0 : __builtin_fma(-(double)n, hpi, x0);
  const double sgn = ((n + 1) & 2) ? -1.0 : 1.0;
  const bool neg = (n & 2) != 0;
  const double xs = x * sgn, x2 = x * x;
  // sine polynomial of xs
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  const double x3 = xs * x2;
  const double s1 = __builtin_fma(x2, S3, S2);
  const double x7 = x3 * x2;
  const double sp = __builtin_fma(x3, S1, xs);
  const float fs = (float)__builtin_fma(x7, s1, sp);
  // cosine polynomial (coefficients negated for neg: exact sign flips)
  const double sg = neg ? -1.0 : 1.0;
  const double C0 = sg * 0x1p0, C1 = sg * -0x1.ffffffd0c621cp-2, C2 = sg * 0x1.55553e1068f19p-5, C3 = sg * -0x1.6c087e89a359dp-10,
               C4 = sg * 0x1.99343027bf8c3p-16;
  const double x4 = x2 * x2;
  const double c2 = __builtin_fma(x2, C4, C3);
  const double c1 = __builtin_fma(x2, C1, C0);
  const double x6 = x4 * x2;
  const double cc = __builtin_fma(x4, C2, c1);
  const float fc = (float)__builtin_fma(x6, c2, cc);
  const bool odd = (n & 1) != 0, tiny = top < 0x398u;
  s_out = tiny ? y : (odd ? fc : fs);
  c_out = tiny ? 1.0f : (odd ? fs : fc);
}

template <bool T440>
__global__ __launch_bounds__(64, 4) void k_describe(Geom g, Pyr p, DescPlan dp, const uint32_t* __restrict__ sel,
                                                                     const int* __restrict__ selCount, const int* __restrict__ slot,
                                                                     orbx_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                                     int* __restrict__ nOut, int* __restrict__ mono, int xcdImages) {
  __shared__ __attribute__((aligned(16))) uint32_t ldsW[BM_WAVE_BYTES / 4];   // raw window [43 rows][48 B] (+ rows the operands over-read)
  __shared__ __attribute__((aligned(16))) uint32_t ldsP[48 * BM_P / 4];       // blurred patch [37 (48 written) rows][48 B]
  const int NK = dp.nk;
  const int lane = threadIdx.x;
  int t = blockIdx.x, img = blockIdx.y;
  if (xcdImages && dp.magic) {  // all workgroups of an image on ONE XCD (image i -> XCD i mod 8), see k_describe
    const unsigned nbx = gridDim.x, flat = blockIdx.y * nbx + blockIdx.x, full = (gridDim.y / 8u) * 8u;
    if (flat < full * nbx) {
      const unsigned c = flat & 7u, j = flat >> 3, jq = __umulhi(j, dp.magic);
      img = (int)(c + 8u * jq);
      t = (int)(j - jq * nbx);
    }
  }
  // ---- front, once per wave: level of the task, the image's counts, the task's keys
  int l = 0, tbase = 0;
#pragma unroll
  for (int q = 1; q < ORBX_MAX_LEVELS; q++) {
    const bool ge = t >= dp.taskOff[q];
    l += ge ? 1 : 0;
    tbase = ge ? dp.taskOff[q] : tbase;
  }
  const LevelDev L = g.lv[l];
  const int idx0 = (t - tbase) * NK;
  const int* sc = selCount + img * g.nlevels;   // (the array is over-allocated by ORBX_MAX_LEVELS entries: unconditional loads)
  int before = 0, total = 0, cntL = 0;
#pragma unroll
  for (int q = 0; q < ORBX_MAX_LEVELS; q++) {
    const int c = q < g.nlevels ? sc[q] : 0;
    total += c;
    before += q < l ? c : 0;
    cntL = q == l ? c : cntL;
  }
  if (!slot && t == 0 && lane == 0) {  // k_slots skipped: this wave publishes the image's counts
    nOut[img] = total;
    mono[img] = total;
  }
  const int nk = min(NK, cntL - idx0);
  if (nk <= 0) return;
  const long long sbase = (long long)img * g.selImg + L.selOff + idx0;
  const int lk = min(lane, nk - 1);
  const uint32_t keyv = sel[sbase + lk];
  // slot == nullptr: no keypoint can lie in the lapping area, the serial-order slot is (keypoints of the earlier levels) + idx
  const int slotv = slot ? slot[sbase + lk] : before + idx0 + lane;
  int pitch;
  const uint8_t* im = level_ptr(g, p, img, l, pitch);
  const int rowValid = l ? L.pitch : L.w;   // bytes of an image row that may be read (level 0 is the caller's buffer)
  const int bodyEnd = T440 && g.cv440 > 1 ? (L.w & ~(g.cv440 - 1)) : 0;   // (k_blur: the columns OpenCV 4.0 .. 4.5.0's vector body floors)
  // constant operands, resident across the wave's keypoints
  const BlurMfmaTab& tab = T440 ? c_blur_tab_440 : c_blur_tab_451;
  orbx_v4i bh[3], av[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    bh[i] = *reinterpret_cast<const orbx_v4i*>(tab.bh[i][lane]);
    av[i] = *reinterpret_cast<const orbx_v4i*>(tab.av[i][lane]);
  }
  uint32_t patb[4];
#pragma unroll
  for (int gI = 0; gI < 4; gI++) patb[gI] = c_pattern_b.v[64 * gI + lane];
  // lane constants
  const int li = lane & 15, lg = lane >> 4;
  const uint8_t* arow = reinterpret_cast<const uint8_t*>(ldsW) + li * BM_P + 16 * lg;
  uint8_t* bl = reinterpret_cast<uint8_t*>(ldsP);
  uint8_t* prow = bl + li * BM_P + 4 * lg;
  const int icr0 = (lane * 7282) >> 16, icc = lane - 9 * icr0;   // lane / 9, lane % 9
  // DMA chunks of a window: q = lane + 64 j -> (row q / 3, 16-byte part q % 3); chunk 128 (row 42, part 2) by lane 0
  const int q1 = lane + 64, dr0 = (lane * 43691) >> 17, dr1 = (q1 * 43691) >> 17;
  const unsigned doff0 = (unsigned)(__mul24(dr0, pitch) + 16 * (lane - 3 * dr0)), doff1 = (unsigned)(__mul24(dr1, pitch) + 16 * (q1 - 3 * dr1));
  auto win = [&](int k, int& X, int& Y, int& xs, int& mis, bool& interior, bool& dma) {
    const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)keyv, k);
    X = key_x(key);
    Y = key_y(key);
    xs = (X - 21) & ~3;
    mis = (X - 21) - xs;
    interior = X >= 21 && Y >= 21 && X + 21 < L.w && Y + 21 < L.h;
    dma = interior && xs + 48 <= rowValid;
    return key;
  };
  // LDS-DMA by hand: the compiler's own tracking of global_load_lds serialises the three loads of a window (a vmcnt(0) in front
  // of each) and waits for them right behind their issue -- and missed the wait in front of the first window read.  As inline asm
  // it does not see them at all: its waits for its own loads can only become stricter, and the waits for a window are ours.
  const unsigned ldsWaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)ldsW;
  auto issue_dma = [&](int Y, int xs) {
    const uint8_t* base = im + (long long)(Y - 21) * pitch + xs;   // uniform
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsWaddr), "v"(doff0), "s"(base) : "memory");
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsWaddr + 1024u), "v"(doff1), "s"(base) : "memory");
    if (lane == 0)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(ldsWaddr + 2048u), "v"(42u * (unsigned)pitch + 32u), "s"(base) : "memory");
  };
  int X, Y, xs, mis;
  bool interior, dma;
  uint32_t key = win(0, X, Y, xs, mis, interior, dma);
  if (dma) issue_dma(Y, xs);
  // IC table of a keypoint (depends on its window's misalignment): requested one keypoint ahead, behind the matrix work
  uint2 ice[5];
  {
    const uint2* ictab = c_ic.e[(6 + mis) & 3] + lane;
#pragma unroll
    for (int tt = 0; tt < 5; tt++) ice[tt] = ictab[64 * tt];
  }
  // results of a keypoint are stored one keypoint later, under the next one's matrix work (loads and stores share the wave's
  // memory counter and may complete out of order with respect to each other: the wait for a window is a wait for everything)
  uint32_t pendKp = 0, pendDescLo = 0, pendDescHi = 0;
  int pendSlot = -1;
  auto store_pending = [&]() {
    if (pendSlot >= 0) {
      if (lane < 7) reinterpret_cast<uint32_t*>(kps + ((long long)img * g.outCap + pendSlot))[lane] = pendKp;
      if (lane < 4) reinterpret_cast<uint2*>(desc + ((long long)img * g.outCap + pendSlot) * 32)[lane] = make_uint2(pendDescLo, pendDescHi);
    }
  };
  for (int k = 0; k < nk; k++) {
    const int c0 = (6 + mis) >> 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // window k (DMA) has landed
    if (!dma) {
      if (interior) {  // the 48-byte span of the window's rows ends past the row: dword loads, the last columns clamped (they only feed unused outputs)
        const int r0 = (lane * 5462) >> 16, c = lane - 12 * r0;
        const int cl = min(c, (L.w - 1 - xs) >> 2);
        const uint8_t* base = im + (long long)(Y - 21) * pitch + xs;
        const unsigned off0 = (unsigned)(__mul24(r0, pitch) + 4 * cl), step = 5u * (unsigned)pitch;
        if (lane < 60) {
          uint32_t w[9];
#pragma unroll
          for (int tt = 0; tt < 9; tt++)
            w[tt] = *reinterpret_cast<const uint32_t*>(base + (tt < 8 ? off0 + (unsigned)tt * step : min(off0 + 8u * step, 42u * (unsigned)pitch + 4u * (unsigned)cl)));
#pragma unroll
          for (int tt = 0; tt < 8; tt++) ldsW[lane + 60 * tt] = w[tt];
          if (r0 < 3) ldsW[lane + 480] = w[8];
        }
      } else {  // window crosses the level's edge: BORDER_REFLECT_101, byte by byte (coordinates further out are never used)
        for (int i = lane; i < DW_ROWS * DW_RP; i += 64) {
          const int r = i / DW_RP, c = i - r * DW_RP;
          const int yy = reflect101(min(max(Y - 21 + r, -3), L.h + 2), L.h);
          uint32_t v = 0;
#pragma unroll
          for (int kk = 0; kk < 4; kk++) {
            const int xx = reflect101(min(max(xs + 4 * c + kk, -3), L.w + 2), L.w);
            v |= (uint32_t)im[(long long)yy * pitch + xx] << (8 * kk);
          }
          ldsW[i] = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- the window into registers: matrix operands (pixel - 128 as signed bytes) and the IC dwords
    orbx_v4i ah[3];
#pragma unroll
    for (int rt = 0; rt < 3; rt++) ah[rt] = *reinterpret_cast<const orbx_v4i*>(arow + 16 * rt * BM_P);
    uint32_t icw[5];
    {
      const uint32_t* wp = ldsW + (6 + icr0) * DW_RP + c0 + icc;
#pragma unroll
      for (int tt = 0; tt < 5; tt++) icw[tt] = wp[7 * DW_RP * tt];
    }
    const int n_out_slot = __builtin_amdgcn_readlane(slotv, k);
    const int Xk = X, Yk = Y, misk = mis;
    const uint32_t keyk = key;
    // ---- the window is dead: fetch the next one under this keypoint's work
    if (k + 1 < nk) {
      key = win(k + 1, X, Y, xs, mis, interior, dma);
          if (dma) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of the window has returned before the DMA may overwrite it
        issue_dma(Y, xs);
      }
    }
    // ---- IC_Angle moments on the window dwords read above (the angle itself is evaluated beside the matrix work)
    int m10, m01;
    {
      int srs = 0, sw = 0, rv = icr0 - 15;
      m01 = 0;
#pragma unroll
      for (int tt = 0; tt < 5; tt++) {
        const int rs = (int)__builtin_amdgcn_udot4(icw[tt], ice[tt].x, 0u, false);
        sw = (int)__builtin_amdgcn_udot4(icw[tt], ice[tt].y, (uint32_t)sw, false);
        srs += rs;
        m01 += __mul24(rv, rs);
        rv += 7;
      }
      m10 = wave_sum_dpp(sw - 16 * srs);
      m01 = wave_sum_dpp(m01);
    }
    store_pending();
#pragma unroll
    for (int rt = 0; rt < 3; rt++) ah[rt] ^= (int)0x80808080;
    // ---- blur (orbx_blur_mfma.h; see k_describe)
    constexpr int kBias = BlurMfmaConst<T440>::bias, kKc = BlurMfmaConst<T440>::kc;
    const orbx_v4i cz = {0, 0, 0, 0}, cb = {kBias, kBias, kBias, kBias};
    const int sBody = T440 ? bodyEnd - (Xk - 18 - misk) - 4 * lg : 0;   // patch columns 16 ct + r < sBody of this lane: floor
#pragma unroll
    for (int ct = 0; ct < 3; ct++) {
      orbx_v4i lo = cz, hi = cz;
      uint32_t kcv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) kcv[r] = T440 ? (uint32_t)kKc - (16 * ct + r < sBody ? 32768u : 0u) : (uint32_t)kKc;
#pragma unroll
      for (int rt = 0; rt < 3; rt++) {
        const orbx_v4i x = __builtin_amdgcn_mfma_i32_16x16x64_i8(ah[rt], bh[ct], T440 ? cb : cz, 0, 0, 0);
        const uint32_t pa = __builtin_amdgcn_perm((uint32_t)x[1], (uint32_t)x[0], 0x05010400u);
        const uint32_t pb = __builtin_amdgcn_perm((uint32_t)x[3], (uint32_t)x[2], 0x05010400u);
        lo[rt] = (int)(__builtin_amdgcn_perm(pb, pa, 0x05040100u) ^ 0x80808080u);
        hi[rt] = (int)__builtin_amdgcn_perm(pb, pa, 0x07060302u);
      }
#pragma unroll
      for (int mt = 0; mt < 3; mt++) {
        const orbx_v4i HI = __builtin_amdgcn_mfma_i32_16x16x64_i8(hi, av[mt], cz, 0, 0, 0);
        orbx_v4i cin;
#pragma unroll
        for (int r = 0; r < 4; r++) cin[r] = (int)(((uint32_t)HI[r] << 8) + kcv[r]);
        const orbx_v4i V = __builtin_amdgcn_mfma_i32_16x16x64_i8(lo, av[mt], cin, 0, 0, 0);
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = T440 ? min((uint32_t)V[r], 0x00FFFFFFu) : (uint32_t)V[r];
        const uint32_t t0 = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u), t1 = __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu);
        *reinterpret_cast<uint32_t*>(prow + 16 * mt * BM_P + 16 * ct) = t0 | t1;
      }
    }
    {  // the next keypoint's IC table (the last keypoint re-reads its own: no branch in this block)
      const uint2* ictab = c_ic.e[(6 + mis) & 3] + lane;
#pragma unroll
      for (int tt = 0; tt < 5; tt++) ice[tt] = ictab[64 * tt];
    }
    const float angle = fast_atan2_sel((float)m01, (float)m10);
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    glibc_sincosf_sel(__fmul_rn(angle, factorPI), b, a);   // a = cosf, b = sinf
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- rBRIEF on the blurred patch (blurred (y, x) = patch[y][x + mis])
    const float kMagic = 12582912.0f;
    const uint8_t* centre = bl + 18 * BM_P + 18 + misk;
    const uint32_t kFold = 0x400000u * BM_P + 0x4B400000u;
#pragma unroll
    for (int gI = 0; gI < 4; gI++) {
      const uint32_t pw = patb[gI];
      const float x0 = (float)(int)(int8_t)pw, y0 = (float)(int)(int8_t)(pw >> 8), x1 = (float)(int)(int8_t)(pw >> 16), y1 = (float)((int)pw >> 24);
      const uint32_t iy0 = __builtin_bit_cast(uint32_t, __fadd_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)), kMagic));
      const uint32_t ix0 = __builtin_bit_cast(uint32_t, __fadd_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)), kMagic));
      const uint32_t iy1 = __builtin_bit_cast(uint32_t, __fadd_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)), kMagic));
      const uint32_t ix1 = __builtin_bit_cast(uint32_t, __fadd_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)), kMagic));
      const int t0 = centre[(int)((uint32_t)__mul24((int)iy0, BM_P) + ix0 - kFold)],
                t1 = centre[(int)((uint32_t)__mul24((int)iy1, BM_P) + ix1 - kFold)];
      const uint64_t bits = __ballot(t0 < t1);   // 8 descriptor bytes in the reference's byte / bit order; lane gI keeps them
      pendDescLo = lane == gI ? (uint32_t)bits : pendDescLo;
      pendDescHi = lane == gI ? (uint32_t)(bits >> 32) : pendDescHi;
    }
    {  // the 28-byte keypoint record, dword i in lane i
      const float kx = l ? __fmul_rn((float)Xk, L.scale) : (float)Xk, ky = l ? __fmul_rn((float)Yk, L.scale) : (float)Yk;
      uint32_t v = __builtin_bit_cast(uint32_t, kx);
      v = lane == 1 ? __builtin_bit_cast(uint32_t, ky) : v;
      v = lane == 2 ? __builtin_bit_cast(uint32_t, L.patch) : v;
      v = lane == 3 ? __builtin_bit_cast(uint32_t, angle) : v;
      v = lane == 4 ? __builtin_bit_cast(uint32_t, (float)key_r(keyk)) : v;
      v = lane == 5 ? (uint32_t)l : v;
      v = lane == 6 ? 0xFFFFFFFFu : v;
      pendKp = v;
      pendSlot = n_out_slot;
    }
  }
  store_pending();
}

static DescPlan make_desc_plan(const Geom& g, int nimg) {
  DescPlan dp{};
  // keypoints per wave: as many as keep >= ~12 k waves in the launch (64 images x 1788 slots: 8; 16 images x 1288: 1 -- at small
  // batches the launch is its longest wave, and eight keypoints in a row made a 16-frame step's k_describe 26 -> 33 us)
  dp.nk = (int)std::min<long long>(8, std::max<long long>(1, (long long)nimg * g.selImg / 12288));
  int tks = 0;
  for (int l = 0; l < ORBX_MAX_LEVELS; l++) {
    dp.taskOff[l] = l < g.nlevels ? tks : 0x7fffffff;
    if (l < g.nlevels) tks += (g.lv[l].selCap + dp.nk - 1) / dp.nk;
  }
  dp.tasks = tks;
  // j / tasks by a multiply: m = floor(2^32 / d) + 1 is exact while j * (m * d - 2^32) < 2^32
  const uint64_t d = (uint64_t)tks, m = (1ull << 32) / d + 1, e = m * d - (1ull << 32), jmax = (uint64_t)nimg * d;
  dp.magic = (d > 1 && m < (1ull << 32) && jmax * e < (1ull << 32)) ? (unsigned)m : 0u;
  return dp;
}

hipError_t launch_describe(const Geom& g, const Pyr& p, int nimg, const uint32_t* sel, const int* selCount,
                           const int* slot, orbx_keypoint* kps, uint8_t* desc, int* nOut, int* mono, hipStream_t s) {
  const DescPlan dp = make_desc_plan(g, nimg);
  const dim3 grid(dp.tasks, nimg), block(64);
  if (g.cv440) hipLaunchKernelGGL(k_describe<true>, grid, block, 0, s, g, p, dp, sel, selCount, slot, kps, desc, nOut, mono, nimg >= 8 ? 1 : 0);
  else hipLaunchKernelGGL(k_describe<false>, grid, block, 0, s, g, p, dp, sel, selCount, slot, kps, desc, nOut, mono, nimg >= 8 ? 1 : 0);
  return hipGetLastError();
}

// One launch instead of six device-to-host copies for the single-frame host entries: section = blockIdx.y
// (0 / 1: keypoints of image 0 / 1, 2 / 3: descriptors, 4: uRight, 5: depth, 6: counts); dword copies, count-trimmed.
__global__ __launch_bounds__(256) void k_result_pack(ResultPack a) {
  const int sec = blockIdx.y;
  if (!((a.mask >> sec) & 1)) return;
  if (sec == 6) {
    if (blockIdx.x == 0 && threadIdx.x < 2) {
      const int t = threadIdx.x;
      a.hCnt[t] = t < a.nimg ? (uint32_t)a.nOut[t] : 0u;
      a.hCnt[2 + t] = t < a.nimg ? (uint32_t)a.mono[t] : 0u;
    }
    return;
  }
  const int img = sec < 4 ? (sec & 1) : 0;
  if (img >= a.nimg || (sec >= 4 && !a.stereo)) return;
  const int n = (sec >= 4 && a.fixedN >= 0) ? min(a.fixedN, a.cap) : min(a.nOut[img], a.cap);
  const uint32_t* src;
  uint32_t* dst;
  int len;
  if (sec < 2) {
    src = a.kps + (size_t)img * a.cap * 7; dst = a.hKps + (size_t)img * a.cap * 7; len = n * 7;
  } else if (sec < 4) {
    src = a.desc + (size_t)img * a.cap * 8; dst = a.hDesc + (size_t)img * a.cap * 8; len = n * 8;
  } else {
    src = sec == 4 ? a.uR : a.depth; dst = sec == 4 ? a.hUr : a.hDepth; len = n;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < len; i += gridDim.x * 256) dst[i] = src[i];
}
hipError_t launch_result_pack(const ResultPack& a, hipStream_t s) {
  hipLaunchKernelGGL(k_result_pack, dim3(12, 7), dim3(256), 0, s, a);
  return hipGetLastError();
}

// The dynamic-LDS limit of a kernel is a property of the DEVICE's copy of the function, shared by every handle on it: a second
// handle with a smaller geometry (the reference constructs mpIniORBextractor with 5 x nFeatures beside the left extractor,
// src/Tracking.cc:628-637) must not lower it under the first.  Per device and kernel the largest value ever asked for is kept.
hipError_t raise_dynamic_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> high;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  size_t& h = high[{dev, fn}];
  if (bytes <= h) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) h = bytes;
  return e;
}

hipError_t prepare_kernels(const Geom& g) {
  hipError_t e = raise_dynamic_lds(reinterpret_cast<const void*>(k_octree), octree_lds_bytes(g));
  if (e != hipSuccess) return e;
  if (g.nlevels > 1) {
    e = raise_dynamic_lds(reinterpret_cast<const void*>(k_resize<0>), std::max<size_t>(resize_lds_bytes(g), 1024));
    if (e != hipSuccess) return e;
    e = raise_dynamic_lds(reinterpret_cast<const void*>(k_resize<kResizeNdw>), std::max<size_t>(resize_lds_bytes(g), 1024));
    if (e != hipSuccess) return e;
  }
  return prepare_detect(g);
}

// Test entry: both glibc sinf/cosf variants on the device (compared with the host libm in tests).
__global__ void k_debug_sincos(const float* __restrict__ ang, int n, int fused, float* __restrict__ s, float* __restrict__ c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sv, cv;
  if (fused) glibc_sincosf<true>(ang[i], sv, cv);
  else glibc_sincosf<false>(ang[i], sv, cv);
  s[i] = sv;
  c[i] = cv;
}
hipError_t launch_debug_sincos(const float* ang, int n, int fused, float* s, float* c, hipStream_t st) {
  hipLaunchKernelGGL(k_debug_sincos, dim3((n + 255) / 256), dim3(256), 0, st, ang, n, fused, s, c);
  return hipGetLastError();
}

// Host-callable check of the introsort replica (tests compare with std::sort).
void debug_introsort_host(uint64_t* v, int n) { introsort<uint64_t, KeyLess>(v, n, KeyLess()); }

}  // namespace orbx
