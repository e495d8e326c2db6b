// orbx_resize_tail.hip — the small levels of ComputePyramid (src/ORBextractor.cc:1108-1145) in ONE launch.
//
// Levels lA .. L-1 of a 1.2 pyramid are a few hundred pixels wide: as separate k_resize launches each costs a dependent
// launch + memory-latency chain (7-14 us for < 1 us of arithmetic, and more than that when other handles' kernels are
// resident: profiles/r3c_*).  Here a workgroup owns a band of rows of the LAST level and walks the cascade inside LDS:
// it stages the rows of level lA-1 that band depends on, resizes them to the band's rows of level lA (kept in LDS and
// written to the pyramid), those to level lA+1, ... — the rows a neighbouring band also needs are simply computed twice
// (same arithmetic, same bytes), and every pyramid row is WRITTEN by exactly one band (TailBand::ownEnd).
//
// Arithmetic = k_resize's = cv::resize INTER_LINEAR 8U: t = S[sx]*a0 + S[sx+1]*a1 from the same per-column
// table (build_coefs), D = (((b0*(t0>>4))>>16) + ((b1*(t1>>4))>>16) + 2) >> 2 with the same per-row table.
#include "orbx_device.h"

namespace orbx {

// ds_read2_b32 of the aligned dword pair at byte offset `off` of an LDS tile
__device__ __forceinline__ uint2 lds_pair(const uint8_t* base, int off) {
  const uint32_t* q = reinterpret_cast<const uint32_t*>(base + off);
  return make_uint2(q[0], q[1]);
}

constexpr int kTailNT = 512;  // threads per workgroup

__global__ __launch_bounds__(kTailNT) void k_resize_tail(Geom g, Pyr p, TailPlan tp, const TailBand* __restrict__ bands,
                                                         const uint4* __restrict__ xtab, const int* __restrict__ yofs,
                                                         const short* __restrict__ yab, int img0) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x;
  const int band = blockIdx.x, img = img0 + blockIdx.y;
  const TailBand* tb = bands + (size_t)band * (tp.nT + 1);
  // (tile offsets, not a pointer array: pointers selected at run time decay to generic addresses and the tile reads
  // become flat_load instead of ds_read)
  uint4* rtab = reinterpret_cast<uint4*>(smem + tp.tileBytes[0] + tp.tileBytes[1]);
#ifdef RT_PROF  // phase timing of a few workgroups (tools/resize_prof.py, build with -DRT_PROF): 10 ns ticks
  long long tq[32];
  int nq_ = 0;
#define RT_MK() do { if (nq_ < 32) tq[nq_++] = wall_clock64(); } while (0)
#else
#define RT_MK() do {} while (0)
#endif
  RT_MK();
  // ---- the rows of level lA-1 this band depends on: 16-byte loads, all of a thread's loads in flight before its first
  // LDS store (the level's buffer is 64-byte aligned with a 64-byte pitch: whole uint4s up to align_up(w, 16) exist)
  {
    const int ls = tp.lA - 1;
    const LevelDev S = g.lv[ls];
    const int r0 = tb[0].first, nr = tb[0].last - r0 + 1;
    const int nq = (S.w + 15) >> 4;  // uint4s per row
    const int sp = tp.pitch[0];
    // (a segment that starts at level 1 -- the single-frame plans -- stages the caller's level 0: its own buffer and pitches,
    // 16-byte aligned rows guaranteed by the host side before it picks such a plan)
    const long long spitch = ls ? (long long)S.pitch : p.l0Row;
    const uint8_t* src = (ls ? p.pyr + (long long)img * g.pyrImg + S.off : p.l0 + (long long)img * p.l0Img) + (long long)r0 * spitch;
    const float inv_nq = __builtin_amdgcn_rcpf((float)nq);
    const int total = nr * nq;
    constexpr int kB = 3;
    for (int base = 0; base < total; base += kTailNT * kB) {
      uint4 v[kB];
      int dst[kB];
#pragma unroll
      for (int k = 0; k < kB; k++) {
        const int i = min(base + k * kTailNT + tid, total - 1);   // clamped: the tail re-writes the last element
        const int r = (int)(((float)i + 0.5f) * inv_nq), c = i - r * nq;
        v[k] = *reinterpret_cast<const uint4*>(src + (long long)r * spitch + 16 * c);
        dst[k] = r * sp + 16 * c;
      }
#pragma unroll
      for (int k = 0; k < kB; k++) *reinterpret_cast<uint4*>(smem + dst[k]) = v[k];
    }
  }
  // row tables of every level (source-row offsets inside the level's source tile, b0 << 12, b1 << 12): rtab[rtOff[t] + r];
  // thread i takes entry i of the concatenated tables, so all levels share ONE global-memory round trip
  for (int i = tid; i < tp.rtTotal; i += kTailNT) {
    int t = 0;
    while (t + 1 < tp.nT && i >= tp.rtOff[t + 1]) t++;
    const int r = i - tp.rtOff[t];
    const int dFirst = tb[t + 1].first, nRows = tb[t + 1].last - dFirst + 1;
    if (r < nRows) {
      const LevelDev D = g.lv[tp.lA + t];
      const int Sh = g.lv[tp.lA + t - 1].h, sFirst = tb[t].first, sp = tp.pitch[t];
      const int dy = dFirst + r;
      const int sy = yofs[D.ycoef + dy];
      const uint32_t bb = reinterpret_cast<const uint32_t*>(yab)[D.ycoef + dy];
      const int ra = min(max(sy, 0), Sh - 1) - sFirst, rb = min(max(sy + 1, 0), Sh - 1) - sFirst;
      rtab[i] = make_uint4((uint32_t)(ra * sp), (uint32_t)(rb * sp), (bb & 0xFFFFu) << 12, (bb >> 16) << 12);
    }
  }
  // Thread = (quad of 4 dst columns, row phase): the quad's column coefficients live in registers for the whole level and
  // the thread walks down its share of the band's rows.  (A per-item table in LDS -- 64 bytes of coefficients per four
  // output bytes -- made the LDS pipe the kernel's whole run time.)  The coefficients of level t+1 are requested before
  // the rows of level t are computed: no global-memory latency between the levels.
  uint4 cnext[4];
  auto fetch_coefs = [&](int t) {
    const LevelDev D = g.lv[tp.lA + t];
    const int nQuads = (D.w + 3) >> 2;
    const int ph = (int)(((float)tid + 0.5f) * __builtin_amdgcn_rcpf((float)nQuads)), q = tid - ph * nQuads;
#pragma unroll
    for (int j = 0; j < 4; j++) cnext[j] = xtab[D.xcoef + 4 * q + j];  // (padded to whole 256-column blocks: always valid)
  };
  fetch_coefs(0);
  RT_MK();
  for (int t = 0; t < tp.nT; t++) {
    const int l = tp.lA + t;
    const LevelDev D = g.lv[l];
    const int dFirst = tb[t + 1].first, dLast = tb[t + 1].last, ownEnd = tb[t + 1].ownEnd;
    const int nQuads = (D.w + 3) >> 2, nRows = dLast - dFirst + 1;
    const int so = (t & 1) ? tp.tileBytes[0] : 0, dofs = (t & 1) ? 0 : tp.tileBytes[0];  // source / destination tile
    const int dp = tp.pitch[t + 1];
    const uint4* rt = rtab + tp.rtOff[t];
    const int nPh = kTailNT / nQuads;                     // >= 1 (build_tail_plan)
    const int ph = (int)(((float)tid + 0.5f) * __builtin_amdgcn_rcpf((float)nQuads)), q = tid - ph * nQuads;
    int cofs[4];
    uint32_t sel[4], coef[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      cofs[j] = so + (int)cnext[j].y;
      sel[j] = cnext[j].z;
      coef[j] = cnext[j].w;
    }
    if (t + 1 < tp.nT) fetch_coefs(t + 1);
    RT_MK();
    __syncthreads();  // (also: the source tile is complete)
    RT_MK();
    uint8_t* gdst = p.pyr + (long long)img * g.pyrImg + D.off;
    const bool keep = t + 1 < tp.nT;
    if (ph < nPh) {
      const int per = (nRows + nPh - 1) / nPh;
      const int rBeg = ph * per, rEnd = min(rBeg + per, nRows);
      // horizontal pass of ONE source row for this quad: t = S[sx]*a0 + S[sx+1]*a1, kept as t & ~15 (the vertical pass
      // uses t >> 4).  Consecutive dst rows share a source row (sy advances by 1 or 2): the last two rows stay in registers.
      uint32_t HA[4], HB[4];
      int offA = -1, offB = -1;
      auto hrow = [&](int off, uint32_t (&H)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint2 pr = lds_pair(smem, cofs[j] + off);
          H[j] = udot2_u16(__builtin_amdgcn_perm(pr.y, pr.x, sel[j]), coef[j], 0u) & 0xFFFF0u;
        }
      };
      for (int r = rBeg; r < rEnd; r++) {
        const uint4 re = rt[r];
        uint32_t na[4], nb[4];
        if ((int)re.x == offA) {
#pragma unroll
          for (int j = 0; j < 4; j++) na[j] = HA[j];
        } else if ((int)re.x == offB) {
#pragma unroll
          for (int j = 0; j < 4; j++) na[j] = HB[j];
        } else {
          hrow((int)re.x, na);
        }
        if (re.y == re.x) {
#pragma unroll
          for (int j = 0; j < 4; j++) nb[j] = na[j];
        } else if ((int)re.y == offB) {
#pragma unroll
          for (int j = 0; j < 4; j++) nb[j] = HB[j];
        } else {
          hrow((int)re.y, nb);
        }
        // (b * (t >> 4)) >> 16 == mulhi(b << 12, t & ~15): b <= 2048, t < 2^20.  The sum + 2 has 10 bits, so its bits 2..9
        // ARE the output byte: two sums per dword, one shift per pair, one v_perm picks bytes 0 and 2 of both.
        uint32_t sm[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          HA[j] = na[j];
          HB[j] = nb[j];
          sm[j] = __umulhi(re.z, na[j]) + __umulhi(re.w, nb[j]) + 2u;
        }
        const uint32_t s01 = ((sm[1] << 16) | sm[0]) >> 2, s23 = ((sm[3] << 16) | sm[2]) >> 2;
        const uint32_t outw = __builtin_amdgcn_perm(s23, s01, 0x06040200u);
        offA = (int)re.x;
        offB = (int)re.y;
        if (keep) *reinterpret_cast<uint32_t*>(smem + dofs + r * dp + 4 * q) = outw;
        if (dFirst + r < ownEnd) *reinterpret_cast<uint32_t*>(gdst + (long long)(dFirst + r) * D.pitch + 4 * q) = outw;
      }
    }
    RT_MK();
    __syncthreads();  // the next level reads the destination tile and overwrites rtab / the old source tile
    RT_MK();
  }
#ifdef RT_PROF
  if (tid == 0 && img == 0 && band == 1) {  // load + tables | per level: {barrier, rows, barrier}
    printf("k_resize_tail L%d..%d band 1:", tp.lA, tp.lA + tp.nT - 1);
    for (int i = 1; i < nq_; i++) printf(" %d", (int)(tq[i] - tq[i - 1]));
    printf("  (x10 ns; workgroup %d)\n", (int)(tq[nq_ - 1] - tq[0]));
  }
#endif
}

hipError_t launch_resize_tail(const Geom& g, const Pyr& p, const TailPlan& tp, const TailBand* bands, int img0, int nimg,
                              const uint4* xtab, const int* yofs, const short* yab, hipStream_t s) {
  hipLaunchKernelGGL(k_resize_tail, dim3(tp.nBands, nimg), dim3(kTailNT), tp.ldsBytes, s, g, p, tp, bands, xtab, yofs, yab, img0);
  return hipGetLastError();
}

// The limit is per device and shared by every handle on it: it is set to the largest plan build_tail_plans can produce
// (kTailLdsMax) and never lowered -- a second handle configuring a smaller geometry used to shrink it under the first.
hipError_t prepare_resize_tail(unsigned ldsBytes) {
  constexpr unsigned kTailLdsCeil = 156 * 1024;  // == kLatLdsMax >= kTailLdsMax (orbx_api.hip)
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_resize_tail), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)(ldsBytes > kTailLdsCeil ? ldsBytes : kTailLdsCeil));
}

}  // namespace orbx
