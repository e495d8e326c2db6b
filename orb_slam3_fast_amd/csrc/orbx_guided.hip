// orbx_guided.hip — grid-guided matchers: ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:618-764) and the two
// SearchByProjection flavours (:41-221, :1594-1806), pinhole and stereo-fisheye.
#include <algorithm>

#include "orbx_device.h"

namespace orbx {

// ================================================================================================ search init
// Frame grid (64 x 48, PosInGrid rounds to the nearest cell, src/Frame.cc:833-844) as CSR lists with
// ascending keypoint indices.
__device__ __forceinline__ int grid_cell(const orbx_keypoint& k, const InitArgs& a) {
  const int px = (int)roundf(__fmul_rn(__fsub_rn(k.x, a.minX), a.invW));
  const int py = (int)roundf(__fmul_rn(__fsub_rn(k.y, a.minY), a.invH));
  if (px < 0 || px >= 64 || py < 0 || py >= 48) return -1;
  return px * 48 + py;
}

// flags[kProjChanged + r] = round r of a fixed-point resolve changed a choice (the flag arrays hold 40 + 48 entries)
constexpr int kProjChanged = 40;
// flags[kProjLast] = which of the two alternating claim / write buffers the LAST EXECUTED round of a fixed-point resolve filled
// (SearchForInitialization, stereo-fisheye SearchByProjection): rounds enqueued behind the fixed point return at once (round 5),
// so the finishing kernels take the buffer from here, not from the number of rounds the host enqueued
constexpr int kProjLast = 34;
constexpr int kGridThreads = 1024;
// Kernel-argument views (round 4): the one-shot entry points pass their argument block by value (kernel arguments); the batched
// SearchByProjection entries launch every kernel ONCE for all frames of an extraction batch with blockIdx.y = frame and the
// frames' argument blocks in a device array.  The kernel bodies are the same code.
struct GridVal {
  InitArgs v;
  __device__ __forceinline__ const InitArgs& get() const { return v; }
};
struct GridOfProj {  // the frame grid of frame blockIdx.y
  const ProjArgs* p;
  __device__ __forceinline__ const InitArgs& get() const { return p[blockIdx.y].grid; }
};
struct ScanOfProj {  // k_init_scan over frame blockIdx.y's candidate counts
  const ProjArgs* p;
  __device__ __forceinline__ InitArgs get() const {
    const ProjArgs& a = p[blockIdx.y];
    InitArgs sc = a.grid;
    sc.candOff = a.candOff;
    sc.n1 = a.nmp;
    sc.candCap = a.candCap;
    return sc;
  }
};
struct InitOfArr {  // SearchForInitialization on the frames of a batch: frame blockIdx.y's argument block (round 5)
  const InitArgs* p;
  __device__ __forceinline__ const InitArgs& get() const { return p[blockIdx.y]; }
};
template <bool B> struct ProjRef;
template <> struct ProjRef<false> {
  ProjArgs v;
  __device__ __forceinline__ const ProjArgs& get() const { return v; }
};
template <> struct ProjRef<true> {
  const ProjArgs* p;
  __device__ __forceinline__ const ProjArgs& get() const { return p[blockIdx.y]; }
};

template <class R>
__global__ __launch_bounds__(kGridThreads) void k_init_grid(R ar) {  // single block (per frame)
  const InitArgs& a = ar.get();
  // Counting sort of the keypoints by grid cell, ascending keypoint index inside a cell (mGrid[i][j].push_back order,
  // src/Frame.cc:536-546): count -> block scan -> unordered atomic fill -> per-cell insertion sort of the short lists.
  // One workgroup, pure latency: 1024 threads so that the keypoints are read in one or two trips, and the cell of a
  // keypoint is kept in a register between the counting and the filling pass.
  __shared__ int cnt[64 * 48];
  __shared__ int wsum[kGridThreads / 64];
  constexpr int kCells = 64 * 48, kPer = kCells / kGridThreads;  // 3 consecutive cells per thread
  constexpr int kKeep = 4;                                       // cached cells per thread (n2 <= 4096)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int c = tid; c < kCells; c += kGridThreads) cnt[c] = 0;
  __syncthreads();
  int mycell[kKeep];
#pragma unroll
  for (int j = 0; j < kKeep; j++) {
    const int i = tid + j * kGridThreads;
    mycell[j] = i < a.n2 ? grid_cell(a.k2[i], a) : -1;
  }
#pragma unroll
  for (int j = 0; j < kKeep; j++)
    if (mycell[j] >= 0) atomicAdd(&cnt[mycell[j]], 1);
  for (int i = tid + kKeep * kGridThreads; i < a.n2; i += kGridThreads) {
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
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wv; w++) run += wsum[w];
#pragma unroll
  for (int k = 0; k < kPer; k++) {
    a.cellStart[tid * kPer + k] = run;
    cnt[tid * kPer + k] = run;  // becomes the fill cursor of the cell
    run += local[k];
  }
  if (tid == kGridThreads - 1) a.cellStart[kCells] = run;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kKeep; j++)
    if (mycell[j] >= 0) a.cellItems[atomicAdd(&cnt[mycell[j]], 1)] = tid + j * kGridThreads;
  for (int i = tid + kKeep * kGridThreads; i < a.n2; i += kGridThreads) {
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
  for (int i = tid; i < a.n2; i += kGridThreads) {
    a.matchedDist[i] = 0x7FFFFFFF;
    a.matches21[i] = -1;
  }
  for (int i = tid; i < a.n1; i += kGridThreads) a.matches12[i] = -1;
  if (tid == 0) {
    a.result[0] = 0;
    a.result[1] = 0;
  }
}

// GetFeaturesInArea(x, y, r, 0, 0) (src/Frame.cc:765-831) for one level-0 keypoint of F1 per wave, in the
// reference's candidate order (ix outer, iy inner, in-cell order).  pass 0 counts, pass 1 writes (i2, dist).
template <class R>
__global__ __launch_bounds__(256) void k_init_cands(R ar, int pass) {
  const InitArgs& a = ar.get();
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
      // the cells (ix, cy0..cy1) of a grid column are neighbours in the CSR (cell = ix * 48 + iy), so a column is ONE
      // contiguous item range, already in the reference's order (iy inner, in-cell order): a trip per 64 items of a
      // column instead of a trip per cell (cells hold ~0.5 keypoints; a 100 px window spans ~120 of them)
      for (int ix = cx0; ix <= cx1; ix++) {
        const int b = a.cellStart[ix * 48 + cy0], e = a.cellStart[ix * 48 + cy1 + 1];
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

template <class R>
__global__ __launch_bounds__(256) void k_init_scan(R ar) {  // single block: exclusive scan of candOff
  const InitArgs& a = ar.get();
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
  // offsets are clamped to the capacity of the candidate arrays: when the lists do not fit, every later kernel sees
  // truncated (never out-of-range) lists, result[1] reports the size that was needed and the host repeats the call
  int run = tid ? tsum[tid - 1] : 0;
  for (int i = b; i < e; i++) {
    const int t = a.candOff[i];
    a.candOff[i] = min(run, a.candCap);
    run += t;
  }
  if (tid == 255) {
    a.candOff[n] = min(tsum[255], a.candCap);
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
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
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
    for (int ix = cx0; ix <= cx1; ix++) {  // a grid column is one contiguous CSR range (see k_init_cands)
      const int b = a.cellStart[ix * 48 + cy0], e = a.cellStart[ix * 48 + cy1 + 1];
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
  hipLaunchKernelGGL(k_init_grid<GridVal>, dim3(1), dim3(kGridThreads), 0, s, GridVal{a});
  return hipGetLastError();
}
// ---- ORBmatcher::Fuse, the search of the loop body (src/ORBmatcher.cc:1195-1256) -----------------------------------------------
// Every map point is independent (the Replace / AddObservation bookkeeping after a hit does not feed back into the search).
// 16 lanes per point walk the window's grid columns (a column of cells is one contiguous CSR range, already in the reference's
// ix-outer / iy-inner / in-cell order); "first strict minimum" = minimum of (distance, position in that order).
__global__ __launch_bounds__(256) void k_fuse_search(FuseArgs a) {
  const int sub = threadIdx.x & 15;
  const int ip = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (ip >= a.npts) return;
  const InitArgs& g = a.grid;
  const orbx_fuse_point& p = a.pts[ip];
  uint64_t best = ~0ull;  // (dist << 52) | (position << 32) | keypoint index
  if (p.valid) {
    const float x = p.u, y = p.v, r = p.radius, ur = p.ur;
    const int lvl = p.predicted_level;
    const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int cx1 = min(63, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW)));
    const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH)));
    const int cy1 = min(47, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH)));
    if (cx0 < 64 && cx1 >= 0 && cy0 < 48 && cy1 >= 0) {
      uint32_t d1[8];
#pragma unroll
      for (int i = 0; i < 8; i++) d1[i] = reinterpret_cast<const uint32_t*>(p.desc)[i];
      int seen = 0;
      for (int ix = cx0; ix <= cx1; ix++) {
        const int b = g.cellStart[ix * 48 + cy0], e = g.cellStart[ix * 48 + cy1 + 1];
        for (int j = b + sub; j < e; j += 16) {
          const int idx = g.cellItems[j];
          const orbx_keypoint kp = g.k2[idx];
          if (!(fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r)) continue;  // KeyFrame.cc:739-743
          if (kp.octave < lvl - 1 || kp.octave > lvl) continue;                             // :1215
          const float ex = __fsub_rn(x, kp.x), ey = __fsub_rn(y, kp.y);
          float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
          const float kr = a.uRight ? a.uRight[idx] : -1.0f;
          double lim = 5.99;
          if (kr >= 0) {  // stereo observation: three residuals, chi-square with 3 dof (:1217-1227)
            const float er = __fsub_rn(ur, kr);
            e2 = __fadd_rn(e2, __fmul_rn(er, er));
            lim = 7.8;
          }
          if ((double)__fmul_rn(e2, a.invSigma2[kp.octave]) > lim) continue;
          const int dist = hamming256(d1, a.desc + (long long)idx * 8);
          const uint64_t key = ((uint64_t)(uint32_t)dist << 52) | ((uint64_t)(uint32_t)(seen + (j - b)) << 32) | (uint32_t)idx;
          best = key < best ? key : best;
        }
        seen += e - b;
      }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const uint64_t ob = __shfl_xor((unsigned long long)best, o, 16);
    best = ob < best ? ob : best;
  }
  bool hit = false;
  if (sub == 0) {
    const int dist = best == ~0ull ? 256 : (int)(best >> 52);
    a.bestDist[ip] = dist;
    hit = dist <= a.maxDist;  // TH_LOW (Fuse, :1258) or TH_HIGH (SearchBySim3, :1494)
    a.bestIdx[ip] = hit ? (int)(uint32_t)best : -1;
  }
  const uint64_t m = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.result[0], __popcll(m));
}

hipError_t launch_fuse_search(const FuseArgs& a, hipStream_t s) {
  hipError_t e = launch_grid_build(a.grid, s);
  if (e != hipSuccess) return e;
  if (a.npts > 0) hipLaunchKernelGGL(k_fuse_search, dim3((a.npts + 15) / 16), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_area_query(const InitArgs& a, const float* q, int nq, int* qOff, int* out, int pass, hipStream_t s) {
  if (nq > 0) hipLaunchKernelGGL(k_area_query, dim3((nq + 3) / 4), dim3(256), 0, s, a, q, nq, qOff, out, pass);
  return hipGetLastError();
}
hipError_t launch_scan_offsets(const InitArgs& a, hipStream_t s) {  // exclusive scan of a.candOff[0..n1] (n1 = #queries)
  hipLaunchKernelGGL(k_init_scan<GridVal>, dim3(1), dim3(256), 0, s, GridVal{a});
  return hipGetLastError();
}

hipError_t launch_search_init(const InitArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_grid<GridVal>, dim3(1), dim3(kGridThreads), 0, s, GridVal{a});
  if (a.n1 > 0) {
    hipLaunchKernelGGL(k_init_cands<GridVal>, dim3((a.n1 + 3) / 4), dim3(256), 0, s, GridVal{a}, 0);
    hipLaunchKernelGGL(k_init_scan<GridVal>, dim3(1), dim3(256), 0, s, GridVal{a});
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

template <class R>
__global__ __launch_bounds__(256) void k_init_round(R ar, int round_no) {
  const InitArgs& a = ar.get();
  const int lane = threadIdx.x & 63;
  const int i1 = blockIdx.x * 4 + (threadIdx.x >> 6);
  // claimer lists rotate through three buffers (read [r % 3], append to [(r + 1) % 3], clear [(r + 2) % 3] for the next
  // round): one launch per round; claim[] is rewritten completely every round and alternates between two
  const int prev = round_no % 3, next = (round_no + 1) % 3, c2 = round_no & 1;
  if (round_no > 0 && a.flags[kProjChanged + round_no - 1] == 0) return;  // the previous round changed nothing: fixed point reached
  if (blockIdx.x == 0 && threadIdx.x == 0) a.flags[kProjLast] = c2 ^ 1;
  {
    int* clr = a.nclaimers[(round_no + 2) % 3];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n2; i += gridDim.x * 256) clr[i] = 0;
  }
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
    const int2 o = a.claim[c2][i1];
    if (round_no == 0 || o.x != cl.x || o.y != cl.y) a.flags[kProjChanged + round_no] = 1;
    a.claim[c2 ^ 1][i1] = cl;
    if (cl.x >= 0) {
      const int pos = atomicAdd(&a.nclaimers[next][cl.x], 1);
      if (pos < kFeWriters) a.claimers[next][cl.x * kFeWriters + pos] = make_int2(i1, cl.y);
      else a.flags[1] = 1;
    }
  }
}

template <class R>
__global__ __launch_bounds__(256) void k_init_reset(R ar) {  // before round 0: empty claimer lists, no owners
  const InitArgs& a = ar.get();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n2; i += gridDim.x * 256) {
    a.nclaimers[1][i] = 0;
    a.matches21[i] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x < kProjChanged + 48) a.flags[threadIdx.x] = 0;
}

__device__ __forceinline__ int init_bin(const InitArgs& a, int i1, int i2) {
  float rot = __fsub_rn(a.k1[i1].angle, a.k2[i2].angle);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
  if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
  return bin;
}

template <class R>
__global__ __launch_bounds__(256) void k_init_owner(R ar, int) {  // vnMatches21 = the last claimer; votes
  const InitArgs& a = ar.get();
  const int last = a.flags[kProjLast];
  const int i1 = blockIdx.x * 256 + threadIdx.x;
  if (i1 >= a.n1) return;
  const int2 cl = a.claim[last][i1];
  if (cl.x < 0) return;
  atomicMax(&a.matches21[cl.x], i1);
  if (a.checkOri) atomicAdd(&a.flags[4 + init_bin(a, i1, cl.x)], 1);  // stolen matches stay in rotHist (:712-719)
}

template <class R>
__global__ __launch_bounds__(256) void k_init_finish(R ar, int) {
  const InitArgs& a = ar.get();
  const int last = a.flags[kProjLast];
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

template <class R>
__global__ void k_init_result(R ar) {
  const InitArgs& a = ar.get();
  a.result[0] = a.flags[2];
}

hipError_t launch_search_init_cands_fill(const InitArgs& a, hipStream_t s) {
  if (a.n1 > 0) hipLaunchKernelGGL(k_init_cands<GridVal>, dim3((a.n1 + 3) / 4), dim3(256), 0, s, GridVal{a}, 1);
  return hipGetLastError();
}
hipError_t launch_search_init_rounds(const InitArgs& a, int first_round, int rounds, hipStream_t s) {
  const int gb = (a.n2 + 255) / 256 > 0 ? (a.n2 + 255) / 256 : 1;
  if (first_round == 0) hipLaunchKernelGGL(k_init_reset<GridVal>, dim3(gb), dim3(256), 0, s, GridVal{a});
  for (int r = first_round; r < first_round + rounds; r++)
    hipLaunchKernelGGL(k_init_round<GridVal>, dim3((a.n1 + 3) / 4), dim3(256), 0, s, GridVal{a}, r);
  return hipGetLastError();
}
hipError_t launch_search_init_finish(const InitArgs& a, int last_round, hipStream_t s) {
  const int last = (last_round & 1) ^ 1;
  hipLaunchKernelGGL(k_init_owner<GridVal>, dim3((a.n1 + 255) / 256), dim3(256), 0, s, GridVal{a}, last);
  hipLaunchKernelGGL(k_init_finish<GridVal>, dim3((a.n1 + 255) / 256), dim3(256), 0, s, GridVal{a}, last);
  hipLaunchKernelGGL(k_init_result<GridVal>, dim3(1), dim3(1), 0, s, GridVal{a});
  return hipGetLastError();
}
// SearchForInitialization for every frame of a batch, one launch per kernel (blockIdx.y = frame): frame grid, candidate counts,
// scan, candidate fill, `rounds` blind fixed-point rounds (no convergence flag is read in between: flags[kProjChanged + r] of
// every frame travel back with the results), last-claimer ownership + rotation votes, cull + vbPrevMatched update, count.
// d_frames: device array of nFrames argument blocks; maxN1 / maxN2: the largest keypoint counts of F1 / F2.
hipError_t launch_search_init_batch(const InitArgs* d_frames, int nFrames, int maxN1, int maxN2, int rounds, hipStream_t s) {
  if (nFrames <= 0) return hipSuccess;
  const dim3 one(1, nFrames), n1w((std::max(maxN1, 1) + 3) / 4, nFrames), n1t((std::max(maxN1, 1) + 255) / 256, nFrames),
      n2t((std::max(maxN2, 1) + 255) / 256, nFrames);
  const InitOfArr r{d_frames};
  hipLaunchKernelGGL(k_init_grid<InitOfArr>, one, dim3(kGridThreads), 0, s, r);
  hipLaunchKernelGGL(k_init_cands<InitOfArr>, n1w, dim3(256), 0, s, r, 0);
  hipLaunchKernelGGL(k_init_scan<InitOfArr>, one, dim3(256), 0, s, r);
  hipLaunchKernelGGL(k_init_cands<InitOfArr>, n1w, dim3(256), 0, s, r, 1);
  hipLaunchKernelGGL(k_init_reset<InitOfArr>, n2t, dim3(256), 0, s, r);
  for (int i = 0; i < rounds; i++) hipLaunchKernelGGL(k_init_round<InitOfArr>, n1w, dim3(256), 0, s, r, i);
  const int last = ((rounds - 1) & 1) ^ 1;
  hipLaunchKernelGGL(k_init_owner<InitOfArr>, n1t, dim3(256), 0, s, r, last);
  hipLaunchKernelGGL(k_init_finish<InitOfArr>, n1t, dim3(256), 0, s, r, last);
  hipLaunchKernelGGL(k_init_result<InitOfArr>, one, dim3(1), 0, s, r);
  return hipGetLastError();
}
hipError_t launch_search_init_resolve_serial(const InitArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_resolve, dim3(1), dim3(64), (size_t)((a.n1 + 15) & ~15) + 16, s, a);
  return hipGetLastError();
}

hipError_t launch_search_init_fill(const InitArgs& a, hipStream_t s) {
  if (a.n1 > 0) hipLaunchKernelGGL(k_init_cands<GridVal>, dim3((a.n1 + 3) / 4), dim3(256), 0, s, GridVal{a}, 1);
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

template <bool B>
__global__ __launch_bounds__(256) void k_proj_cands(ProjRef<B> ar, int pass) {
  const ProjArgs& a = ar.get();
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
      for (int ix = cx0; ix <= cx1; ix++) {  // a grid column is one contiguous CSR range (see k_init_cands)
        const int b = g.cellStart[ix * 48 + cy0], e = g.cellStart[ix * 48 + cy1 + 1];
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

template <bool B>
__global__ __launch_bounds__(64) void k_proj_resolve(ProjRef<B> ar) {
  const ProjArgs& a = ar.get();
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
    if (bestDist <= a.maxDist) {  // TH_HIGH, or ORBdist (:1886)
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
        a.occupied[bestIdx] = a.mode == 0 ? a.mps[im].has_observations : (uint8_t)(a.claimAll | a.pts[im].has_observations);
        if (a.mode == 1 && a.checkOri) {
          float rot = __fsub_rn(a.pts[im].angle, a.grid.k2[bestIdx].angle);
          if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
          int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
          if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
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
        if (a.claimAll) a.occupied[idx] = 0;
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
template <bool B>
__global__ __launch_bounds__(256) void k_proj_round(ProjRef<B> ar, int round_no) {
  const ProjArgs& a = ar.get();
  const int lane = threadIdx.x & 63;
  const int im = blockIdx.x * 4 + (threadIdx.x >> 6);
  // the host enqueues a fixed number of rounds without looking: once a round has changed nothing the fixed point is reached
  // and every later round returns at once (its changed[] entry stays 0)
  if (round_no > 0 && a.flags[kProjChanged + round_no - 1] == 0) return;
  const int* takerPrev = a.taker[round_no % 3];
  int* takerNew = a.taker[(round_no + 1) % 3];
  {  // the buffer the NEXT round writes is cleared here (nobody touches it in this round): one launch per round
    int* takerClr = a.taker[(round_no + 2) % 3];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.grid.n2; i += gridDim.x * 256) takerClr[i] = 0x7FFFFFFF;
  }
  if (im >= a.nmp) return;
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
    if (bestDist <= a.maxDist) {  // TH_HIGH, or ORBdist (:1886)
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
    if (round_no == 0 || a.choice[im] != chosen) a.flags[kProjChanged + round_no] = 1;
    a.choice[im] = chosen;
    if (chosen >= 0 && (a.mode == 0 ? a.mps[im].has_observations : (a.claimAll | a.pts[im].has_observations)))
      atomicMin(&takerNew[chosen], im);
  }
}

template <bool B>
__global__ __launch_bounds__(256) void k_proj_reset(ProjRef<B> ar) {
  const ProjArgs& a = ar.get();  // before round 0: taker[1] = +inf, no matches, flags 0
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.grid.n2; i += gridDim.x * 256) {
    a.taker[1][i] = 0x7FFFFFFF;
    a.match[i] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x < kProjChanged + 48) a.flags[threadIdx.x] = 0;  // accepted, removed, histogram, changed[]
}

// After convergence: match[k] = the LAST point that chose k (later assignments overwrite), occupied[k] = that point's
// observation flag, orientation histogram of the accepted pairs (mode 1).
template <bool B>
__global__ __launch_bounds__(256) void k_proj_assign(ProjRef<B> ar) {
  const ProjArgs& a = ar.get();
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
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
        atomicAdd(&a.flags[3 + bin], 1);
      }
    }
  }
  const uint64_t m = __ballot(acc);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.flags[1], __popcll(m));
}

template <bool B>
__global__ __launch_bounds__(256) void k_proj_cull(ProjRef<B> ar) {
  const ProjArgs& a = ar.get();
  // occupied: set by the last chooser (a keypoint whose first chooser has observations has no later chooser)
  for (int k = blockIdx.x * 256 + threadIdx.x; k < a.grid.n2; k += gridDim.x * 256) {
    const int im = a.match[k];
    if (im >= 0) a.occupied[k] = a.mode == 0 ? a.mps[im].has_observations : (uint8_t)(a.claimAll | a.pts[im].has_observations);
  }
}

template <bool B>
__global__ __launch_bounds__(256) void k_proj_cull2(ProjRef<B> ar) {
  const ProjArgs& a = ar.get();  // rotation-consistency cull (:1780-1800, :1920-1955)
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
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
      if (bin != ind1 && bin != ind2 && bin != ind3) {
        a.match[k] = -1;  // CurrentFrame.mvpMapPoints[idx] = NULL, even if a later point re-took the slot
        if (a.claimAll) a.occupied[k] = 0;
        rem = true;
      }
    }
  }
  const uint64_t m = __ballot(rem);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(&a.flags[2], __popcll(m));
}

template <bool B>
__global__ void k_proj_result(ProjRef<B> ar) {
  const ProjArgs& a = ar.get();
  a.result[0] = a.flags[1] - a.flags[2];
}

hipError_t launch_proj_cands_fill(const ProjArgs& a, hipStream_t s) {
  if (a.nmp > 0) hipLaunchKernelGGL(k_proj_cands<false>, dim3((a.nmp + 3) / 4), dim3(256), 0, s, ProjRef<false>{a}, 1);
  return hipGetLastError();
}
hipError_t launch_proj_rounds(const ProjArgs& a, int first_round, int rounds, hipStream_t s) {
  if (a.nmp <= 0) return hipSuccess;
  if (first_round == 0) hipLaunchKernelGGL(k_proj_reset<false>, dim3((a.grid.n2 + 255) / 256), dim3(256), 0, s, ProjRef<false>{a});
  for (int r = first_round; r < first_round + rounds; r++)
    hipLaunchKernelGGL(k_proj_round<false>, dim3((a.nmp + 3) / 4), dim3(256), 0, s, ProjRef<false>{a}, r);
  return hipGetLastError();
}
hipError_t launch_proj_finish(const ProjArgs& a, int last_round, hipStream_t s) {
  (void)last_round;
  if (a.nmp > 0) {
    hipLaunchKernelGGL(k_proj_assign<false>, dim3((a.nmp + 255) / 256), dim3(256), 0, s, ProjRef<false>{a});
    hipLaunchKernelGGL(k_proj_cull<false>, dim3((a.grid.n2 + 255) / 256), dim3(256), 0, s, ProjRef<false>{a});
    if (a.mode == 1 && a.checkOri) hipLaunchKernelGGL(k_proj_cull2<false>, dim3((a.nmp + 255) / 256), dim3(256), 0, s, ProjRef<false>{a});
  }
  hipLaunchKernelGGL(k_proj_result<false>, dim3(1), dim3(1), 0, s, ProjRef<false>{a});
  return hipGetLastError();
}
hipError_t launch_proj_resolve_serial(const ProjArgs& a, hipStream_t s) {
  const size_t lds = (a.mode == 1 && a.checkOri) ? (size_t)(a.nmp + 4) * 4 : 16;
  hipLaunchKernelGGL(k_proj_resolve<false>, dim3(1), dim3(64), lds, s, ProjRef<false>{a});
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
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
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

// Kernel-argument views of the stereo-fisheye resolve (round 5): by value for the one-shot calls, frame blockIdx.y of a device
// array for orbx_search_by_projection[_frame]_fisheye_batch.
struct FeVal {
  ProjFeArgs v;
  __device__ __forceinline__ const ProjFeArgs& get() const { return v; }
};
struct FeOfArr {
  const ProjFeArgs* p;
  __device__ __forceinline__ const ProjFeArgs& get() const { return p[blockIdx.y]; }
};
template <class R>
__global__ __launch_bounds__(256) void k_proj_round_fe(R ar, int round_no) {
  const ProjFeArgs& a = ar.get();
  const int lane = threadIdx.x & 63;
  const int im = blockIdx.x * 4 + (threadIdx.x >> 6);
  // writer lists rotate through three buffers: round r reads [r % 3], appends to [(r + 1) % 3] and clears the counts of
  // [(r + 2) % 3] for the next round (nobody touches that one now) -- one launch per round; writes[] (fully rewritten every
  // round) alternates between two
  const int prev = round_no % 3, next = (round_no + 1) % 3, w2 = round_no & 1;
  if (round_no > 0 && a.flags[kProjChanged + round_no - 1] == 0) return;  // the previous round changed nothing: fixed point reached
  if (blockIdx.x == 0 && threadIdx.x == 0) a.flags[kProjLast] = w2 ^ 1;
  {
    int* clr = a.nwriters[(round_no + 2) % 3];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) clr[i] = 0;
  }
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
    const int4 o = a.writes[w2][im];
    if (round_no == 0 || o.x != w.x || o.y != w.y || o.z != w.z || o.w != w.w) a.flags[kProjChanged + round_no] = 1;
    a.writes[w2 ^ 1][im] = w;
    const int ws[4] = {w.x, w.y, w.z, w.w};
    for (int t = 0; t < 4; t++) {
      if (ws[t] < 0) continue;
      bool dup = false;
      for (int u = 0; u < t; u++) dup = dup || ws[u] == ws[t];
      if (dup) continue;
      const int pos = atomicAdd(&a.nwriters[next][ws[t]], 1);
      if (pos < kFeWriters) a.writers[next][ws[t] * kFeWriters + pos] = im;
      else a.flags[1] = 1;
    }
  }
}

template <class R>
__global__ __launch_bounds__(256) void k_proj_reset_fe(R ar) {  // before round 0: empty writer lists, no matches
  const ProjFeArgs& a = ar.get();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n; i += gridDim.x * 256) {
    a.nwriters[1][i] = 0;
    a.match[i] = -1;
  }
  if (blockIdx.x == 0 && threadIdx.x < kProjChanged + 48) a.flags[threadIdx.x] = 0;  // overflow, counts, histogram, changed[]
}

__device__ __forceinline__ int fe_bin(const ProjFeArgs& a, int im, int slot) {
  float rot = __fsub_rn(a.pts[im].angle, a.kps[slot].angle);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
  if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
  return bin;
}

template <class R>
__global__ __launch_bounds__(256) void k_proj_assign_fe(R ar, int) {  // last writer wins every slot
  const ProjFeArgs& a = ar.get();
  const int last = a.flags[kProjLast];
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

template <class R>
__global__ __launch_bounds__(256) void k_proj_occ_fe(R ar) {
  const ProjFeArgs& a = ar.get();
  for (int k = blockIdx.x * 256 + threadIdx.x; k < a.n; k += gridDim.x * 256) {
    const int im = a.match[k];
    if (im >= 0) a.occupied[k] = a.mode == 0 ? a.mps[im].has_observations : a.pts[im].has_observations;
  }
}

template <class R>
__global__ __launch_bounds__(256) void k_proj_cull_fe(R ar, int) {
  const ProjFeArgs& a = ar.get();
  const int last = a.flags[kProjLast];
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

template <class R>
__global__ void k_proj_result_fe(R ar) {
  const ProjFeArgs& a = ar.get();
  a.result[0] = a.flags[2] - a.flags[3];
}

hipError_t launch_proj_rounds_fisheye(const ProjFeArgs& a, int first_round, int rounds, hipStream_t s) {
  if (a.nmp <= 0) return hipSuccess;
  if (first_round == 0) hipLaunchKernelGGL(k_proj_reset_fe<FeVal>, dim3((a.n + 255) / 256), dim3(256), 0, s, FeVal{a});
  for (int r = first_round; r < first_round + rounds; r++)
    hipLaunchKernelGGL(k_proj_round_fe<FeVal>, dim3((a.nmp + 3) / 4), dim3(256), 0, s, FeVal{a}, r);
  return hipGetLastError();
}
hipError_t launch_proj_finish_fisheye(const ProjFeArgs& a, int last_round, hipStream_t s) {
  const int last = (last_round & 1) ^ 1;  // round r wrote writes[(r & 1) ^ 1]
  hipLaunchKernelGGL(k_proj_assign_fe<FeVal>, dim3((a.nmp + 255) / 256), dim3(256), 0, s, FeVal{a}, last);
  hipLaunchKernelGGL(k_proj_occ_fe<FeVal>, dim3((a.n + 255) / 256), dim3(256), 0, s, FeVal{a});
  if (a.mode == 1 && a.checkOri) hipLaunchKernelGGL(k_proj_cull_fe<FeVal>, dim3((a.nmp + 255) / 256), dim3(256), 0, s, FeVal{a}, last);
  hipLaunchKernelGGL(k_proj_result_fe<FeVal>, dim3(1), dim3(1), 0, s, FeVal{a});
  return hipGetLastError();
}
// The stereo-fisheye SearchByProjection chain for every frame of a batch, one launch per kernel.  d_sides: 2 * nFrames per-camera
// argument blocks (frame f: left = 2 f, right = 2 f + 1) for the grids / candidate lists (the pinhole batch's kernels with
// blockIdx.y = side), d_frames: nFrames resolve blocks; `rounds` blind fixed-point rounds, then assignment / occupancy / cull.
hipError_t launch_proj_fisheye_batch(const ProjArgs* d_sides, const ProjFeArgs* d_frames, int nFrames, int maxPts, int maxN, int mode,
                                     int checkOri, int rounds, hipStream_t s) {
  if (nFrames <= 0) return hipSuccess;
  const int P = std::max(maxPts, 1);
  const dim3 oneS(1, 2 * nFrames), pts4S((P + 3) / 4, 2 * nFrames);
  const ProjRef<true> rs{d_sides};
  hipLaunchKernelGGL(k_init_grid<GridOfProj>, oneS, dim3(kGridThreads), 0, s, GridOfProj{d_sides});
  hipLaunchKernelGGL(k_proj_cands<true>, pts4S, dim3(256), 0, s, rs, 0);
  hipLaunchKernelGGL(k_init_scan<ScanOfProj>, oneS, dim3(256), 0, s, ScanOfProj{d_sides});
  hipLaunchKernelGGL(k_proj_cands<true>, pts4S, dim3(256), 0, s, rs, 1);
  const FeOfArr rf{d_frames};
  const dim3 nT((std::max(maxN, 1) + 255) / 256, nFrames), pts4((P + 3) / 4, nFrames), ptsT((P + 255) / 256, nFrames);
  hipLaunchKernelGGL(k_proj_reset_fe<FeOfArr>, nT, dim3(256), 0, s, rf);
  for (int r = 0; r < rounds; r++) hipLaunchKernelGGL(k_proj_round_fe<FeOfArr>, pts4, dim3(256), 0, s, rf, r);
  const int last = ((rounds - 1) & 1) ^ 1;
  hipLaunchKernelGGL(k_proj_assign_fe<FeOfArr>, ptsT, dim3(256), 0, s, rf, last);
  hipLaunchKernelGGL(k_proj_occ_fe<FeOfArr>, nT, dim3(256), 0, s, rf);
  if (mode == 1 && checkOri) hipLaunchKernelGGL(k_proj_cull_fe<FeOfArr>, ptsT, dim3(256), 0, s, rf, last);
  hipLaunchKernelGGL(k_proj_result_fe<FeOfArr>, dim3(1, nFrames), dim3(1), 0, s, rf);
  return hipGetLastError();
}

hipError_t launch_proj_resolve_fisheye(const ProjFeArgs& a, hipStream_t s) {
  const size_t lds = (a.mode == 1 && a.checkOri) ? (size_t)(2 * a.nmp + 4) * 4 : 16;  // up to two votes per point
  hipLaunchKernelGGL(k_proj_resolve_fe, dim3(1), dim3(64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_proj_count(const ProjArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_init_grid<GridVal>, dim3(1), dim3(kGridThreads), 0, s, GridVal{a.grid});
  if (a.nmp > 0) {
    hipLaunchKernelGGL(k_proj_cands<false>, dim3((a.nmp + 3) / 4), dim3(256), 0, s, ProjRef<false>{a}, 0);
    InitArgs sc = a.grid;  // k_init_scan scans candOff[0 .. n1]
    sc.candOff = a.candOff;
    sc.n1 = a.nmp;
    sc.candCap = a.candCap;
    hipLaunchKernelGGL(k_init_scan<GridVal>, dim3(1), dim3(256), 0, s, GridVal{sc});
  }
  return hipGetLastError();
}
// Every frame of a batch through the whole pinhole SearchByProjection chain, one launch per kernel (blockIdx.y = frame): grid,
// candidate counts, scan, candidate fill, `rounds` blind fixed-point rounds, assignment, occupancy, orientation cull, count.
// d_frames: device array of nFrames argument blocks (same mode / checkOri); maxPts / maxN2: the largest point / keypoint count.
hipError_t launch_proj_batch(const ProjArgs* d_frames, int nFrames, int maxPts, int maxN2, int mode, int checkOri, int rounds,
                             hipStream_t s) {
  if (nFrames <= 0) return hipSuccess;
  const dim3 one(1, nFrames), pts4((std::max(maxPts, 1) + 3) / 4, nFrames), pts256((std::max(maxPts, 1) + 255) / 256, nFrames),
      n2b((std::max(maxN2, 1) + 255) / 256, nFrames);
  const ProjRef<true> r{d_frames};
  hipLaunchKernelGGL(k_init_grid<GridOfProj>, one, dim3(kGridThreads), 0, s, GridOfProj{d_frames});
  hipLaunchKernelGGL(k_proj_cands<true>, pts4, dim3(256), 0, s, r, 0);
  hipLaunchKernelGGL(k_init_scan<ScanOfProj>, one, dim3(256), 0, s, ScanOfProj{d_frames});
  hipLaunchKernelGGL(k_proj_cands<true>, pts4, dim3(256), 0, s, r, 1);
  hipLaunchKernelGGL(k_proj_reset<true>, n2b, dim3(256), 0, s, r);
  for (int i = 0; i < rounds; i++) hipLaunchKernelGGL(k_proj_round<true>, pts4, dim3(256), 0, s, r, i);
  hipLaunchKernelGGL(k_proj_assign<true>, pts256, dim3(256), 0, s, r);
  hipLaunchKernelGGL(k_proj_cull<true>, n2b, dim3(256), 0, s, r);
  if (mode == 1 && checkOri) hipLaunchKernelGGL(k_proj_cull2<true>, pts256, dim3(256), 0, s, r);
  hipLaunchKernelGGL(k_proj_result<true>, one, dim3(1), 0, s, r);
  return hipGetLastError();
}

// The keypoints / descriptors of a two-camera frame as ONE array [left | right] (mvKeys | mvKeysRight, the indexing the resolve
// uses: slot < Nleft = left camera) from the two images of an extraction batch: frame blockIdx.y, words copied by the whole grid.
__global__ __launch_bounds__(256) void k_fe_concat(FeConcatArgs a) {
  const int f = blockIdx.y;
  const int nL = min(max(a.nOut[a.firstL + f], 0), a.cap), nR = min(max(a.nOut[a.firstR + f], 0), a.cap);
  const uint32_t* kL = reinterpret_cast<const uint32_t*>(a.kps + (size_t)(a.firstL + f) * a.cap);
  const uint32_t* kR = reinterpret_cast<const uint32_t*>(a.kps + (size_t)(a.firstR + f) * a.cap);
  const uint32_t* dL = reinterpret_cast<const uint32_t*>(a.desc + (size_t)(a.firstL + f) * a.cap * 32);
  const uint32_t* dR = reinterpret_cast<const uint32_t*>(a.desc + (size_t)(a.firstR + f) * a.cap * 32);
  uint32_t* ko = reinterpret_cast<uint32_t*>(a.kcat + (size_t)f * 2 * a.cap);
  uint32_t* dout = reinterpret_cast<uint32_t*>(a.dcat + (size_t)f * 2 * a.cap * 32);
  const int t0 = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
  for (int i = t0; i < nL * 7; i += step) ko[i] = kL[i];
  for (int i = t0; i < nR * 7; i += step) ko[nL * 7 + i] = kR[i];
  for (int i = t0; i < nL * 8; i += step) dout[i] = dL[i];
  for (int i = t0; i < nR * 8; i += step) dout[nL * 8 + i] = dR[i];
}
hipError_t launch_fe_concat(const FeConcatArgs& a, int nFrames, hipStream_t s) {
  if (nFrames > 0) hipLaunchKernelGGL(k_fe_concat, dim3(16, nFrames), dim3(256), 0, s, a);
  return hipGetLastError();
}

// Frame::isInFrustum (src/Frame.cc:632-690, pinhole case Nleft == -1) + MapPoint::PredictScale (src/MapPoint.cc:559-573) for every
// (local map point, frame of the batch): what Tracking::SearchLocalPoints computes on the host before SearchByProjection reads it
// back from the MapPoint members (src/ORBmatcher.cc:62-76).  One thread per point and frame; float arithmetic in the reference's
// expression order, no contraction (TU flag); the logarithm of PredictScale is the device's logf (the level can differ from a
// glibc build where log(ratio) / logScaleFactor lies within rounding of an integer: tolerance parity, tests/test_projection_device.py).
// Writes the orbx_map_point_view records the matcher kernels consume -- the views stay in HBM.
__global__ __launch_bounds__(256) void k_project_map(MapProjArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
  if (i >= a.n) return;
  const orbx_frame_pose& T = a.poses[f];
  orbx_map_point_view v;
  v.proj_x = -1.f; v.proj_y = -1.f; v.proj_xr = 0.f; v.view_cos = 0.f; v.track_depth = 0.f; v.predicted_level = 0;
  v.in_view = 0;
  const uint8_t fl = a.flags[i];
  v.bad = fl & 1; v.has_observations = (fl >> 1) & 1; v.pad_ = 0;
  const uint4* d4 = reinterpret_cast<const uint4*>(a.desc + (size_t)i * 32);
  const uint4 da = d4[0], db = d4[1];
  const float Px = a.pos[3 * i], Py = a.pos[3 * i + 1], Pz = a.pos[3 * i + 2];
  bool ok = !(a.skip && a.skip[(size_t)f * a.n + i]);
  // Pc = mRcw * P + mtcw
  const float X = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.Rcw[0], Px), __fmul_rn(T.Rcw[1], Py)), __fmul_rn(T.Rcw[2], Pz)), T.tcw[0]);
  const float Y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.Rcw[3], Px), __fmul_rn(T.Rcw[4], Py)), __fmul_rn(T.Rcw[5], Pz)), T.tcw[1]);
  const float Z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.Rcw[6], Px), __fmul_rn(T.Rcw[7], Py)), __fmul_rn(T.Rcw[8], Pz)), T.tcw[2]);
  const float pcDist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(X, X), __fmul_rn(Y, Y)), __fmul_rn(Z, Z)));
  const float invz = __fdiv_rn(1.0f, Z);
  ok = ok && !(Z < 0.0f);
  const float u = __fadd_rn(__fdiv_rn(__fmul_rn(T.fx, X), Z), T.cx), vv = __fadd_rn(__fdiv_rn(__fmul_rn(T.fy, Y), Z), T.cy);
  ok = ok && !(u < a.minX || u > a.maxX) && !(vv < a.minY || vv > a.maxY);
  const float ox = __fsub_rn(Px, T.Ow[0]), oy = __fsub_rn(Py, T.Ow[1]), oz = __fsub_rn(Pz, T.Ow[2]);
  const float dist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(ox, ox), __fmul_rn(oy, oy)), __fmul_rn(oz, oz)));
  const float maxD = __fmul_rn(1.2f, a.maxDist[i]), minD = __fmul_rn(0.8f, a.minDist[i]);
  ok = ok && !(dist < minD || dist > maxD);
  const float viewCos = __fdiv_rn(__fadd_rn(__fadd_rn(__fmul_rn(ox, a.normal[3 * i]), __fmul_rn(oy, a.normal[3 * i + 1])),
                                           __fmul_rn(oz, a.normal[3 * i + 2])), dist);
  ok = ok && !(viewCos < a.viewCosLimit);
  if (ok) {
    const float ratio = __fdiv_rn(a.maxDist[i], dist);
    int lvl = (int)ceilf(__fdiv_rn(logf(ratio), a.logScaleFactor));
    lvl = lvl < 0 ? 0 : (lvl >= a.nlevels ? a.nlevels - 1 : lvl);
    v.in_view = 1;
    v.proj_x = u;
    v.proj_y = vv;
    v.proj_xr = __fsub_rn(u, __fmul_rn(T.bf, invz));
    v.track_depth = pcDist;
    v.view_cos = viewCos;
    v.predicted_level = lvl;
  }
  uint32_t* o = reinterpret_cast<uint32_t*>(a.views + (size_t)f * a.n + i);   // 60-byte records: 15 dwords
  o[0] = __builtin_bit_cast(uint32_t, v.proj_x); o[1] = __builtin_bit_cast(uint32_t, v.proj_y); o[2] = __builtin_bit_cast(uint32_t, v.proj_xr);
  o[3] = __builtin_bit_cast(uint32_t, v.view_cos); o[4] = __builtin_bit_cast(uint32_t, v.track_depth); o[5] = (uint32_t)v.predicted_level;
  o[6] = (uint32_t)v.in_view | ((uint32_t)v.bad << 8) | ((uint32_t)v.has_observations << 16);
  o[7] = da.x; o[8] = da.y; o[9] = da.z; o[10] = da.w; o[11] = db.x; o[12] = db.y; o[13] = db.z; o[14] = db.w;
}
hipError_t launch_project_map(const MapProjArgs& a, int nFrames, hipStream_t s) {
  if (a.n > 0 && nFrames > 0) hipLaunchKernelGGL(k_project_map, dim3((a.n + 255) / 256, nFrames), dim3(256), 0, s, a);
  return hipGetLastError();
}

// The projection block of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1606-1669, pinhole
// case) for every (LastFrame point, camera) of a batch: Tcw * x3Dw is Sophus' unit-quaternion sandwich
// (Thirdparty/Sophus/sophus/so3.hpp:358-366: uv = q.vec x p; uv += uv; p + q.w uv + q.vec x uv) plus the translation, in Eigen's
// coefficient order without contraction; invzc = 1.0 / z in double, narrowed (:1626); Pinhole::project; the bounds skips; radius,
// the level window by the caller's forward / backward decision, ur.  Writes the 64-byte orbx_projected_point records the matcher
// kernels consume -- they stay in HBM.
__global__ __launch_bounds__(256) void k_project_last(LastProjArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
  if (i >= a.stride) return;
  const size_t e = (size_t)f * a.stride + i;
  const orbx_frame_pose_q& T = a.poses[f];
  const uint8_t fl = i < a.npts[f] ? a.flags[e] : 0;
  const uint4* d4 = reinterpret_cast<const uint4*>(a.desc + e * 32);
  const uint4 da = d4[0], db = d4[1];
  float u = 0.f, v = 0.f, ur = 0.f, radius = 0.f;
  int minL = 0, maxL = 0;
  bool ok = (fl & 1) != 0;
  if (ok) {
    const float px = a.pos[3 * e], py = a.pos[3 * e + 1], pz = a.pos[3 * e + 2];
    const float qx = T.q[0], qy = T.q[1], qz = T.q[2], qw = T.q[3];
    float ux = __fsub_rn(__fmul_rn(qy, pz), __fmul_rn(qz, py)), uy = __fsub_rn(__fmul_rn(qz, px), __fmul_rn(qx, pz)),
          uz = __fsub_rn(__fmul_rn(qx, py), __fmul_rn(qy, px));
    ux = __fadd_rn(ux, ux); uy = __fadd_rn(uy, uy); uz = __fadd_rn(uz, uz);
    const float cxv = __fsub_rn(__fmul_rn(qy, uz), __fmul_rn(qz, uy)), cyv = __fsub_rn(__fmul_rn(qz, ux), __fmul_rn(qx, uz)),
                czv = __fsub_rn(__fmul_rn(qx, uy), __fmul_rn(qy, ux));
    const float xc = __fadd_rn(__fadd_rn(__fadd_rn(px, __fmul_rn(qw, ux)), cxv), T.t[0]);
    const float yc = __fadd_rn(__fadd_rn(__fadd_rn(py, __fmul_rn(qw, uy)), cyv), T.t[1]);
    const float zc = __fadd_rn(__fadd_rn(__fadd_rn(pz, __fmul_rn(qw, uz)), czv), T.t[2]);
    const float invzc = (float)__ddiv_rn(1.0, (double)zc);
    ok = !(invzc < 0.f);
    u = __fadd_rn(__fdiv_rn(__fmul_rn(T.fx, xc), zc), T.cx);
    v = __fadd_rn(__fdiv_rn(__fmul_rn(T.fy, yc), zc), T.cy);
    ok = ok && u == u && v == v && !(u < a.minX || u > a.maxX) && !(v < a.minY || v > a.maxY);
    const int oct = min(max(a.octave[e], 0), a.nlevels - 1);
    radius = __fmul_rn(a.th, a.scale[oct]);
    if (T.direction == 1) { minL = oct; maxL = -1; }
    else if (T.direction == 2) { minL = 0; maxL = oct; }
    else { minL = oct - 1; maxL = oct + 1; }
    ur = __fsub_rn(u, __fmul_rn(T.bf, invzc));
  }
  uint32_t* o = reinterpret_cast<uint32_t*>(a.views + e);   // 64-byte records: 16 dwords
  const uint32_t z = 0u;
  o[0] = ok ? __builtin_bit_cast(uint32_t, u) : z; o[1] = ok ? __builtin_bit_cast(uint32_t, v) : z;
  o[2] = ok ? __builtin_bit_cast(uint32_t, ur) : z; o[3] = ok ? __builtin_bit_cast(uint32_t, radius) : z;
  o[4] = __builtin_bit_cast(uint32_t, i < a.npts[f] ? a.angle[e] : 0.f);
  o[5] = ok ? (uint32_t)minL : z; o[6] = ok ? (uint32_t)maxL : z;
  o[7] = (ok ? 1u : 0u) | ((uint32_t)((fl >> 1) & 1) << 8);
  o[8] = da.x; o[9] = da.y; o[10] = da.z; o[11] = da.w; o[12] = db.x; o[13] = db.y; o[14] = db.z; o[15] = db.w;
}
hipError_t launch_project_last(const LastProjArgs& a, int nFrames, hipStream_t s) {
  if (a.stride > 0 && nFrames > 0) hipLaunchKernelGGL(k_project_last, dim3((a.stride + 255) / 256, nFrames), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_proj_fill(const ProjArgs& a, hipStream_t s) {
  if (a.nmp > 0) hipLaunchKernelGGL(k_proj_cands<false>, dim3((a.nmp + 3) / 4), dim3(256), 0, s, ProjRef<false>{a}, 1);
  const size_t lds = (a.mode == 1 && a.checkOri) ? (size_t)(a.nmp + 4) * 4 : 16;  // one int per accepted match
  hipLaunchKernelGGL(k_proj_resolve<false>, dim3(1), dim3(64), lds, s, ProjRef<false>{a});
  return hipGetLastError();
}

}  // namespace orbx
