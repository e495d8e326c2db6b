// orbx_stereo.hip — association kernels of liborbx: Frame::ComputeStereoMatches (src/Frame.cc:921-1084), BFMatcher kNN-2
// (:1293-1302) and ComputeStereoFishEyeMatches with the KannalaBrandt8 triangulation (:1273-1331).
#include <mutex>

#include "orbx_device.h"

namespace orbx {

// ================================================================================================ stereo

// Both eyes' keypoints sorted by integer row (counting sort, one block per (pair, eye)): the analogue of the reference's
// vRowIndices table (src/Frame.cc:930-949).  Next to the CSR row table the block writes what the matcher reads, in row
// order: one 16-byte record per keypoint {x, y, octave | index << 8, minr | maxr << 16} -- [minr, maxr] = the +-2*scale row
// band of a RIGHT keypoint (:944-948); maxr = -1 marks the (0,0) points the reference skips (:943) -- and its descriptor.
// A band of left rows then finds its own keypoints and its candidates as two contiguous, coalesced ranges.
constexpr int kSortNT = 512, kSortItems = 4;  // register-resident path: up to 2048 keypoints per image
__global__ __launch_bounds__(kSortNT) void k_stereo_sort(Geom g, StereoArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int* hist = reinterpret_cast<int*>(smem);  // imgH + 1 counters, then running offsets
  __shared__ int wsum[kSortNT / 64];
  const int tid = threadIdx.x, pair = blockIdx.x >> 1, eye = blockIdx.x & 1;
#ifdef ST_PROF
  long long sq[12]; int nsq = 0;
#define SS_MK() do { if (nsq < 12) sq[nsq++] = wall_clock64(); } while (0)
#else
#define SS_MK() do {} while (0)
#endif
  SS_MK();
  const int img = (eye ? a.firstR : a.firstL) + pair;
  const int cap = eye ? a.capR : a.capL;
  const int n = (eye ? a.nR : a.nL)[img];
  const orbx_keypoint* kp = (eye ? a.kR : a.kL) + (long long)img * cap;
  const uint4* dsc = reinterpret_cast<const uint4*>((eye ? a.dR : a.dL) + (long long)img * cap * 32);
  int* rowStart = a.rowStart + (long long)(pair * 2 + eye) * (a.imgH + 1);
  uint4* rec = a.srec + (long long)(pair * 2 + eye) * a.cap;
  uint4* sd = a.sdesc + (long long)(pair * 2 + eye) * a.cap * 2;
  // everything the scatter needs is requested up front, beside the load of n (one memory round trip for the whole kernel
  // when n <= 2048)
  __shared__ float lvS[ORBX_MAX_LEVELS];
  if (tid < g.nlevels) lvS[tid] = g.lv[tid].scale;
  float kx[kSortItems], ky[kSortItems];
  int ko[kSortItems];
  uint4 kd0[kSortItems], kd1[kSortItems];
  {
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
      const int i = min(tid + j * kSortNT, cap - 1);  // (entries past n are inside the buffer and never used)
      kx[j] = kp[i].x;
      ky[j] = kp[i].y;
      ko[j] = kp[i].octave;
      kd0[j] = dsc[2 * i];
      kd1[j] = dsc[2 * i + 1];
    }
  }
  SS_MK();
  for (int r = tid; r <= a.imgH; r += kSortNT) hist[r] = 0;
  const bool inRegs = n <= kSortNT * kSortItems;
  __syncthreads();
  auto row_of = [&](float y) { return min(max((int)y, 0), a.imgH - 1); };
  if (inRegs) {
#pragma unroll
    for (int j = 0; j < kSortItems; j++)
      if (tid + j * kSortNT < n) atomicAdd(&hist[row_of(ky[j])], 1);
  } else {
    for (int i = tid; i < n; i += kSortNT) atomicAdd(&hist[row_of(kp[i].y)], 1);
  }
  SS_MK();
  __syncthreads();
  SS_MK();
  // exclusive prefix over the rows: a contiguous chunk per thread, wave scans of the chunk sums, wave totals through LDS
  const int per = (a.imgH + kSortNT) / kSortNT;
  const int b = min(tid * per, a.imgH + 1), e = min(b + per, a.imgH + 1);
  int sum = 0;
  for (int r = b; r < e; r++) sum += hist[r];
  const int incl = wave_scan_dpp(sum);
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < (tid >> 6); w++) run += wsum[w];
  for (int r = b; r < e; r++) {
    const int c = hist[r];
    hist[r] = run;
    rowStart[r] = run;
    run += c;
  }
  SS_MK();
  __syncthreads();
  SS_MK();
  // the permutation happens in LDS (when n <= 2048): 16-byte stores scattered over global memory, three per keypoint, were
  // half of the kernel's time; the sorted arrays then leave as linear, coalesced copies
  uint4* srecL = reinterpret_cast<uint4*>(smem + (((size_t)(a.imgH + 2) * 4 + 15) & ~(size_t)15));  // [kSortNT * kSortItems]
  uint4* sdL = srecL + kSortNT * kSortItems;                                                          // [.. * 2]
  auto emit = [&](int i, float x, float y, int oct, const uint4& d0, const uint4& d1, uint4* orec, uint4* odesc) {
    const int pos = atomicAdd(&hist[row_of(y)], 1);
    const float r = __fmul_rn(2.0f, lvS[oct]);
    int maxr = (int)ceilf(__fadd_rn(y, r)), minr = (int)floorf(__fsub_rn(y, r));
    if (y == 0.0f && x == 0.0f) maxr = -1;
    orec[pos] = make_uint4(__float_as_uint(x), __float_as_uint(y), (uint32_t)oct | ((uint32_t)i << 8),
                           ((uint32_t)minr & 0xFFFFu) | ((uint32_t)maxr << 16));
    odesc[2 * pos] = d0;
    odesc[2 * pos + 1] = d1;
  };
  if (inRegs) {
#pragma unroll
    for (int j = 0; j < kSortItems; j++)
      if (tid + j * kSortNT < n) emit(tid + j * kSortNT, kx[j], ky[j], ko[j], kd0[j], kd1[j], srecL, sdL);
    __syncthreads();
    for (int i = tid; i < n; i += kSortNT) rec[i] = srecL[i];
    for (int i = tid; i < 2 * n; i += kSortNT) sd[i] = sdL[i];
  } else {
    for (int i = tid; i < n; i += kSortNT) {
      const orbx_keypoint k = kp[i];
      emit(i, k.x, k.y, k.octave, dsc[2 * i], dsc[2 * i + 1], rec, sd);
    }
  }
  SS_MK();
#ifdef ST_PROF
  if (tid == 0 && blockIdx.x == 9) {
    printf("k_stereo_sort n %d:", n);
    for (int i = 1; i < nsq; i++) printf(" %d", (int)(sq[i] - sq[i - 1]));
    printf("  (x10 ns: issue loads | zero+hist | bar | scan | bar | scatter)\n");
  }
#endif
}

hipError_t launch_stereo_sort(const Geom& g, const StereoArgs& a, int npairs, hipStream_t s) {
  const size_t lds = (((size_t)(a.imgH + 2) * 4 + 15) & ~(size_t)15) + (size_t)kSortNT * kSortItems * 48;
  // > 64 KB of dynamic LDS needs the attribute, and the attribute is PER DEVICE (a process may hold handles on several GPUs):
  // one flag per device, set under a lock (two threads call the matcher concurrently)
  {
    static std::mutex mu;
    static bool prepared[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64 || !prepared[dev]) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_stereo_sort), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 64) prepared[dev] = true;
    }
  }
  hipLaunchKernelGGL(k_stereo_sort, dim3(2 * npairs), dim3(kSortNT), lds, s, g, a);
  return hipGetLastError();
}

// Median-of-SAD outlier cut (:1072-1083): median = element size/2 of the ascending (SAD, iL) list, i.e. the (size/2)-th
// smallest SAD; matches with SAD >= 1.5*1.4*median are dropped.  One 256-thread workgroup per pair, its own launch: run
// instead by the pair's last k_stereo_band workgroup to finish (arrival counter + __threadfence) it was bit-exact and 25x
// slower -- a device-scope release writes the XCD's L2 back, once per workgroup (k_stereo_band 23 -> 612 us).
// The order statistic is found exactly with a two-level LDS histogram (SAD <= 121*255 < 2^15: high 8 bits, low 7 bits); the
// bucket holding a rank is located with a block-wide prefix sum.
// first bucket k with hist[0] + .. + hist[k] > rank; returns (k, rank - (hist[0] + .. + hist[k-1])) through s_v[0], s_v[1]
__device__ __forceinline__ void bucket_of_rank(const int* hist, int rank, int* wsum, int* s_v, int tid) {
  const int h = hist[tid];
  const int incl = wave_scan_dpp(h);
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
  const int hi = base + incl, lo = hi - h;
  if (lo <= rank && rank < hi) {
    s_v[0] = tid;
    s_v[1] = rank - lo;
  }
  __syncthreads();
}
constexpr int kFiltItems = 8;  // register-resident path: up to 2048 left keypoints per pair
// hUr / hDepth (the single-frame entry): the filtered uRight / depth of the pair also go to the host block, from the thread that
// decides the cut -- its values are fetched beside the SADs, so the copy costs no second pass and no further memory round trip.
__device__ __forceinline__ void stereo_filter_pair(const StereoArgs& a, int pair, int tid, int* hist, int* wsum, int* s_v,
                                                   float* hUr = nullptr, float* hDepth = nullptr) {
  const int nL = a.nL[a.firstL + pair];
  const int* sad = a.sad + (long long)pair * a.capL;
  const bool inRegs = nL <= 256 * kFiltItems;
  int sv[kFiltItems];  // this thread's SADs (-1: no match / past the end): one memory round trip for all three passes
  float pu[kFiltItems], pd[kFiltItems];
#pragma unroll
  for (int j = 0; j < kFiltItems; j++) {
    const bool in = inRegs && tid + j * 256 < nL;
    sv[j] = in ? sad[tid + j * 256] : -1;
    pu[j] = (in && hUr) ? a.uRight[(long long)pair * a.capL + tid + j * 256] : -1.f;
    pd[j] = (in && hUr) ? a.depth[(long long)pair * a.capL + tid + j * 256] : -1.f;
  }
  hist[tid] = 0;
  __syncthreads();
  int cnt = 0;
  if (inRegs) {
#pragma unroll
    for (int j = 0; j < kFiltItems; j++)
      if (sv[j] >= 0) {
        atomicAdd(&hist[min(sv[j] >> 7, 255)], 1);
        cnt++;
      }
  } else {
    for (int i = tid; i < nL; i += 256) {
      const int s = sad[i];
      if (s >= 0) {
        atomicAdd(&hist[min(s >> 7, 255)], 1);
        cnt++;
      }
    }
  }
  cnt = wave_sum_dpp(cnt);
  if ((tid & 63) == 0) wsum[4 + (tid >> 6)] = cnt;
  __syncthreads();
  const int m = wsum[4] + wsum[5] + wsum[6] + wsum[7];
  if (m == 0) {  // the reference reads vDistIdx[0] of an empty vector here (UB) — guarded
    if (hUr) {
      for (int i = tid; i < nL; i += 256) {
        hUr[i] = a.uRight[(long long)pair * a.capL + i];
        hDepth[i] = a.depth[(long long)pair * a.capL + i];
      }
    }
    return;
  }
  bucket_of_rank(hist, m / 2, wsum, s_v, tid);  // 0-based rank of the median
  const int bucket = s_v[0], rankIn = s_v[1];
  __syncthreads();
  hist[tid] = 0;
  __syncthreads();
  if (inRegs) {
#pragma unroll
    for (int j = 0; j < kFiltItems; j++)
      if (sv[j] >= 0 && min(sv[j] >> 7, 255) == bucket) atomicAdd(&hist[sv[j] & 127], 1);
  } else {
    for (int i = tid; i < nL; i += 256) {
      const int s = sad[i];
      if (s >= 0 && min(s >> 7, 255) == bucket) atomicAdd(&hist[s & 127], 1);
    }
  }
  __syncthreads();
  bucket_of_rank(hist, rankIn, wsum, s_v, tid);
  const float median = (float)((bucket << 7) | s_v[0]);
  const float th = __fmul_rn(1.5f * 1.4f, median);
  auto cut = [&](int i, int s) {
    const bool c = s >= 0 && !((float)s < th);
    if (c) {
      a.uRight[(long long)pair * a.capL + i] = -1.f;
      a.depth[(long long)pair * a.capL + i] = -1.f;
    }
    return c;
  };
  if (inRegs) {
#pragma unroll
    for (int j = 0; j < kFiltItems; j++) {
      const bool c = cut(tid + j * 256, sv[j]);
      if (hUr && tid + j * 256 < nL) {
        hUr[tid + j * 256] = c ? -1.f : pu[j];
        hDepth[tid + j * 256] = c ? -1.f : pd[j];
      }
    }
  } else {
    for (int i = tid; i < nL; i += 256) {
      const bool c = cut(i, sad[i]);
      if (hUr) {
        hUr[i] = c ? -1.f : a.uRight[(long long)pair * a.capL + i];
        hDepth[i] = c ? -1.f : a.depth[(long long)pair * a.capL + i];
      }
    }
  }
}

// ComputeStereoMatches (src/Frame.cc:951-1068) for a band of kStereoBand left rows per workgroup.
//   1. the band's left keypoints (<= 64 per trip) and the right keypoints of rows [r0 - band, r0 + kStereoBand + band)
//      (<= kStereoRC per trip) come from the row-sorted arrays into LDS: two dependent memory round trips per workgroup
//      where the wave-per-keypoint kernel of rounds 1-2 made four per keypoint;
//   2. a wave per left keypoint, lanes over the staged candidates: the reference scans vRowIndices[vL] in ascending iR and
//      keeps the first strict minimum -- the minimum of (dist, iR) over the candidates passing the row-band / octave /
//      disparity tests, which is what the lanes compute (descriptor words are stored word-major: conflict-free);
//   3. the 11 x 11 SAD refinement (:1007-1062) with 16 lanes per matched keypoint, four keypoints per wave: a lane owns one
//      window row -- the left row's 11 bytes and the 21 right bytes all 11 shifts touch sit in registers (aligned dword
//      loads + v_alignbyte), three v_sad_u8 per shift -- and the 11 row sums meet in lane 15 by DPP row shifts.
//
// kDirect (round 5, the single-frame path): no k_stereo_sort in front.  Every workgroup scans the two UNSORTED keypoint arrays of
// its pair itself (x, y, octave of <= 2 x cap keypoints: L2-resident, one memory round trip beside the counts), keeps the indices of
// the left keypoints of its rows and of the right keypoints whose +-2*scale band touches those rows in two LDS lists, and stages
// records and descriptors through the lists.  Nothing below depends on the order of either list: the winner is the minimum of
// the unique keys (dist, iR), results are written by iL.  In a 0.2 ms frame the sort was 7.6 us of dependent launch; in a batch
// it is shared by the pair's 30 workgroups and stays.
constexpr int kStereoSelCap = 4096;  // per-image keypoint capacity the direct form's index lists hold
__device__ __forceinline__ uint32_t stereo_right_band(float x, float y, int oct, const float* lvS) {  // minr | maxr << 16 (:943-948)
  const float r = __fmul_rn(2.0f, lvS[oct]);
  int maxr = (int)ceilf(__fadd_rn(y, r));
  const int minr = (int)floorf(__fsub_rn(y, r));
  if (y == 0.0f && x == 0.0f) maxr = -1;
  return ((uint32_t)minr & 0xFFFFu) | ((uint32_t)maxr << 16);
}
// Workgroups behind the last band (direct form, rp.hKps != nullptr): the single-frame entry's gather of both eyes' keypoints and
// descriptors into the page-locked result block (k_result_pack's sections 0 - 3, orbx_kernels.hip) -- 180 KB of PCIe writes that do
// not depend on the association and so leave beside it instead of behind it, in the frame's last launch.
template <int kStereoBand, int NT, int kStereoRC, int kStereoLC, bool kDirect>
__global__ __launch_bounds__(NT) void k_stereo_band(Geom g, Pyr pl, Pyr pr, StereoArgs a, ResultPack rp) {
  if (kDirect) {
    const int nBands = (a.imgH + kStereoBand - 1) / kStereoBand;
    if ((int)blockIdx.x >= nBands) {
      const int nx = (int)gridDim.x - nBands, bx = (int)blockIdx.x - nBands;
      for (int sec = 0; sec < 4; sec++) {
        const int img = sec & 1;
        if (img >= rp.nimg) continue;
        const int n = min(rp.nOut[img], rp.cap);
        const uint32_t* src = sec < 2 ? rp.kps + (size_t)img * rp.cap * 7 : rp.desc + (size_t)img * rp.cap * 8;
        uint32_t* dst = sec < 2 ? rp.hKps + (size_t)img * rp.cap * 7 : rp.hDesc + (size_t)img * rp.cap * 8;
        const int len = n * (sec < 2 ? 7 : 8);
        for (int i = bx * NT + (int)threadIdx.x; i < len; i += nx * NT) dst[i] = src[i];
      }
      if (rp.hFlag) {  // the last gather workgroup to arrive publishes the counts and then the frame's sequence number
        __threadfence_system();   // this workgroup's writes to the host block are complete before its arrival counts
        __syncthreads();
        if (threadIdx.x == 0) {
          const int prev = __hip_atomic_fetch_add(rp.packCtr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
          if (prev == nx - 1) {
            __hip_atomic_store(rp.packCtr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int t = 0; t < 2; t++) {
              rp.hCnt[t] = t < rp.nimg ? (uint32_t)rp.nOut[t] : 0u;
              rp.hCnt[2 + t] = t < rp.nimg ? (uint32_t)rp.mono[t] : 0u;
            }
            __hip_atomic_store(rp.hFlag, rp.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
      return;
    }
  }
  __shared__ uint16_t selL[kDirect ? kStereoSelCap : 1], selR[kDirect ? kStereoSelCap : 1];
  __shared__ int cntSel[2];
  __shared__ uint4 Lrec[kStereoLC];
  __shared__ uint32_t Ldesc[kStereoLC][8];
  __shared__ uint32_t Lbest[kStereoLC];
  __shared__ float Lbx[kStereoLC];
  __shared__ uint8_t Lq[kStereoLC];  // entries with an ORB match, in entry order
  __shared__ int nQ, nQw[2];
  __shared__ uint4 Rrec[kStereoRC];
  __shared__ uint32_t Rdesc[8][kStereoRC];
  __shared__ int lvW[ORBX_MAX_LEVELS], lvP[ORBX_MAX_LEVELS];
  __shared__ long long lvO[ORBX_MAX_LEVELS];
  __shared__ float lvS[ORBX_MAX_LEVELS];
  const int tid = threadIdx.x;
  const int pair = blockIdx.y, r0 = blockIdx.x * kStereoBand;
#ifdef ST_PROF
  long long tq[16]; int nq_ = 0;
#define ST_MK() do { if (nq_ < 16) tq[nq_++] = wall_clock64(); } while (0)
#else
#define ST_MK() do {} while (0)
#endif
  ST_MK();
  const int imgL = a.firstL + pair, imgR = a.firstR + pair;
  int jLb, jLe, jRb, jRe;
  const uint4 *recL = nullptr, *recR = nullptr, *sdL, *sdR;
  const uint32_t *kwL = nullptr, *kwR = nullptr;  // (direct) the keypoint records as 7 dwords: x, y, .., octave at 5
  if (!kDirect) {
    const int* rsL = a.rowStart + (long long)(pair * 2) * (a.imgH + 1);
    const int* rsR = rsL + (a.imgH + 1);
    jLb = rsL[r0];
    jLe = rsL[min(r0 + kStereoBand, a.imgH)];
    if (jLb == jLe) return;  // no left keypoint in these rows
    jRb = rsR[max(r0 - a.band, 0)];
    jRe = rsR[min(r0 + kStereoBand + a.band, a.imgH)];
    recL = a.srec + (long long)(pair * 2) * a.cap;
    recR = recL + a.cap;
    sdL = a.sdesc + (long long)(pair * 2) * a.cap * 2;
    sdR = sdL + (long long)a.cap * 2;
  } else {
    kwL = reinterpret_cast<const uint32_t*>(a.kL + (long long)imgL * a.capL);
    kwR = reinterpret_cast<const uint32_t*>(a.kR + (long long)imgR * a.capR);
    sdL = reinterpret_cast<const uint4*>(a.dL + (long long)imgL * a.capL * 32);
    sdR = reinterpret_cast<const uint4*>(a.dR + (long long)imgR * a.capR * 32);
    if (tid < 2) cntSel[tid] = 0;
  }
  if (tid < g.nlevels) {
    lvW[tid] = g.lv[tid].w;
    lvP[tid] = tid ? g.lv[tid].pitch : (int)pl.l0Row;
    lvO[tid] = g.lv[tid].off;
    lvS[tid] = g.lv[tid].scale;
  }
  const float maxD = __fdiv_rn(a.bf, a.b);
  if (kDirect) {
    const int nL = a.nL[imgL], nR = a.nR[imgR];
    constexpr int kIt = 4;  // keypoints per thread and trip: one trip up to NT * 4 keypoints per image
    for (int base = 0; base == 0 || base < max(nL, nR); base += NT * kIt) {
      uint32_t yl[kIt], xr[kIt], yr[kIt], oc[kIt];
#pragma unroll
      for (int j = 0; j < kIt; j++) {  // (entries past n are inside the buffers and never used)
        const int i = base + tid + j * NT;
        yl[j] = kwL[min(i, a.capL - 1) * 7 + 1];
        const uint32_t* q = kwR + min(i, a.capR - 1) * 7;
        xr[j] = q[0];
        yr[j] = q[1];
        oc[j] = q[5];
      }
      if (base == 0) __syncthreads();  // the level table and the two counters
#pragma unroll
      for (int j = 0; j < kIt; j++) {
        const int i = base + tid + j * NT;
        if (i < nL) {
          const int row = min(max((int)__uint_as_float(yl[j]), 0), a.imgH - 1);  // k_stereo_sort's row_of
          if (row >= r0 && row < r0 + kStereoBand) selL[atomicAdd(&cntSel[0], 1)] = (uint16_t)i;
        }
        if (i < nR) {
          const uint32_t mm = stereo_right_band(__uint_as_float(xr[j]), __uint_as_float(yr[j]), (int)oc[j], lvS);
          const int minr = (int)(int16_t)(mm & 0xFFFF), maxr = (int)mm >> 16;
          if (maxr >= r0 && minr < r0 + kStereoBand) selR[atomicAdd(&cntSel[1], 1)] = (uint16_t)i;
        }
      }
    }
    __syncthreads();
    jLb = 0;
    jLe = cntSel[0];
    jRb = 0;
    jRe = cntSel[1];
    if (jLe == 0) return;  // no left keypoint in these rows
  }
  ST_MK();
  for (int lc = jLb; lc < jLe; lc += kStereoLC) {
    const int nl = min(kStereoLC, jLe - lc);
    __syncthreads();  // (the previous trip's SAD stage has read Lrec / Lbest)
    if (tid < nl) {
      if (kDirect) {
        const uint32_t k = selL[lc + tid];
        const uint32_t* q = kwL + k * 7;
        Lrec[tid] = make_uint4(q[0], q[1], q[5] | (k << 8), 0u);
      } else {
        Lrec[tid] = recL[lc + tid];
      }
      Lbest[tid] = 100u << 16;  // TH_HIGH, strict '<'
      Lbx[tid] = 0.f;
    }
    for (int i = tid; i < 2 * nl; i += NT) {
      const uint4 d = kDirect ? sdL[2 * (int)selL[lc + (i >> 1)] + (i & 1)] : sdL[2 * lc + i];
      uint32_t* q = &Ldesc[i >> 1][(i & 1) * 4];
      q[0] = d.x; q[1] = d.y; q[2] = d.z; q[3] = d.w;
    }
    for (int rc = jRb; rc < jRe; rc += kStereoRC) {
      const int nr = min(kStereoRC, jRe - rc);
      __syncthreads();  // (the previous trip's match stage has read Rrec / Rdesc)
      for (int i = tid; i < nr; i += NT) {
        if (kDirect) {
          const uint32_t k = selR[rc + i];
          const uint32_t* q = kwR + k * 7;
          const uint32_t x = q[0], y = q[1], oct = q[5];
          Rrec[i] = make_uint4(x, y, oct | (k << 8), stereo_right_band(__uint_as_float(x), __uint_as_float(y), (int)oct, lvS));
        } else {
          Rrec[i] = recR[rc + i];
        }
      }
      for (int i = tid; i < 2 * nr; i += NT) {
        const uint4 d = kDirect ? sdR[2 * (int)selR[rc + (i >> 1)] + (i & 1)] : sdR[2 * rc + i];
        const int c = i >> 1, k0 = (i & 1) * 4;
        Rdesc[k0][c] = d.x; Rdesc[k0 + 1][c] = d.y; Rdesc[k0 + 2][c] = d.z; Rdesc[k0 + 3][c] = d.w;
      }
      ST_MK();
      __syncthreads();
      ST_MK();
      // 16 lanes per left keypoint (four keypoints per wave side by side), lane s takes candidates s, s + 16, ...
      for (int li = tid >> 4; li < nl; li += NT / 16) {
        const uint4 lr = Lrec[li];
        const float uL = __uint_as_float(lr.x), vL = __uint_as_float(lr.y);
        const int levelL = (int)(lr.z & 0xFF);
        const float minU = __fsub_rn(uL, maxD), maxU = uL;
        const int row = (int)vL;
        uint32_t best = 100u << 16;
        float bestX = 0.f;
        if (!(maxU < 0)) {
          uint32_t dl[8];
#pragma unroll
          for (int k = 0; k < 8; k++) dl[k] = Ldesc[li][k];
          for (int c = tid & 15; c < nr; c += 16) {
            const uint4 rr = Rrec[c];
            const float x = __uint_as_float(rr.x);
            const int oct = (int)(rr.z & 0xFF);
            const int minr = (int)(int16_t)(rr.w & 0xFFFF), maxr = (int)rr.w >> 16;
            if (row >= minr && row <= maxr && oct >= levelL - 1 && oct <= levelL + 1 && x >= minU && x <= maxU) {
              int d = 0;
#pragma unroll
              for (int k = 0; k < 8; k++) d += __popc(dl[k] ^ Rdesc[k][c]);
              const uint32_t cand = ((uint32_t)d << 16) | (rr.z >> 8);
              if (cand < best) {
                best = cand;
                bestX = x;
              }
            }
          }
        }
        // minimum over the 16 lanes of the row (lanes without a source keep their own value), then lane 15's total back
        // to every lane of the row; keys are unique (iR), so exactly one lane owns the minimum
        uint32_t m = best;
        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x111, 0xf, 0xf, false));
        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x112, 0xf, 0xf, false));
        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x114, 0xf, 0xf, false));
        m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x118, 0xf, 0xf, false));
        m = (uint32_t)__shfl((int)m, 15, 16);
        if (best == m && m < Lbest[li]) {  // (this row of lanes is the only writer of entry li)
          Lbest[li] = m;
          Lbx[li] = bestX;
        }
      }
    }
    ST_MK();
    __syncthreads();
    ST_MK();
    // ---- left keypoints without an ORB match are finished; the others queue for the SAD refinement
    {
      const bool matched = tid < nl && (int)(Lbest[min(tid, kStereoLC - 1)] >> 16) < 75;  // thOrbDist = (TH_HIGH + TH_LOW) / 2
      const uint64_t mm = __ballot(matched);  // (entries live in the first kStereoLC / 64 waves)
      if (kStereoLC > 64) {
        if (tid < kStereoLC && (tid & 63) == 0) nQw[tid >> 6] = (int)__popcll(mm);
        __syncthreads();
      }
      const int base = (kStereoLC > 64 && tid >= 64) ? nQw[0] : 0;
      if (matched) Lq[base + prefix_count(mm)] = (uint8_t)tid;
      if (tid == 0) nQ = kStereoLC > 64 ? nQw[0] + nQw[1] : (int)__popcll(mm);
      if (tid < nl && !matched) {
        const long long o = (long long)pair * a.capL + (int)(Lrec[tid].z >> 8);
        a.uRight[o] = -1.f;
        a.depth[o] = -1.f;
        a.sad[o] = -1;
      }
    }
    __syncthreads();
    // ---- SAD refinement: group = 16 lanes, lane r < 11 = window row r; groups take queue entries grp, grp + NT/16, ...
    const int grp = tid >> 4, r = tid & 15;
    const int nq = nQ;
    for (int qi = grp; qi < nq; qi += NT / 16) {
      const int li = Lq[qi];
      const uint4 lr = Lrec[li];
      const uint32_t bestKey = Lbest[li];
      const float uL = __uint_as_float(lr.x), vL = __uint_as_float(lr.y), uR0 = Lbx[li];
      const int levelL = (int)(lr.z & 0xFF), iL = (int)(lr.z >> 8);
      float uR_out = -1.f, depth_out = -1.f;
      int sad_out = -1;
      (void)bestKey;
      {
        const float scl = lvS[levelL];
        const float sf = 1.0f / scl;  // mvInvScaleFactors
        const float su = roundf(__fmul_rn(uL, sf)), sv = roundf(__fmul_rn(vL, sf)), sr = roundf(__fmul_rn(uR0, sf));
        const float endu = sr + 11.0f;
        if (!(sr < 0 || endu >= (float)lvW[levelL])) {
          const int pitchL = lvP[levelL], pitchR = levelL ? pitchL : (int)pr.l0Row;
          const uint8_t* imL = levelL ? pl.pyr + (long long)imgL * g.pyrImg + lvO[levelL] : pl.l0 + (long long)imgL * pl.l0Img;
          const uint8_t* imR = levelL ? pr.pyr + (long long)imgR * g.pyrImg + lvO[levelL] : pr.l0 + (long long)imgR * pr.l0Img;
          const int yl = (int)sv - 5, xl = (int)su - 5, xr = (int)sr - 10;  // right window of shift inc starts at xr + inc
          int sadv[11];
          {
            uint32_t L0 = 0, L1 = 0, L2 = 0, R[6] = {0, 0, 0, 0, 0, 0};
            if (r < 11) {
              const int oL = (yl + r) * pitchL + xl, oR = (yl + r) * pitchR + xr;
              const uint32_t* qL = reinterpret_cast<const uint32_t*>(imL + (oL & ~3));
              const uint32_t* qR = reinterpret_cast<const uint32_t*>(imR + (oR & ~3));
              const uint32_t l0 = qL[0], l1 = qL[1], l2 = qL[2], l3 = qL[3];
              const uint32_t e0 = qR[0], e1 = qR[1], e2 = qR[2], e3 = qR[3], e4 = qR[4], e5 = qR[5];
              const uint32_t shL = (uint32_t)(oL & 3), shR = (uint32_t)(oR & 3);
              L0 = __builtin_amdgcn_alignbyte(l1, l0, shL);
              L1 = __builtin_amdgcn_alignbyte(l2, l1, shL);
              L2 = __builtin_amdgcn_alignbyte(l3, l2, shL) & 0x00FFFFFFu;  // bytes 8..10
              R[0] = __builtin_amdgcn_alignbyte(e1, e0, shR);
              R[1] = __builtin_amdgcn_alignbyte(e2, e1, shR);
              R[2] = __builtin_amdgcn_alignbyte(e3, e2, shR);
              R[3] = __builtin_amdgcn_alignbyte(e4, e3, shR);
              R[4] = __builtin_amdgcn_alignbyte(e5, e4, shR);
              R[5] = __builtin_amdgcn_alignbyte(0u, e5, shR);
            }
#pragma unroll
            for (int inc = 0; inc < 11; inc++) {
              const int k = inc >> 2, sh = inc & 3;
              const uint32_t W0 = sh ? __builtin_amdgcn_alignbyte(R[k + 1], R[k], (uint32_t)sh) : R[k];
              const uint32_t W1 = sh ? __builtin_amdgcn_alignbyte(R[k + 2], R[k + 1], (uint32_t)sh) : R[k + 1];
              const uint32_t W2 = (sh ? __builtin_amdgcn_alignbyte(R[k + 3], R[k + 2], (uint32_t)sh) : R[k + 2]) & 0x00FFFFFFu;
              uint32_t sum = __builtin_amdgcn_sad_u8(L0, W0, 0u);
              sum = __builtin_amdgcn_sad_u8(L1, W1, sum);
              sum = __builtin_amdgcn_sad_u8(L2, W2, sum);
              int v = (int)sum;  // rows 11..15 hold zeros: L* = R* = 0
              v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1 .. 8: lane 15 = the row's total
              v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
              v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
              v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
              sadv[inc] = v;
            }
          }
          if (r == 15) {
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
                float bestuR = __fmul_rn(scl, __fadd_rn(__fadd_rn(sr, (float)bestinc), deltaR));
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
      }
      if (r == 15) {
        const long long o = (long long)pair * a.capL + iL;
        a.uRight[o] = uR_out;
        a.depth[o] = depth_out;
        a.sad[o] = sad_out;
      }
    }
    ST_MK();
  }
#ifdef ST_PROF
  if (tid == 0 && ((pair % 8) == 5 || gridDim.y == 1) && (blockIdx.x % 9) == 4) {
    char buf[200]; (void)buf;
    printf("band %2d pair %2d nL %2d nR %2d start %5lld: tab %d | stage %d bar %d match %d bar %d queue+SAD %d  (x10 ns; total %d)\n", (int)blockIdx.x, pair,
           jLe - jLb, jRe - jRb, tq[0] % 100000, (int)(tq[1] - tq[0]), (int)(tq[2] - tq[1]), (int)(tq[3] - tq[2]), (int)(tq[4] - tq[3]),
           (int)(tq[5] - tq[4]), (int)(tq[6] - tq[5]), (int)(tq[nq_ - 1] - tq[0]));
  }
#endif
}

__global__ __launch_bounds__(256) void k_stereo_filter(StereoArgs a) {
  __shared__ int fhist[256], fw[8], fv[2];
  stereo_filter_pair(a, blockIdx.x, threadIdx.x, fhist, fw, fv);
}
hipError_t launch_stereo_filter(const StereoArgs& a, int npairs, hipStream_t s) {
  hipLaunchKernelGGL(k_stereo_filter, dim3(npairs), dim3(256), 0, s, a);
  return hipGetLastError();
}

// Single stereo frame through the host API (orbx_extract_stereo): the median cut of pair 0 and the gather of every result into
// the handle's page-locked host block (k_result_pack, orbx_kernels.hip) in ONE launch -- the two were 4.6 + 5.9 us of dependent
// launches at the end of a 0.2 ms frame.  Sections as in k_result_pack (blockIdx.y: 0 / 1 keypoints, 2 / 3 descriptors of the
// two eyes, 6 counts) run beside block (0, 4), which filters and then copies uRight and depth itself.
__global__ __launch_bounds__(256) void k_stereo_filter_pack(StereoArgs sa, ResultPack a) {
  __shared__ int fhist[256], fw[8], fv[2];
  // (grid (1, 1): keypoints and descriptors have left with k_stereo_band's extra workgroups; this one publishes the counts too)
  const int sec = gridDim.y == 1 ? 4 : blockIdx.y;
  if (gridDim.y == 1 && threadIdx.x < 2) {
    const int t = threadIdx.x;
    a.hCnt[t] = t < a.nimg ? (uint32_t)a.nOut[t] : 0u;
    a.hCnt[2 + t] = t < a.nimg ? (uint32_t)a.mono[t] : 0u;
  }
  if (sec == 6) {
    if (blockIdx.x == 0 && threadIdx.x < 2) {
      const int t = threadIdx.x;
      a.hCnt[t] = t < a.nimg ? (uint32_t)a.nOut[t] : 0u;
      a.hCnt[2 + t] = t < a.nimg ? (uint32_t)a.mono[t] : 0u;
    }
    return;
  }
  if (sec == 5) return;
  if (sec == 4) {
    if (blockIdx.x != 0) return;
    // (nL of pair 0 == nOut[0] <= cap: the filter's own threads write the host copies of uRight / depth)
    stereo_filter_pair(sa, 0, threadIdx.x, fhist, fw, fv, reinterpret_cast<float*>(a.hUr), reinterpret_cast<float*>(a.hDepth));
    return;
  }
  const int img = sec & 1;
  if (img >= a.nimg) return;
  const int n = min(a.nOut[img], a.cap);
  const uint32_t* src;
  uint32_t* dst;
  int len;
  if (sec < 2) {
    src = a.kps + (size_t)img * a.cap * 7; dst = a.hKps + (size_t)img * a.cap * 7; len = n * 7;
  } else {
    src = a.desc + (size_t)img * a.cap * 8; dst = a.hDesc + (size_t)img * a.cap * 8; len = n * 8;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < len; i += gridDim.x * 256) dst[i] = src[i];
}
hipError_t launch_stereo_filter_pack(const StereoArgs& sa, const ResultPack& a, hipStream_t s, bool stereoOnly) {
  hipLaunchKernelGGL(k_stereo_filter_pack, stereoOnly ? dim3(1, 1) : dim3(12, 7), dim3(256), 0, s, sa, a);
  return hipGetLastError();
}

// pairs per call up to which the direct form runs (default 1: the single-frame path; ORBX_STEREO_DIRECT_PAIRS / test hook)
static int g_stereo_direct_pairs = getenv("ORBX_STEREO_DIRECT_PAIRS") ? atoi(getenv("ORBX_STEREO_DIRECT_PAIRS")) : 1;
void debug_set_stereo_direct(int max_pairs) { g_stereo_direct_pairs = max_pairs < 0 ? 1 : max_pairs; }
bool stereo_direct_ok(const StereoArgs& a, int npairs) {
  return npairs <= g_stereo_direct_pairs && a.capL <= kStereoSelCap && a.capR <= kStereoSelCap;
}
hipError_t launch_stereo_match(const Geom& g, const Pyr& pl, const Pyr& pr, const StereoArgs& a, int npairs,
                               hipStream_t s, bool direct, const ResultPack* pack) {
  const int packBlocks = (direct && pack && npairs == 1) ? 8 : 0;   // (workgroups behind the bands: the result gather)
  ResultPack rp{};
  if (packBlocks) rp = *pack;
  // band rows / threads / right-trip / left-trip sizes measured at 1280x720, 32 pairs (kernel alone): 8/256/256/64 22.9 us,
  // 16/256/256/64 26.5, 16/512/256/64 24.7, 24/512/256/128 20.7, 32/512/256/128 20.3, 32/1024/512/128 23.6, 8/128/256/64 28.8
#define ORBX_SB(BR, NT, RC, LC) do { if (direct) hipLaunchKernelGGL((k_stereo_band<BR, NT, RC, LC, true>), dim3((a.imgH + BR - 1) / BR + packBlocks, npairs), dim3(NT), 0, s, g, pl, pr, a, rp); \
    else hipLaunchKernelGGL((k_stereo_band<BR, NT, RC, LC, false>), dim3((a.imgH + BR - 1) / BR, npairs), dim3(NT), 0, s, g, pl, pr, a, rp); } while (0)
  // (64-thread workgroups -- 4 rows / 64 candidates per trip -- run 41 us alone and the 3-handle step of bench.py is 0.8 %
  // SHORTER with them, 0.4853 vs 0.4895 ms: small workgroups get scheduled between k_detect's one-wave cells, large ones
  // wait for it to drain; not adopted, the single-frame latency matters more than 0.8 %)
#ifdef ORBX_BAND_SWEEP   // measurement aid (make prof PROF_FLAGS=-DORBX_BAND_SWEEP): band shapes of the direct form, ORBX_BAND_SHAPE=0..5
  static const int shape = getenv("ORBX_BAND_SHAPE") ? atoi(getenv("ORBX_BAND_SHAPE")) : 0;
  if (direct && shape == 1) ORBX_SB(12, 256, 256, 64);
  else if (direct && shape == 2) ORBX_SB(24, 512, 256, 128);
  else if (direct && shape == 3) ORBX_SB(12, 512, 256, 64);
  else if (direct && shape == 4) ORBX_SB(8, 256, 128, 64);
  else if (direct && shape == 5) ORBX_SB(32, 1024, 256, 128);
  else
#endif
  // the direct form (one pair): 16-row bands -- the C call of a 1280x720 frame 0.1923 (24 rows) / 0.1900 (16) / 0.1902 (12 rows, 64 left
  // entries) / 0.1905 ms (8 rows, 256 threads), tools/lat_c.py on one box, three runs each within 0.5 us
  if (direct) ORBX_SB(16, 512, 256, 128);
  else ORBX_SB(24, 512, 256, 128);
#undef ORBX_SB
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

// The same 2-NN with 16 lanes per query (train rows t = lane, lane + 16, ...): a one-shot call has ~1500 queries, and a
// thread per query left all but six CUs idle behind a 1500-step serial loop.  (distance << 16 | train index) keys are all
// different, so "best = first minimum, second = the next smallest, an equal later distance included" is simply the two
// smallest keys, and partial results merge with min / max.  nT < 65536.
__global__ __launch_bounds__(256) void k_bf_knn2_split(const uint8_t* __restrict__ dQ, int nQ, const uint8_t* __restrict__ dT,
                                                       int nT, int* __restrict__ idx2, int* __restrict__ dist2,
                                                       uint8_t* __restrict__ ok) {
  const int q = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
  if (q >= nQ) return;  // whole 16-lane groups leave together
  uint32_t dq[8];
#pragma unroll
  for (int i = 0; i < 8; i++) dq[i] = reinterpret_cast<const uint32_t*>(dQ)[(long long)q * 8 + i];
  uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
  for (int t = sub; t < nT; t += 16) {
    const uint32_t k = ((uint32_t)hamming256(dq, reinterpret_cast<const uint32_t*>(dT) + (long long)t * 8) << 16) | (uint32_t)t;
    k1 = min(k1, max(k0, k));
    k0 = min(k0, k);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const uint32_t o0 = (uint32_t)__shfl_xor((int)k0, o, 16), o1 = (uint32_t)__shfl_xor((int)k1, o, 16);
    k1 = min(min(k1, o1), max(k0, o0));
    k0 = min(k0, o0);
  }
  if (sub == 0) {
    const bool h0 = k0 != 0xFFFFFFFFu, h1 = k1 != 0xFFFFFFFFu;
    const int b0 = (int)(k0 >> 16), b1 = (int)(k1 >> 16);
    idx2[2 * q] = h0 ? (int)(k0 & 0xFFFFu) : -1;
    idx2[2 * q + 1] = h1 ? (int)(k1 & 0xFFFFu) : -1;
    dist2[2 * q] = h0 ? b0 : -1;
    dist2[2 * q + 1] = h1 ? b1 : -1;
    ok[q] = (h0 && h1 && (double)(float)b0 < __dmul_rn((double)(float)b1, 0.7)) ? 1 : 0;
  }
}

hipError_t launch_bf_knn2(const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, int* idx2, int* dist2,
                          uint8_t* ok, hipStream_t s) {
  if (nQ <= 0) return hipSuccess;
  if (nT < 65536)
    hipLaunchKernelGGL(k_bf_knn2_split, dim3((nQ + 15) / 16), dim3(256), 0, s, dQ, nQ, dT, nT, idx2, dist2, ok);
  else
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

// 1 / x and 1 / sqrt(x) in double from the hardware seeds (v_rcp_f64 / v_rsq_f64) and two Newton steps each: relative error
// ~1e-16 instead of correctly rounded, at a fifth of the IEEE division / square-root expansions (the Jacobi sweeps below are a
// serial chain of them per triangulated match: round 4 measured 35 us of k_fisheye_batch's 60 us there).  The rotation angles of
// a one-sided Jacobi may be off by an ulp without changing what it converges to.
__device__ __forceinline__ double rcp64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double rsqrt64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y * __builtin_fma(-0.5 * x * y, y, 1.5);
}

// Right singular vector of the smallest singular value (= JacobiSVD::matrixV().col(3), :429-431) of a row-major 4x4.
// Round 6: shifted inverse iteration on B = A^T A in double instead of the one-sided Jacobi sweeps of rounds 1 - 5 (a serial chain
// of ~4000 double operations per triangulating lane: half of k_fisheye_batch).  The vector is the eigenvector of B's smallest
// eigenvalue; B + mu I (mu = 1e-14 trace: keeps the LDL^T pivots positive) is factored once, each solve multiplies the wanted
// component by (s3^2 + mu) / (s4^2 + mu) >= 10^2 .. 10^4 for a pair that passes the parallax gate (:356), and the iteration stops
// when the normalised vector has settled to 1e-13.  Against the Jacobi vector: the eigenvector of A^T A carries
// eps (s1 / s3)^2 ~ 1e-11 of relative error, far inside the 2e-4 of the float-tail parity (tests/test_fisheye.py).
__device__ void null_vector4(const float A[16], float v[4]) {
  double B[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = i; j < 4; j++) {
      double sum = 0.0;
#pragma unroll
      for (int k = 0; k < 4; k++) sum = __builtin_fma((double)A[4 * k + i], (double)A[4 * k + j], sum);
      B[i][j] = sum;
    }
  const double mu = 1e-14 * (B[0][0] + B[1][1] + B[2][2] + B[3][3]) + 1e-300;
  // LDL^T of the symmetric positive definite B + mu I (upper triangle in, unit lower L and 1 / d out)
  double L10, L20, L30, L21, L31, L32, id0, id1, id2, id3;
  {
    const double d0 = B[0][0] + mu;
    id0 = rcp64(d0);
    L10 = B[0][1] * id0; L20 = B[0][2] * id0; L30 = B[0][3] * id0;
    const double d1 = B[1][1] + mu - L10 * B[0][1];
    id1 = rcp64(d1);
    const double t21 = B[1][2] - L10 * B[0][2], t31 = B[1][3] - L10 * B[0][3];
    L21 = t21 * id1; L31 = t31 * id1;
    const double d2 = B[2][2] + mu - L20 * B[0][2] - L21 * t21;
    id2 = rcp64(d2);
    const double t32 = B[2][3] - L20 * B[0][3] - L21 * t31;
    L32 = t32 * id2;
    const double d3 = B[3][3] + mu - L30 * B[0][3] - L31 * t31 - L32 * t32;
    id3 = rcp64(d3);
  }
  double x0 = 0.5, x1 = 0.5, x2 = 0.5, x3 = 0.5;
  for (int it = 0; it < 30; it++) {   // (3 - 4 solves; a start vector that happens to be orthogonal to the answer needs ~10)
    // L y = x, z = y / d, L^T w = z
    const double y0 = x0, y1 = x1 - L10 * y0, y2 = x2 - L20 * y0 - L21 * y1, y3 = x3 - L30 * y0 - L31 * y1 - L32 * y2;
    const double w3 = y3 * id3, w2 = y2 * id2 - L32 * w3, w1 = y1 * id1 - L21 * w2 - L31 * w3, w0 = y0 * id0 - L10 * w1 - L20 * w2 - L30 * w3;
    const double rn = rsqrt64(w0 * w0 + w1 * w1 + w2 * w2 + w3 * w3);
    const double n0 = w0 * rn, n1 = w1 * rn, n2 = w2 * rn, n3 = w3 * rn;
    // (the sign is fixed by the solve itself: (B + mu I)^-1 is positive definite, consecutive iterates never flip)
    const double dx = fabs(n0 - x0) + fabs(n1 - x1) + fabs(n2 - x2) + fabs(n3 - x3);
    x0 = n0; x1 = n1; x2 = n2; x3 = n3;
    if (it > 0 && dx < 1e-13) break;
  }
  v[0] = (float)x0;
  v[1] = (float)x1;
  v[2] = (float)x2;
  v[3] = (float)x3;
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

// ORBmatcher::SearchForTriangulation, two-camera-rig branch (src/ORBmatcher.cc:906-923, 1007-1064): k_tri_match's walk (16 lanes per
// feature of pKF1, minimum of (distance, -position) over the gate-passers) with KannalaBrandt8::epipolarConstrain as the gate --
// TriangulateMatches(...) > 0.0001f on the (R12, t12, cameras) of the eyes the two features sit in.
__global__ __launch_bounds__(256) void k_tri_match_rig(TriArgs a) {
  const int sub = threadIdx.x & 15;
  const int g = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (g >= a.nList1) return;
  int lo = 0, hi = a.nNodes1 - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.start1[mid] <= g) lo = mid; else hi = mid - 1;
  }
  const uint32_t node = a.nodes1[lo];
  int l2 = 0, h2 = a.nNodes2;
  while (l2 < h2) {
    const int mid = (l2 + h2) >> 1;
    if (a.nodes2[mid] < node) l2 = mid + 1; else h2 = mid;
  }
  if (l2 >= a.nNodes2 || a.nodes2[l2] != node) return;
  const int b = a.start2[l2], e = a.start2[l2 + 1];
  const int idx1 = (int)a.feat1[g];
  if (a.mp1[idx1] || a.onlyStereo) return;  // bStereo1 is false for a rig: bOnlyStereo skips every feature (:957-959)
  const orbx_keypoint kp1 = a.k1[idx1];
  const bool right1 = idx1 >= a.nLeft1;
  KB8Cam c1;
#pragma unroll
  for (int i = 0; i < 8; i++) c1.p[i] = a.rig->cam[right1 ? 1 : 0][i];
  c1.precision = a.rig->precision;
  const float sigma1 = a.sigma1[kp1.octave];
  uint32_t d1[8];
#pragma unroll
  for (int i = 0; i < 8; i++) d1[i] = a.d1[(long long)idx1 * 8 + i];
  uint32_t best = 0xFFFFFFFFu;  // (dist << 24) | (0xFFFFFF - position)
  for (int pos = b + sub; pos < e; pos += 16) {
    const int idx2 = (int)a.feat2[pos];
    if (a.mp2[idx2]) continue;
    const int dist = hamming256(d1, a.d2 + (long long)idx2 * 8);
    if (dist > 50) continue;  // TH_LOW
    if (!a.coarse) {
      const orbx_keypoint kp2 = a.k2[idx2];
      const bool right2 = idx2 >= a.nLeft2;
      const int sel = (right1 ? 2 : 0) + (right2 ? 1 : 0);  // ll, lr, rl, rr (:1008-1041)
      KB8Cam c2;
#pragma unroll
      for (int i = 0; i < 8; i++) c2.p[i] = a.rig->cam[right2 ? 3 : 2][i];
      c2.precision = a.rig->precision;
      float P[3];
      const float z = kb8_triangulate_matches(c1, c2, kp1.x, kp1.y, kp2.x, kp2.y, a.rig->R[sel], a.rig->t[sel], sigma1,
                                              a.sigma2[kp2.octave], P);
      if (!(z > 0.0001f)) continue;  // KannalaBrandt8.cpp:248-249
    }
    best = min(best, ((uint32_t)dist << 24) | (0xFFFFFFu - (uint32_t)(pos - b)));
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 16));
  if (sub == 0 && best != 0xFFFFFFFFu) {
    const int idx2 = (int)a.feat2[b + (int)(0xFFFFFFu - (best & 0xFFFFFFu))];
    a.match[idx1] = idx2;
    atomicAdd(&a.flags[0], 1);
    if (a.checkOri) {
      float rot = kp1.angle - a.k2[idx2].angle;
      if (rot < 0.0f) rot = rot + 360.0f;
      int bin = (int)roundf(rot * (1.0f / 30));
      if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
      atomicAdd(&a.flags[2 + bin], 1);
    }
  }
}

hipError_t launch_tri_match_rig(const TriArgs& a, hipStream_t s) {
  if (a.nList1 > 0 && a.nNodes2 > 0) hipLaunchKernelGGL(k_tri_match_rig, dim3((a.nList1 + 15) / 16), dim3(256), 0, s, a);
  return hipGetLastError();
}

// Batched variant on the extractors' device-resident results: the 2-NN of every lapping-area left keypoint of pair blockIdx.y over
// the pair's right lapping rows and the Lowe test (k_fisheye_scan), then the triangulation of the accepted pairs (k_fisheye_tri).
// Round 6: the N x N Hamming scan runs on the matrix pipe.  With the train's bits b in {0, 1} and the query's bits mapped to
// a" = 1 - 2 a in {+1, -1}, a".b = |b| - 2 a.b, so popcount(a ^ b) = |a| + |b| - 2 a.b = |a| + a".b: one integer dot product over
// the 256 bits with the per-query constant as the MFMA's C input.  A workgroup expands a sub-tile of 128 train descriptors ONCE
// into LDS in the A-operand layout of v_mfma_i32_16x16x64_i8 (16 bits -> 16 bytes by one v_mul_u32_u24 per nibble:
// (nib * 0x204081) & 0x01010101); a wave keeps the +-1 bytes of 32 queries as B operands for the whole scan and takes a share of
// every sub-tile's trains: a 16-train x 16-query block of distances is four MFMAs, every A tile read from LDS serves two of them,
// and the D layout (lane = query column, registers = trains 4 g + r) leaves a lane with ONE query per block, so the top-2 update
// is v_lshl_or (key = distance << 16 | train index) + min / max / min per distance where the v_bcnt loop of rounds 1 - 5 spent
// 21 instructions per train (measurements: DESIGN.md 4 / HISTORY.md round 6).
// Keys keep the stable first-minimum order of BFMatcher::knnMatch; the lanes and waves that share a query merge their lists at the
// end (lexicographic keys: any merge order gives the serial scan's result).  The accepted (query, train) pairs go to a compact
// list, so the double-precision triangulation runs with full waves in its own launch instead of one sparse wave per workgroup.
#ifndef FE_ABLATE
#define FE_ABLATE 0
#endif
typedef int orbx_v4i_s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ orbx_v4i_s expand16(uint32_t h) {   // bit j of h -> byte j (0 / 1) of the 16-byte operand
  orbx_v4i_s v;
#pragma unroll
  for (int j = 0; j < 4; j++) v[j] = (int)(__umul24((h >> (4 * j)) & 0xFu, 0x00204081u) & 0x01010101u);
  return v;
}
__device__ __forceinline__ void top2_insert(uint32_t& k0, uint32_t& k1, uint32_t key) {
  const uint32_t lo = min(k0, key);
  k1 = min(k1, max(k0, key));
  k0 = lo;
}
constexpr int kFeSub = 128;                  // trains per expanded sub-tile: eight 16-train groups
constexpr int kFeQ = 128;                    // queries per workgroup: four wave columns of 32
constexpr int kFeTW = 2;                     // train shares: wave (qw, tw) takes groups tw * 8 / kFeTW .. of every sub-tile
constexpr int kFeThreads = 64 * 4 * kFeTW;
constexpr int kFeGW = 8 / kFeTW;             // 16-train groups per wave and sub-tile
constexpr int kFePF = 4;                     // sub-tiles of raw descriptor dwords in flight (even: the LDS buffer is sub-tile & 1)
__global__ __launch_bounds__(kFeThreads, 4) void k_fisheye_scan(FisheyeBatchArgs a) {
  __shared__ uint4 exA[2][8][4][64];          // [buffer][16-train group][k step][lane]: 2 x 32 KB
  __shared__ uint32_t fin[kFeTW][2][kFeQ];    // per train share: the queries' two best keys
  const int pr = blockIdx.y;
  const int imL = a.firstL + pr, imR = a.firstR + pr;
  const int nL = min(a.nL[imL], a.capL), nR = min(a.nR[imR], a.capR);
  const int monoL = min(max(a.monoL[imL], 0), nL), monoR = min(max(a.monoR[imR], 0), nR);
  const int nQ = nL - monoL, nT = nR - monoR;
  if ((int)blockIdx.x * kFeQ >= nQ) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, g = lane >> 4;
  const int qw = w & 3, tw = w >> 2;
  const uint32_t* dQ = reinterpret_cast<const uint32_t*>(a.dL + ((long long)imL * a.capL + monoL) * 32);
  const uint32_t* dT = reinterpret_cast<const uint32_t*>(a.dR + ((long long)imR * a.capR + monoR) * 32);
  orbx_v4i_s bq[2][4], pq[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {   // this lane's two queries (shared by the four lanes n, n + 16, ..)
    const int q = blockIdx.x * kFeQ + 32 * qw + 16 * h + n;
    uint32_t qd[8];
    int pc = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      qd[d] = q < nQ ? dQ[(long long)q * 8 + d] : 0u;
      pc += __popc(qd[d]);
    }
    pq[h] = orbx_v4i_s{pc, pc, pc, pc};
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {   // bits [64 ks + 16 g, + 16) of the query: 1 -> -1 (0xFF), 0 -> +1
      const uint32_t x = q < nQ ? dQ[(long long)q * 8 + 2 * ks + (g >> 1)] : 0u;   // (a second load, not a select of qd[]: no scratch)
      const orbx_v4i_s m = expand16((g & 1) ? (x >> 16) : (x & 0xFFFFu));
#pragma unroll
      for (int j = 0; j < 4; j++) bq[h][ks][j] = (int)(((uint32_t)m[j] * 0xFEu) ^ 0x01010101u);
    }
  }
  uint32_t k0[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, k1[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};  // (distance << 16 | train index), nT < 65536
#if FE_ABLATE == 5   // timing only: no sub-tiles (the workgroup's fixed part alone)
  const int nSub = nT < 0 ? 1 : 0;
#else
  const int nSub = (nT + kFeSub - 1) / kFeSub;
#endif
  // the workgroup expands dword e >> 7 (wave-uniform) of train e & 127 of the sub-tile, e = tid (+ kFeThreads)
  // The raw dwords travel kFePF sub-tiles ahead in registers: a sub-tile's compute (a few hundred cycles) is far shorter than the
  // latency of its loads (descriptors written by another XCD's k_describe come from the memory side), and one workgroup or two
  // per CU cannot hide it by occupancy.
  constexpr int kItems = 1024 / kFeThreads;
  uint32_t nx[kFePF][kItems];
  auto load_sub = [&](int sIdx, uint32_t (&x)[kItems]) {   // (branch-free: rows behind nT re-read the last train and are penalised below)
#pragma unroll
    for (int u = 0; u < kItems; u++) {
      const int e = tid + u * kFeThreads, t = min(sIdx * kFeSub + (e & 127), nT - 1);
      x[u] = dT[(long long)t * 8 + (e >> 7)];
    }
  };
  auto expand_sub = [&](const uint32_t (&x)[kItems], int buf) {
#if FE_ABLATE != 2   // (2: timing only, no expansion)
#pragma unroll
    for (int u = 0; u < kItems; u++) {
      const int e = tid + u * kFeThreads, ei = e & 127, ed = e >> 7;
      const orbx_v4i_s lo = expand16(x[u] & 0xFFFFu), hi = expand16(x[u] >> 16);
      uint4* dst = &exA[buf][ei >> 4][ed >> 1][32 * (ed & 1) + (ei & 15)];
      dst[0] = make_uint4((uint32_t)lo[0], (uint32_t)lo[1], (uint32_t)lo[2], (uint32_t)lo[3]);
      dst[16] = make_uint4((uint32_t)hi[0], (uint32_t)hi[1], (uint32_t)hi[2], (uint32_t)hi[3]);
    }
#endif
  };
  if (nSub > 0) {
#pragma unroll
    for (int j = 0; j < kFePF; j++) load_sub(j, nx[j]);
    expand_sub(nx[0], 0);
    load_sub(kFePF, nx[0]);
  }
  // Iteration si: barrier; the groups of sub-tile si from buffer si & 1 and, in the same basic block, the expansion of sub-tile
  // si + 1 into the other buffer (its vector work fills the MFMA chains' shadows instead of a phase of its own between two
  // barriers); ring slot (si + 1) % kFePF is then re-loaded with sub-tile si + 1 + kFePF.  One barrier per sub-tile is enough: a
  // wave that writes buffer b in iteration si has passed iteration si's barrier, behind every read of b (iteration si - 1).
  for (int sb = 0; sb < nSub; sb += kFePF) {
#pragma unroll
    for (int j = 0; j < kFePF; j++) {
      const int si = sb + j;
      if (si >= nSub) break;
      const int buf = j & 1, t0 = si * kFeSub;   // (kFePF is even)
      uint32_t (&slot)[kItems] = nx[(j + 1) % kFePF];
      __syncthreads();
      // The wave's kFeGW groups of this sub-tile, branch-free so that the scheduler can overlap a group's LDS reads and its two
      // independent MFMA chains with the previous group's top-2 updates.  A sub-tile that crosses nT (the scan's last) adds
      // 0x4000 to the distances of the rows behind nT through the C input: such keys never beat a train's.
      auto groups = [&](auto partial) {
        const uint4* src = &exA[buf][kFeGW * tw][0][lane];
        orbx_v4i_s av4[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
          const uint4 av = src[ks * 64];
          av4[0][ks] = orbx_v4i_s{(int)av.x, (int)av.y, (int)av.z, (int)av.w};
        }
#pragma unroll
        for (int u = 0; u < kFeGW; u++) {
          if (u + 1 < kFeGW) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
              const uint4 av = src[((u + 1) * 4 + ks) * 64];
              av4[(u + 1) & 1][ks] = orbx_v4i_s{(int)av.x, (int)av.y, (int)av.z, (int)av.w};
            }
          }
          const int tg = t0 + 16 * (kFeGW * tw + u);   // the group's first train
          uint32_t tr[4];
#pragma unroll
          for (int r = 0; r < 4; r++) tr[r] = (uint32_t)(tg + 4 * g + r);
          orbx_v4i_s acc[2] = {pq[0], pq[1]};
          if constexpr (decltype(partial)::value) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int pen = (int)tr[r] < nT ? 0 : 0x4000;
              acc[0][r] += pen;
              acc[1][r] += pen;
            }
          }
#pragma unroll
          for (int ks = 0; ks < 4; ks++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
#if FE_ABLATE == 4   // timing only: no MFMA
              acc[h][ks] ^= av4[u & 1][ks][0] ^ av4[u & 1][ks][1] ^ av4[u & 1][ks][2] ^ av4[u & 1][ks][3] ^ bq[h][ks][0];
#else
              acc[h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av4[u & 1][ks], bq[h][ks], acc[h], 0, 0, 0);
#endif
            }
          }
          // acc[h][r] = |a| + a" . b = the distance of query (h, n) and train tg + 4 g + r
#pragma unroll
          for (int h = 0; h < 2; h++) {
#if FE_ABLATE == 3   // timing only: no top-2 epilogue
            k0[h] ^= (uint32_t)(acc[h][0] + acc[h][1] + acc[h][2] + acc[h][3]);
            continue;
#endif
#pragma unroll
            for (int r = 0; r < 4; r++) top2_insert(k0[h], k1[h], ((uint32_t)acc[h][r] << 16) | tr[r]);
          }
        }
        expand_sub(slot, buf ^ 1);              // sub-tile si + 1 (behind the scan's end: a re-read of the last rows, never used)
        load_sub(si + 1 + kFePF, slot);
      };
      if (t0 + kFeSub <= nT) groups(std::false_type{});
      else groups(std::true_type{});
    }
  }
  // the four lanes (n, g) of a query merge their lists, then the train shares through LDS
#pragma unroll
  for (int h = 0; h < 2; h++) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const uint32_t o0 = (uint32_t)__shfl_xor((int)k0[h], o), o1 = (uint32_t)__shfl_xor((int)k1[h], o);
      top2_insert(k0[h], k1[h], o0);
      top2_insert(k0[h], k1[h], o1);
    }
    if (lane < 16) {
      fin[tw][0][32 * qw + 16 * h + lane] = k0[h];
      fin[tw][1][32 * qw + 16 * h + lane] = k1[h];
    }
  }
  __syncthreads();
  if (w >= kFeQ / 64) return;
  const int ql = 64 * w + lane, qf = blockIdx.x * kFeQ + ql;
  uint32_t f0 = fin[0][0][ql], f1 = fin[0][1][ql];
#pragma unroll
  for (int t = 1; t < kFeTW; t++) {
    top2_insert(f0, f1, fin[t][0][ql]);
    top2_insert(f0, f1, fin[t][1][ql]);
  }
  const int b0 = (int)(f0 >> 16), b1 = (int)(f1 >> 16);
  const bool desc = qf < nQ && b1 < 0x4000 && (double)(float)b0 < __dmul_rn((double)(float)b1, 0.7);  // src/Frame.cc:1302
  const uint64_t md = __ballot(desc);
  if (md == 0) return;
  int base = 0;
  if (lane == 0) {
    atomicAdd(a.counters + 2 * pr + 1, __popcll(md));
    base = atomicAdd(a.candCount, __popcll(md));
  }
  base = __shfl(base, 0);
  if (desc)
    a.cand[base + __popcll(md & ((1ull << lane) - 1))] =
        make_uint2((uint32_t)pr, (uint32_t)(qf + monoL) | ((uint32_t)((int)(f0 & 0xFFFF) + monoR) << 16));
}

// Triangulation of the accepted pairs (src/Frame.cc:1302-1316): one lane per entry of the compact list.
__global__ __launch_bounds__(64) void k_fisheye_tri(FisheyeBatchArgs a) {
  const int nc = *a.candCount;
  for (int i0 = blockIdx.x * 64; i0 < nc; i0 += gridDim.x * 64) {
  const int idx = i0 + threadIdx.x;
  bool matched = false;
  int pr = -1;
  if (idx < nc) {
    const uint2 c = a.cand[idx];
    pr = (int)c.x;
    const int iL = (int)(c.y & 0xFFFF), iR = (int)(c.y >> 16);
    const int imL = a.firstL + pr, imR = a.firstR + pr;
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
  // nMatches per pair: the list is not grouped by pair, so a wave counts the pair of its first matched lane, then the next ..
  uint64_t mm = __ballot(matched);
  while (mm) {
    const int lead = __ffsll((long long)mm) - 1;
    const int p0 = __shfl(pr, lead);
    const uint64_t same = __ballot(matched && pr == p0);
    if ((int)threadIdx.x == lead) atomicAdd(a.counters + 2 * p0, __popcll(same));
    mm &= ~same;
  }
  }
}

// Frame::isInFrustum for stereo-fisheye frames (src/Frame.cc:689-697): isInFrustumChecks (:1333-1410) for the left and the right
// camera of every (map point, frame), with KannalaBrandt8::project; float in the reference's expression order (device atan2f / cosf /
// sinf / logf: tolerance parity).  Writes the two view lists the fisheye matcher consumes: the left camera's (proj_xr =
// mTrackProjXR) and the right camera's (proj_x / proj_y / view_cos / predicted_level / in_view of the right check, the rest copied).
__global__ __launch_bounds__(256) void k_project_map_kb8(MapProjKb8Args a) {
  const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
  if (i >= a.n) return;
  const uint8_t fl = a.flags[i];
  const uint4* d4 = reinterpret_cast<const uint4*>(a.desc + (size_t)i * 32);
  const uint4 da = d4[0], db = d4[1];
  const float Px = a.pos[3 * i], Py = a.pos[3 * i + 1], Pz = a.pos[3 * i + 2];
  const float nx = a.normal[3 * i], ny = a.normal[3 * i + 1], nz = a.normal[3 * i + 2];
  const float maxDist = a.maxDist[i], minDist = a.minDist[i];
  const bool offered = !(a.skip && a.skip[(size_t)f * a.n + i]);
  struct Cam { float u, v, viewCos, depth; int lvl; bool ok; };
  auto check = [&](const orbx_frame_pose_kb8& T) {
    Cam c{0.f, 0.f, 0.f, 0.f, 0, offered};
    const float X = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.R[0], Px), __fmul_rn(T.R[1], Py)), __fmul_rn(T.R[2], Pz)), T.t[0]);
    const float Y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.R[3], Px), __fmul_rn(T.R[4], Py)), __fmul_rn(T.R[5], Pz)), T.t[1]);
    const float Z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(T.R[6], Px), __fmul_rn(T.R[7], Py)), __fmul_rn(T.R[8], Pz)), T.t[2]);
    const float pcDist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(X, X), __fmul_rn(Y, Y)), __fmul_rn(Z, Z)));
    c.ok = c.ok && !(Z < 0.0f);
    KB8Cam cam;
#pragma unroll
    for (int k = 0; k < 8; k++) cam.p[k] = T.kb8[k];
    cam.precision = 0.f;
    const float Pc[3] = {X, Y, Z};
    float uv[2];
    kb8_project(cam, Pc, uv);
    c.ok = c.ok && !(uv[0] < a.minX || uv[0] > a.maxX) && !(uv[1] < a.minY || uv[1] > a.maxY);
    const float ox = __fsub_rn(Px, T.twc[0]), oy = __fsub_rn(Py, T.twc[1]), oz = __fsub_rn(Pz, T.twc[2]);
    const float dist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(ox, ox), __fmul_rn(oy, oy)), __fmul_rn(oz, oz)));
    const float maxD = __fmul_rn(1.2f, maxDist), minD = __fmul_rn(0.8f, minDist);
    c.ok = c.ok && !(dist < minD || dist > maxD);
    const float viewCos = __fdiv_rn(__fadd_rn(__fadd_rn(__fmul_rn(ox, nx), __fmul_rn(oy, ny)), __fmul_rn(oz, nz)), dist);
    c.ok = c.ok && !(viewCos < a.viewCosLimit);
    int lvl = (int)ceilf(__fdiv_rn(logf(__fdiv_rn(maxDist, dist)), a.logScaleFactor));
    lvl = lvl < 0 ? 0 : (lvl >= a.nlevels ? a.nlevels - 1 : lvl);
    c.u = uv[0]; c.v = uv[1]; c.viewCos = viewCos; c.depth = pcDist; c.lvl = lvl;
    return c;
  };
  const Cam L = check(a.posesL[f]), R = check(a.posesR[f]);
  const uint32_t flagsW = ((uint32_t)(fl & 1) << 8) | ((uint32_t)((fl >> 1) & 1) << 16);
  const uint32_t xr = R.ok ? __builtin_bit_cast(uint32_t, R.u) : 0u, depthL = L.ok ? __builtin_bit_cast(uint32_t, L.depth) : 0u;
  uint32_t* o = reinterpret_cast<uint32_t*>(a.viewsL + (size_t)f * a.n + i);   // 60-byte records: 15 dwords
  o[0] = L.ok ? __builtin_bit_cast(uint32_t, L.u) : 0xBF800000u; o[1] = L.ok ? __builtin_bit_cast(uint32_t, L.v) : 0xBF800000u;   // (-1.f: k_project_map)
  o[2] = xr; o[3] = L.ok ? __builtin_bit_cast(uint32_t, L.viewCos) : 0u; o[4] = depthL; o[5] = L.ok ? (uint32_t)L.lvl : 0u;
  o[6] = (L.ok ? 1u : 0u) | flagsW;
  o[7] = da.x; o[8] = da.y; o[9] = da.z; o[10] = da.w; o[11] = db.x; o[12] = db.y; o[13] = db.z; o[14] = db.w;
  uint32_t* r = reinterpret_cast<uint32_t*>(a.viewsR + (size_t)f * a.n + i);
  r[0] = xr; r[1] = R.ok ? __builtin_bit_cast(uint32_t, R.v) : 0u; r[2] = xr; r[3] = R.ok ? __builtin_bit_cast(uint32_t, R.viewCos) : 0u;
  r[4] = depthL; r[5] = R.ok ? (uint32_t)R.lvl : 0u;
  r[6] = (R.ok ? 1u : 0u) | flagsW;
  r[7] = da.x; r[8] = da.y; r[9] = da.z; r[10] = da.w; r[11] = db.x; r[12] = db.y; r[13] = db.z; r[14] = db.w;
}
hipError_t launch_project_map_kb8(const MapProjKb8Args& a, int nFrames, hipStream_t s) {
  if (a.n > 0 && nFrames > 0) hipLaunchKernelGGL(k_project_map_kb8, dim3((a.n + 255) / 256, nFrames), dim3(256), 0, s, a);
  return hipGetLastError();
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
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.candCount = 0;
}

hipError_t launch_fisheye_batch(const FisheyeBatchArgs& a, int npairs, hipStream_t s) {
  const long long work = (long long)npairs * (a.capL > a.capR ? a.capL : a.capR) * 3;
  hipLaunchKernelGGL(k_fisheye_init, dim3((unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048)), dim3(256), 0, s, a,
                     npairs);
  hipLaunchKernelGGL(k_fisheye_scan, dim3((a.capL + kFeQ - 1) / kFeQ, npairs), dim3(kFeThreads), 0, s, a);
  hipLaunchKernelGGL(k_fisheye_tri, dim3((unsigned)std::min<long long>(((long long)npairs * a.capL + 63) / 64, 512)), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_fisheye_triangulate(const FisheyeArgs& a, hipStream_t s) {
  const int nQ = a.nL - a.monoL;
  if (nQ <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_fisheye_triangulate, dim3((nQ + 63) / 64), dim3(64), 0, s, a);
  return hipGetLastError();
}

}  // namespace orbx
