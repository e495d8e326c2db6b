// orbx_stereo.hip — association kernels of liborbx: Frame::ComputeStereoMatches (src/Frame.cc:921-1084), BFMatcher kNN-2
// (:1293-1302) and ComputeStereoFishEyeMatches with the KannalaBrandt8 triangulation (:1273-1331).
#include "orbx_device.h"

namespace orbx {

// ================================================================================================ stereo

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
  const int iL = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // one left keypoint per wave: SGPR
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
  float bestX = 0.f;             // u of this lane's best candidate (saves the dependent kR[best] load after the reduction)
  if (!(maxU < 0)) {
    const int* rowStart = a.rowStart + (long long)pair * (a.imgH + 1);
    const int* items = a.rowItems + (long long)pair * a.capR;
    const int jb = rowStart[min(max(row - a.band, 0), a.imgH)], je = rowStart[min(max(row + a.band + 1, 0), a.imgH)];
    for (int base = jb; base < je; base += 64) {
      const int j = base + lane;
      if (j < je) {
        const int iR = items[j];
        const orbx_keypoint k = kR[iR];
        // the descriptor row is fetched together with the keypoint (both depend only on iR): one memory round trip
        // instead of two on the kernel's critical path (it is latency-, not bandwidth-bound)
        const uint4* dq = reinterpret_cast<const uint4*>(dR + (long long)iR * 8);
        const uint4 q0 = dq[0], q1 = dq[1];
        const float r = __fmul_rn(2.0f, g.lv[k.octave].scale);
        const int maxr = (int)ceilf(__fadd_rn(k.y, r)), minr = (int)floorf(__fsub_rn(k.y, r));
        const bool ok = !(k.y == 0.0f && k.x == 0.0f) && row >= minr && row <= maxr &&
                        k.octave >= levelL - 1 && k.octave <= levelL + 1 && k.x >= minU && k.x <= maxU;
        if (ok) {
          const int d = __popc(dl[0] ^ q0.x) + __popc(dl[1] ^ q0.y) + __popc(dl[2] ^ q0.z) + __popc(dl[3] ^ q0.w) +
                        __popc(dl[4] ^ q1.x) + __popc(dl[5] ^ q1.y) + __popc(dl[6] ^ q1.z) + __popc(dl[7] ^ q1.w);
          const uint32_t cand = ((uint32_t)d << 16) | (uint32_t)iR;
          if (cand < best) {
            best = cand;
            bestX = k.x;
          }
        }
      }
    }
  }
  const uint32_t mine = best;
  best = wave_min_dpp(best);
  const int bestDist = (int)(best >> 16);
  if (bestDist < 75) {  // thOrbDist = (TH_HIGH + TH_LOW) / 2
    const uint64_t owners = __ballot(mine == best);  // iR is unique per candidate: exactly the lane(s) that saw it
    const float uR0 = __shfl(bestX, (int)__builtin_ctzll(owners));
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
        sadv[inc] = wave_sum_dpp(s);  // (six v_add_dpp; the 11 sums are independent chains)
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

}  // namespace orbx
