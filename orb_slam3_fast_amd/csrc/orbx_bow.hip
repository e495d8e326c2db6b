// orbx_bow.hip — bag of words: DBoW2 transform and ORBmatcher::SearchByBoW.
#include "orbx_device.h"

namespace orbx {

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
  for (;;) {  // three dependent loads per level: child range, child ids, child descriptors
    int2 cr;
    __builtin_memcpy(&cr, v.childStart + cur, 8);  // childStart[cur], childStart[cur + 1]
    const int c0 = cr.x, c1 = cr.y;
    if (c0 == c1) break;  // leaf (the root of a non-empty vocabulary has children)
    uint32_t best = 0xffffffffu;
    int bestChild = 0;
    for (int c = c0 + sub; c < c1; c += 16) {
      const int child = v.children[c];
      const uint32_t k = ((uint32_t)hamming256(d, v.desc + (long long)child * 8) << 16) | (uint32_t)(c - c0);
      if (k < best) { best = k; bestChild = child; }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 16));
    cur = __shfl(bestChild, (int)(best & 15u), 16);  // position p was scored by lane p mod 16, whose own best it is
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

  // the two vectors are independent: blockIdx.y = 0 assembles the BowVector, 1 the FeatureVector (half the latency)
  if (blockIdx.y == 0) {
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
    if (tid == 0) a.outCounts[img * 3 + 0] = U;
    return;
  }

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
  hipLaunchKernelGGL(k_bow_assemble, dim3(nimg, 2), dim3(kBowThreads), lds, s, a);
  return hipGetLastError();
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:230-404).  Features are only compared inside a shared
// vocabulary node and a frame feature belongs to one node, so the nodes are independent: one wave per keyframe node finds
// its partner in the frame's node list (binary search) and walks the node's keyframe features in order, like the reference
// (the "already matched" gate makes that walk order dependent); its lanes score the node's frame features, best / second
// by the serial rule (first minimum; an equal later distance becomes the second).  The right-eye branch keeps the
// reference's "|| true" (:363-365): no ratio test, and only inside "bestDist1 <= TH_LOW".
constexpr int kBowNodeCap = 4096;  // frame features of one node tracked in LDS (a node above this: serial fallback on lane 0)
// Kernel-argument views: the one-shot call passes its argument block by value; orbx_search_by_bow_batch (round 5) launches every
// kernel ONCE for all frames of an extraction batch with blockIdx.y = frame and the argument blocks in a device array.
struct BowVal {
  BowMatchArgs v;
  __device__ __forceinline__ const BowMatchArgs& get() const { return v; }
};
struct BowOfArr {
  const BowMatchArgs* p;
  __device__ __forceinline__ const BowMatchArgs& get() const { return p[blockIdx.y]; }
};
template <class R>
__global__ __launch_bounds__(64) void k_bow_match(R ar) {
  const BowMatchArgs& a = ar.get();
  __shared__ uint8_t taken[kBowNodeCap];
  const int lane = threadIdx.x, ia = blockIdx.x;
  if (ia >= a.nKfNodes || a.nFNodes <= 0) return;  // (batched launches are sized for the largest key frame)
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
        if (lane == 0) {  // (the rotation votes are cast afterwards by k_bow_vote: a frame feature is matched at most once,
          if (leftOk) a.match[bi] = iKF;    // so the final matches ARE the votes, and the serial walk stays free of
          if (rightOk) a.match[bir] = iKF;  // dependent keypoint loads)
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

template <class R>
__global__ __launch_bounds__(256) void k_bow_vote(R ar) {  // rotHist[bin].push_back(bestIdxF), :336-346 / :367-377
  const BowMatchArgs& a = ar.get();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.nF) return;
  const int iKF = a.match[i];
  if (iKF < 0) return;
  float rot = __fsub_rn(a.kfKps[iKF].angle, a.fKps[i].angle);
  if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
  int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
  if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
  a.bin[i] = bin;
  atomicAdd(&a.flags[2 + bin], 1);
}

template <class R>
__global__ __launch_bounds__(256) void k_bow_cull(R ar) {  // :384-401 with ComputeThreeMaxima :1920-1955
  const BowMatchArgs& a = ar.get();
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
template <class R>
__global__ void k_bow_result(R ar) {
  const BowMatchArgs& a = ar.get();
  a.result[0] = a.flags[32] ? -1 : a.flags[0] - a.flags[1];
}
template <class R>
__global__ __launch_bounds__(256) void k_bow_match_reset(R ar) {
  const BowMatchArgs& a = ar.get();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.nF) a.match[i] = -1;
  if (i < 33) a.flags[i] = 0;
}

hipError_t launch_bow_match(const BowMatchArgs& a, hipStream_t s) {
  const BowVal r{a};
  hipLaunchKernelGGL(k_bow_match_reset<BowVal>, dim3((max(a.nF, 33) + 255) / 256), dim3(256), 0, s, r);
  if (a.nKfNodes > 0 && a.nFNodes > 0) hipLaunchKernelGGL(k_bow_match<BowVal>, dim3(a.nKfNodes), dim3(64), 0, s, r);
  if (a.checkOri && a.nF > 0) {
    hipLaunchKernelGGL(k_bow_vote<BowVal>, dim3((a.nF + 255) / 256), dim3(256), 0, s, r);
    hipLaunchKernelGGL(k_bow_cull<BowVal>, dim3((a.nF + 255) / 256), dim3(256), 0, s, r);
  }
  hipLaunchKernelGGL(k_bow_result<BowVal>, dim3(1), dim3(1), 0, s, r);
  return hipGetLastError();
}
// SearchByBoW(KeyFrame, Frame) for every frame of a batch, one launch per kernel (blockIdx.y = frame): d_frames = device array of
// nFrames argument blocks, maxKfNodes / maxNF the largest key-frame node count / frame keypoint count.
hipError_t launch_bow_match_batch(const BowMatchArgs* d_frames, int nFrames, int maxKfNodes, int maxNF, int checkOri, hipStream_t s) {
  if (nFrames <= 0) return hipSuccess;
  const BowOfArr r{d_frames};
  const dim3 nft((max(maxNF, 33) + 255) / 256, nFrames);
  hipLaunchKernelGGL(k_bow_match_reset<BowOfArr>, nft, dim3(256), 0, s, r);
  if (maxKfNodes > 0) hipLaunchKernelGGL(k_bow_match<BowOfArr>, dim3(maxKfNodes, nFrames), dim3(64), 0, s, r);
  if (checkOri) {
    hipLaunchKernelGGL(k_bow_vote<BowOfArr>, nft, dim3(256), 0, s, r);
    hipLaunchKernelGGL(k_bow_cull<BowOfArr>, nft, dim3(256), 0, s, r);
  }
  hipLaunchKernelGGL(k_bow_result<BowOfArr>, dim3(1, nFrames), dim3(1), 0, s, r);
  return hipGetLastError();
}

// ---- ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:886-1106), single-camera key frames -------------------------------
// vbMatched2 is declared and tested but never set (:933,976), so a feature of pKF1 only depends on the features of pKF2 in its own
// vocabulary node: all features are independent.  The serial scan keeps a candidate when dist <= min(TH_LOW, bestDist) and the
// geometric gates pass, and the gates do not depend on bestDist, so the result is the LAST candidate of minimal distance among the
// gate-passers = the minimum of (dist, -position).  16 lanes per feature of pKF1 (a node holds ~15 features of a 1500-feature
// frame at levelsup 4), wave64 = 4 features.
__global__ __launch_bounds__(256) void k_tri_init(TriArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.n1) a.match[i] = -1;
  if (i < 33) a.flags[i] = 0;
}

__global__ __launch_bounds__(256) void k_tri_match(TriArgs a) {
  const int sub = threadIdx.x & 15;
  const int g = blockIdx.x * 16 + (threadIdx.x >> 4);  // position in pKF1's feature lists
  if (g >= a.nList1) return;
  // node of this list entry: last j with start1[j] <= g; its partner in pKF2's vector (the lower_bound walk of :943-1085 visits
  // exactly the node ids present in both maps)
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
  if (a.mp1[idx1]) return;  // :953
  const bool stereo1 = a.ur1 && a.ur1[idx1] >= 0;
  if (a.onlyStereo && !stereo1) return;
  const orbx_keypoint kp1 = a.k1[idx1];
  // epipolar line of kp1 in image 2, Pinhole.cpp:136-138 (separately rounded, left to right)
  const float la = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[0]), __fmul_rn(kp1.y, a.F[3])), a.F[6]);
  const float lb = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[1]), __fmul_rn(kp1.y, a.F[4])), a.F[7]);
  const float lc = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[2]), __fmul_rn(kp1.y, a.F[5])), a.F[8]);
  const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
  uint32_t d1[8];
#pragma unroll
  for (int i = 0; i < 8; i++) d1[i] = a.d1[(long long)idx1 * 8 + i];
  uint32_t best = 0xFFFFFFFFu;  // (dist << 24) | (0xFFFFFF - position)
  for (int pos = b + sub; pos < e; pos += 16) {
    const int idx2 = (int)a.feat2[pos];
    if (a.mp2[idx2]) continue;  // :976
    const bool stereo2 = a.ur2 && a.ur2[idx2] >= 0;
    if (a.onlyStereo && !stereo2) continue;
    const int dist = hamming256(d1, a.d2 + (long long)idx2 * 8);
    if (dist > 50) continue;  // TH_LOW
    const orbx_keypoint kp2 = a.k2[idx2];
    if (!stereo1 && !stereo2) {  // too close to the epipole, :997-1004
      const float ex = __fsub_rn(a.ep0, kp2.x), ey = __fsub_rn(a.ep1, kp2.y);
      if (__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)) < __fmul_rn(100.0f, a.scale2[kp2.octave])) continue;
    }
    if (!a.coarse) {  // Pinhole::epipolarConstrain, Pinhole.cpp:140-148
      const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, kp2.x), __fmul_rn(lb, kp2.y)), lc);
      if (den == 0) continue;
      const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
      if (!((double)dsqr < 3.84 * (double)a.sigma2[kp2.octave])) continue;
    }
    const uint32_t key = ((uint32_t)dist << 24) | (0xFFFFFFu - (uint32_t)(pos - b));
    best = min(best, key);
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 16));
  if (sub == 0 && best != 0xFFFFFFFFu) {
    const int idx2 = (int)a.feat2[b + (int)(0xFFFFFFu - (best & 0xFFFFFFu))];
    a.match[idx1] = idx2;
    atomicAdd(&a.flags[0], 1);
    if (a.checkOri) {
      float rot = __fsub_rn(kp1.angle, a.k2[idx2].angle);
      if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
      int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
      if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
      atomicAdd(&a.flags[2 + bin], 1);
    }
  }
}

// rotation-consistency cull (:1087-1102, ComputeThreeMaxima :1920-1955) + the count; one workgroup
__global__ __launch_bounds__(1024) void k_tri_cull(TriArgs a) {
  __shared__ int s_removed;
  if (threadIdx.x == 0) s_removed = 0;
  __syncthreads();
  if (a.checkOri) {
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
    int removed = 0;
    for (int i = threadIdx.x; i < a.n1; i += 1024) {
      const int m = a.match[i];
      if (m < 0) continue;
      float rot = __fsub_rn(a.k1[i].angle, a.k2[m].angle);
      if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
      int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
      if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
      if (bin != ind1 && bin != ind2 && bin != ind3) {
        a.match[i] = -1;
        removed++;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) removed += __shfl_xor(removed, o);
    if ((threadIdx.x & 63) == 0 && removed) atomicAdd(&s_removed, removed);
  }
  __syncthreads();
  if (threadIdx.x == 0) a.result[0] = a.flags[32] ? -1 : a.flags[0] - s_removed;
}

// ---- ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (src/ORBmatcher.cc:766-884) ---------
// Like SearchByBoW(KeyFrame*, Frame&) the walk over a node's features of pKF1 is order dependent (a feature of pKF2 is taken at
// most once, vbMatched2), and nodes are independent: one wave per node of pKF1 walks its features in order, the lanes score the
// still-free features of the partner node (first minimum + second smallest by the serial update rule :826-832), the accepted
// feature is marked in LDS.  Differences to the frame flavour: `bestDist1 < TH_LOW` is strict (:836), both sides carry validity
// flags, the result is indexed by pKF1's features and the rotation votes use idx1 (:845-851).
__global__ __launch_bounds__(64) void k_bow_match_kf(TriArgs a) {
  __shared__ uint8_t taken[kBowNodeCap];
  const int lane = threadIdx.x, ia = blockIdx.x;
  const uint32_t node = a.nodes1[ia];
  int lo = 0, hi = a.nNodes2;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a.nodes2[mid] < node) lo = mid + 1; else hi = mid;
  }
  if (lo >= a.nNodes2 || a.nodes2[lo] != node) return;
  const int f0 = a.start2[lo], nfl = a.start2[lo + 1] - f0;
  const int k0 = a.start1[ia], k1 = a.start1[ia + 1];
  if (nfl > kBowNodeCap) {
    if (lane == 0) a.flags[32] = 1;
    return;
  }
  for (int i = lane; i < nfl; i += 64) taken[i] = 0;
  __syncthreads();
  int i20 = -1, v20 = 0;   // the partner node's first 64 features stay in registers for the whole walk
  uint32_t D0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < nfl) {
    i20 = (int)a.feat2[f0 + lane];
    v20 = a.mp2[i20];
    const uint32_t* d = a.d2 + (long long)i20 * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) D0[i] = d[i];
  }
  int made = 0;
  for (int kc = k0; kc < k1; kc += 64) {
    int my1 = -1, myValid = 0;
    uint32_t myD[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kc + lane < k1) {
      my1 = (int)a.feat1[kc + lane];
      myValid = a.mp1[my1];
      if (myValid) {
        const uint32_t* d = a.d1 + (long long)my1 * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) myD[i] = d[i];
      }
    }
    const int cnt = min(64, k1 - kc);
    for (int c = 0; c < cnt; c++) {
      if (!__builtin_amdgcn_readlane(myValid, c)) continue;
      const int idx1 = __builtin_amdgcn_readlane(my1, c);
      uint32_t d[8];
#pragma unroll
      for (int i = 0; i < 8; i++) d[i] = (uint32_t)__builtin_amdgcn_readlane((int)myD[i], c);
      int b1 = 256, bi = -1, bpos = -1, b2 = 256;
      for (int q0 = 0; q0 < nfl; q0 += 64) {
        const int q = q0 + lane;
        int dist = 0x7fff;
        bool mine = false;
        if (q < nfl && !taken[q]) {
          if (q0 == 0) {
            mine = v20 != 0;
            if (mine) dist = hamming256(d, D0);
          } else {
            const int i2 = (int)a.feat2[f0 + q];
            mine = a.mp2[i2] != 0;
            if (mine) dist = hamming256(d, a.d2 + (long long)i2 * 8);
          }
        }
        uint32_t k1st = mine ? ((uint32_t)dist << 8) | (uint32_t)lane : 0xffffffffu;  // first minimum: lower lane = earlier
        for (int o = 32; o > 0; o >>= 1) k1st = min(k1st, (uint32_t)__shfl_xor((int)k1st, o));
        uint32_t k2nd = (mine && (k1st & 255u) != (uint32_t)lane) ? (uint32_t)dist : 0xffffffffu;
        for (int o = 32; o > 0; o >>= 1) k2nd = min(k2nd, (uint32_t)__shfl_xor((int)k2nd, o));
        if (k1st != 0xffffffffu) {
          const int c1 = (int)(k1st >> 8), cl = (int)(k1st & 255u), c2 = k2nd == 0xffffffffu ? 256 : (int)k2nd;
          if (c1 < b1) {
            b2 = min(b1, c2);
            b1 = c1;
            bpos = q0 + cl;
            bi = q0 == 0 ? __builtin_amdgcn_readlane(i20, cl) : (int)a.feat2[f0 + q0 + cl];
          } else {
            b2 = min(b2, c1);
          }
        }
      }
      if (b1 < 50 && (float)b1 < __fmul_rn(a.nnratio, (float)b2)) {  // :836-838
        if (lane == 0) {
          a.match[idx1] = bi;
          taken[bpos] = 1;
          if (a.checkOri) {
            float rot = __fsub_rn(a.k1[idx1].angle, a.k2[bi].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / 30));
            if (bin == 30) bin = 0;
  bin = min(max(bin, 0), 29);  // (angles outside [0, 360) or NaN: the reference asserts; here the vote stays inside the histogram)
            atomicAdd(&a.flags[2 + bin], 1);
          }
        }
        made++;
        __syncthreads();
      }
    }
  }
  if (lane == 0 && made) atomicAdd(&a.flags[0], made);
}

hipError_t launch_search_by_bow_keyframes(const TriArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_tri_init, dim3((max(a.n1, 33) + 255) / 256), dim3(256), 0, s, a);
  if (a.nNodes1 > 0 && a.nNodes2 > 0) hipLaunchKernelGGL(k_bow_match_kf, dim3(a.nNodes1), dim3(64), 0, s, a);
  hipLaunchKernelGGL(k_tri_cull, dim3(1), dim3(1024), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_search_for_triangulation(const TriArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_tri_init, dim3((max(a.n1, 33) + 255) / 256), dim3(256), 0, s, a);
  if (a.rig) {  // two-camera rigs: the match kernel lives beside the KB8 triangulation
    const hipError_t e = launch_tri_match_rig(a, s);
    if (e != hipSuccess) return e;
  } else if (a.nList1 > 0 && a.nNodes2 > 0) hipLaunchKernelGGL(k_tri_match, dim3((a.nList1 + 15) / 16), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_tri_cull, dim3(1), dim3(1024), 0, s, a);
  return hipGetLastError();
}

}  // namespace orbx
