// orbx_host.h — host-side internals shared by the C-ABI translation units of liborbx (orbx_api*.hip): error state, device
// buffers, the per-call scratch pool and packed staging, and the extractor handle.
#ifndef ORBX_HOST_H
#define ORBX_HOST_H
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
static inline void cpu_pause() { _mm_pause(); }
#else
static inline void cpu_pause() {}
#endif
#include <memory>
#include <new>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "orbx_internal.h"

using namespace orbx;

namespace orbx_host {


extern thread_local std::string g_err;  // defined in orbx_api.hip
void build_coefs(const Geom& g, std::vector<uint4>& xtab, std::vector<int>& yofs, std::vector<short>& yab);   // orbx_api.hip: k_resize's tables
inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
// ORBX_TRACE_SLOW=<ms>: report every wrapped HIP call / launch sequence that blocks the host longer than that (debug aid
// for enqueue stalls; one getenv at first use, two clock reads per call only when it is set).
inline double trace_slow_ms() {
  static const double v = getenv("ORBX_TRACE_SLOW") ? atof(getenv("ORBX_TRACE_SLOW")) : 0.0;
  return v;
}
#define HIPC(expr)                                                                                     \
  do {                                                                                                 \
    const double lim_ = trace_slow_ms();                                                               \
    const auto t0_ = lim_ > 0 ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point(); \
    hipError_t e_ = (expr);                                                                            \
    if (lim_ > 0) {                                                                                    \
      const double ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count(); \
      if (ms_ > lim_) std::fprintf(stderr, "ORBX_TRACE_SLOW %.2f ms: %s (%s:%d)\n", ms_, #expr, __FILE__, __LINE__); \
    }                                                                                                  \
    if (e_ != hipSuccess)                                                                              \
      return fail(ORBX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
  } while (0)

inline int cv_round(float v) { return (int)lrintf(v); }
inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil(float v) { int i = (int)v; return i + (i < v); }
inline short sat_short(float v) {
  int i = cv_round(v);
  return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}
inline int align_up(long long v, int a) { return (int)((v + a - 1) / a * a); }

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    free();
    n = count;
    if (!count) return hipSuccess;
    return hipMalloc((void**)&p, count * sizeof(T));
  }
  hipError_t grow(size_t count) { return count <= n ? hipSuccess : alloc(count); }   // keeps a large enough buffer
  void free() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// Scratch memory of the one-shot matcher entry points (orbx_bf_knn2, orbx_search_*, orbx_fisheye_stereo_match ...):
// they need a dozen small device buffers per call, and hipMalloc / hipFree cost more than their kernels.  Blocks are
// cached per device (size classes: powers of two) and reused; at most kScratchCap bytes stay cached per device.
class ScratchPool {
 public:
  static void* take(size_t bytes, size_t* granted) {
    size_t cls = 4096;
    while (cls < bytes) cls <<= 1;
    *granted = cls;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    {
      std::lock_guard<std::mutex> lk(mu());
      auto& fl = lists()[dev];
      for (size_t i = 0; i < fl.size(); i++)
        if (fl[i].first == cls) {
          void* p = fl[i].second;
          fl[i] = fl.back();
          fl.pop_back();
          cached()[dev] -= cls;
          return p;
        }
    }
    void* p = nullptr;
    if (hipMalloc(&p, cls) != hipSuccess) return nullptr;
    return p;
  }
  static void give(void* p, size_t cls) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDev) {
      std::lock_guard<std::mutex> lk(mu());
      if (cached()[dev] + cls <= kScratchCap) {
        lists()[dev].push_back({cls, p});
        cached()[dev] += cls;
        return;
      }
    }
    (void)hipFree(p);
  }

 private:
  static constexpr int kMaxDev = 64;
  static constexpr size_t kScratchCap = 256u << 20;
  static std::mutex& mu() { static std::mutex m; return m; }
  static std::vector<std::pair<size_t, void*>>* lists() { static std::vector<std::pair<size_t, void*>> l[kMaxDev]; return l; }
  static size_t* cached() { static size_t c[kMaxDev] = {0}; return c; }
};

template <class T>
struct ScratchBuf {  // same face as DevBuf; the caller has made its device current (set_device)
  T* p = nullptr;
  size_t n = 0, cls = 0;
  hipError_t alloc(size_t count) {
    free();
    n = count;
    if (!count) return hipSuccess;
    p = static_cast<T*>(ScratchPool::take(count * sizeof(T), &cls));
    return p ? hipSuccess : hipErrorOutOfMemory;
  }
  void free() {
    if (p) ScratchPool::give(p, cls);
    p = nullptr;
    n = 0;
  }
};

// One-shot entry points move a dozen small arrays per call; a synchronous hipMemcpy from pageable memory costs ~10 us
// each, more than the kernels.  Pack lays the inputs (and the scratch / output areas) of a call out in ONE device block,
// stages the inputs through a per-thread pinned buffer and uploads them with one asynchronous copy on the null stream,
// in front of the kernels; outputs come back the same way (one copy of a contiguous output area, then memcpy out).
class Pack {
 public:
  // reserves `bytes` (256-byte aligned); src != nullptr: filled from the host.  All inputs must be added before any
  // scratch / output area so that one prefix copy covers them.
  // copyBytes < bytes: only that prefix of the area is read from the host (the rest is reserved, e.g. the unused tail of a
  // strided per-frame array whose last frame the caller need not have padded)
  size_t add(const void* src, size_t bytes, size_t copyBytes = (size_t)-1) {
    const size_t off = (total_ + 255) & ~(size_t)255;
    items_.push_back({src, src ? (copyBytes < bytes ? copyBytes : bytes) : bytes, off});
    total_ = off + bytes;
    if (src && bytes) inputEnd_ = total_;
    return off;
  }
  // allocates the device block for everything added so far: ptr<>() is valid from here on, while the input items' host
  // sources are still only read by commit() -- an argument block that holds device pointers INTO the pack can be filled in
  // between (the batched matchers).  No item may be added after it.
  hipError_t reserve() {
    if (dev_.p) return hipSuccess;
    return dev_.alloc(std::max<size_t>(total_, 256));
  }
  hipError_t commit() {
    hipError_t e = reserve();
    if (e != hipSuccess) return e;
    if (inputEnd_) {
      uint8_t* h = pinned(inputEnd_);
      if (!h) return hipErrorOutOfMemory;
      for (const Item& it : items_)
        if (it.src && it.bytes) std::memcpy(h + it.off, it.src, it.bytes);
      e = hipMemcpyAsync(dev_.p, h, inputEnd_, hipMemcpyHostToDevice, nullptr);
      pending_ = true;
    }
    return e;
  }
  template <class T>
  T* ptr(size_t off) const { return reinterpret_cast<T*>(dev_.p + off); }
  // device [off, off + bytes) -> pinned staging; synchronises the null stream.  The returned pointer is valid until the
  // thread's next Pack operation.
  const uint8_t* fetch(size_t off, size_t bytes, hipError_t* e) {
    uint8_t* h = pinned(std::max<size_t>(bytes, 1));
    if (!h) { *e = hipErrorOutOfMemory; return nullptr; }
    *e = hipMemcpyAsync(h, dev_.p + off, bytes, hipMemcpyDeviceToHost, nullptr);
    if (*e == hipSuccess) *e = hipStreamSynchronize(nullptr);
    if (*e == hipSuccess) pending_ = false;
    return h;
  }
  void release() {  // an error path may leave the upload in flight: the staging buffer is reused by the thread's next call
    if (pending_) (void)hipStreamSynchronize(nullptr);
    pending_ = false;
    dev_.free();
  }

 private:
  struct Item { const void* src; size_t bytes, off; };
  static uint8_t* pinned(size_t bytes) {
    thread_local uint8_t* buf = nullptr;
    thread_local size_t cap = 0;
    if (bytes > cap) {
      if (buf) (void)hipHostFree(buf);
      buf = nullptr;
      cap = 0;
      size_t want = 1 << 20;
      while (want < bytes) want <<= 1;
      if (hipHostMalloc(reinterpret_cast<void**>(&buf), want, hipHostMallocDefault) != hipSuccess) return nullptr;
      cap = want;
    }
    return buf;
  }
  std::vector<Item> items_;
  size_t total_ = 0, inputEnd_ = 0;
  bool pending_ = false;
  ScratchBuf<uint8_t> dev_;
};


int set_device(int device);  // makes `device` current; ORBX_E_NODEVICE without a GPU (there is no host path)

}  // namespace orbx_host
using namespace orbx_host;


// Layout of orbx_extractor::hostResults for the host entry points (one or two images):
//   [0,16)  counts[2], mono[2]   | keypoints 2 x cap | descriptors 2 x cap x 32 | uRight cap | depth cap
static inline size_t hr_flag() { return 32; }   // sequence number of the frame whose counts / keypoints / descriptors are in the block
static inline size_t hr_kps(size_t) { return 64; }
static inline size_t hr_desc(size_t cap) { return hr_kps(cap) + 2 * cap * sizeof(orbx_keypoint); }
static inline size_t hr_ur(size_t cap) { return hr_desc(cap) + 2 * cap * 32; }
static inline size_t hr_depth(size_t cap) { return hr_ur(cap) + cap * sizeof(float); }
static inline size_t host_results_bytes(size_t cap) { return hr_depth(cap) + cap * sizeof(float) + 64; }

struct orbx_extractor {
  orbx_params prm{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  int maxW = 0, maxH = 0, maxB = 0;
  std::vector<float> scale, inv, sig2, invsig2;
  std::vector<int> nfeat;
  int umax[16];
  Geom g{};
  Geom gmax{};
  int curW = 0, curH = 0;
  Pyr pyr{};
  int lastN = 0;
  DevBuf<uint8_t> d_pyr, d_blur, d_stage, d_desc;
  DevBuf<uint8_t> d_dbgScore;      // test tap (orbx_debug_score_map): FAST scores at iniThFAST, pyramid layout; normally unallocated
  DevBuf<uint32_t> d_cand, d_cellCand, d_sel;
  DevBuf<uint16_t> d_knode;
  DevBuf<int> d_rowStart, d_cellCount, d_cellPrefix, d_candCount, d_selCount, d_slot, d_nOut, d_mono, d_lap, d_yofs, d_sad;
  DevBuf<short> d_yab;
  DevBuf<uint4> d_xtab;  // k_resize's per-column table (build_coefs)
  DevBuf<int> d_packCtr;    // arrival counter of the single-frame gather workgroups (launch_stereo_match)
  uint32_t packSeq = 0;     // sequence number of the last single stereo frame (hostResults + hr_flag())
  bool packCtrDirty = false; // a fused single-frame launch is (or may be, after a failure) in flight: d_packCtr is cleared before the next one
  DevBuf<uint32_t> d_yrow;  // k_resize's per-row table: clamped source row pair of every destination row (u16 halves)
  DevBuf<uint4> d_srec, d_sdesc;   // row-sorted keypoint records / descriptors of both eyes (k_stereo_sort)
  std::vector<orbx::TailPlan> tails;  // fused small-level resize segments, in level order (empty: every level through k_resize)
  DevBuf<orbx::TailBand> d_tailBands;
  std::vector<orbx::TailPlan> latTails;  // single-frame plans: the whole chain from level 0 as cascade launches (build_latency_plans)
  DevBuf<orbx::TailBand> d_latBands;
  DevBuf<orbx_keypoint> d_kps;
  DevBuf<float> d_uR, d_depth;
  int stagePitch = 0;
  int stereoPairs = 0;             // high-water allocation of d_uR / d_depth / d_sad (pairs)
  int lastStereoPairs = 0;         // pairs of the stereo association run since the last extraction (0: none)
  // device-resident local map (orbx_map_upload) and the views orbx_project_map_points_batch made of it
  DevBuf<float> d_mapPos, d_mapNormal, d_mapMinD, d_mapMaxD;
  DevBuf<uint8_t> d_mapDesc, d_mapFlags, d_mapSkip;
  DevBuf<orbx_frame_pose> d_poses;
  DevBuf<orbx_map_point_view> d_views;
  // LastFrames of the batch's cameras (orbx_last_frames_upload) and their device-made projections (orbx_project_last_frames_batch)
  DevBuf<float> d_lfPos, d_lfAngle;
  DevBuf<int> d_lfOct, d_lfN;
  DevBuf<uint8_t> d_lfDesc, d_lfFlags;
  DevBuf<orbx_frame_pose_q> d_posesQ;
  DevBuf<orbx_projected_point> d_pviews;
  DevBuf<float> d_scaleF;          // mvScaleFactors for k_project_last
  DevBuf<orbx_frame_pose_kb8> d_posesK;                 // stereo-fisheye device projection (orbx_project_map_points_fisheye_batch)
  DevBuf<orbx_map_point_view> d_fviewsL, d_fviewsR;     // its two view lists
  int fviewsFrames = 0, fviewsStride = 0;
  int lfFrames = 0, lfStride = 0, pviewsFrames = 0;
  std::vector<int> lfN;
  int mapN = 0, viewsFrames = 0, viewsStride = 0;
  uint8_t* hostResults = nullptr;  // pinned: results of up to two images land here with async copies and ONE sync
  int hostResImages = 0;           // images of the last single-frame host entry in that block (orbx_host_results)
  bool hostResStereo = false;      // ... and whether uRight / depth of image 0 are there
  uint8_t* hostPyr = nullptr;      // pinned staging of orbx_pyramid_download (one image's pyramid), allocated on first use
  size_t hostPyrBytes = 0;
  // single-frame host entries (orbx_extract / orbx_extract_stereo): with orbx_set_host_pyramid the levels are copied to
  // page-locked memory on streamPyr, behind evPyr (recorded once the pyramids exist), beside the frame's kernels
  bool useLat = false;             // this extraction's resize chain = the cascade plans (latTails)
  hipStream_t streamPyr = nullptr;
  hipEvent_t evPyr = nullptr;
  bool keepHostPyr = false;        // orbx_set_host_pyramid
  uint8_t* hostPyrAll = nullptr;   // pinned: [2] x {level 0 (stagePitch x maxH), levels 1.. (gmax.pyrImg)}
  size_t hostPyrAllBytes = 0;
  int hostPyrImages = 0;           // images of the LAST extraction whose levels are in hostPyrAll (0: none)
  int hostPyrL0Pitch = 0;          // row pitch of level 0 in that copy (= the staging pitch of the extraction)
  int32_t* h_lap = nullptr;        // pinned copy of the lapping areas the device currently holds (lapN images)
  int lapN = 0;
  // hipGraph of the single-image pipeline (host API orbx_extract): index = lapTrivial; valid for (graphW, graphH)
  hipGraphExec_t graphExec[2] = {nullptr, nullptr};
  int graphW = 0, graphH = 0;
  bool graphOff = false;
  // bag of words of the last extraction (orbx_bow_transform_batch): per-feature word / weight / node, assembled vectors
  DevBuf<int> d_bowWord, d_bowNode, d_bowStart, d_bowCounts;
  DevBuf<double> d_bowWeight, d_bowValues;
  DevBuf<uint32_t> d_bowWords, d_bowNodes, d_bowFeats;
  int bowImages = 0;
  DevBuf<int> d_fl2r, d_fr2l, d_fcnt, d_fcand;  // batched fisheye association (orbx_fisheye_stereo_match_batch)
  DevBuf<float> d_fdepth, d_fp3d;
  int fisheyePairs = 0, fisheyeCapR = 0;
  // per-launch HIP event log (orbx_profile_*)
  bool profiling = false;
  int profStage = -1;              // >= 0: only launches of this stage are bracketed
  std::vector<hipEvent_t> evPool;
  size_t evCursor = 0;
  struct EvRec { int stage; size_t e0, e1; };
  std::vector<EvRec> evLog;
  size_t lastEv = 0;
  bool lastEvValid = false;
  int cv440 = 0;                   // orbx_set_opencv_compat: 1 / 16 / 32 = Gaussian taps of OpenCV 4.0 .. 4.5.0 (Geom::cv440)
  bool blurValid = false;          // d_blur holds the blurred levels of the last extraction (filled on demand)
  hipEvent_t next_event() {
    if (evCursor == evPool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      evPool.push_back(e);
    }
    return evPool[evCursor++];
  }
};

namespace orbx_host {
// the whole extraction pipeline of n device-resident images on ex->stream (orbx_api.hip)
int enqueue_extract(orbx_extractor* ex, const uint8_t* d_images, int n, int w, int h, ptrdiff_t row_pitch,
                    ptrdiff_t image_pitch, const int32_t* lap);
}  // namespace orbx_host
#endif
