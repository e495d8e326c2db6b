// orbx_api.hip — C ABI of liborbx (include/orbx.h): handles, buffers, geometry, launch sequencing.
// Host-side restatement of the ORBextractor constructor tables (src/ORBextractor.cc:408-469) and of the
// OpenCV resize coefficient tables; all pixel/bit work is in orbx_kernels.hip.
#include <memory>
#include <new>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "orbx_internal.h"

using namespace orbx;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPC(expr)                                                                                     \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
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
  size_t add(const void* src, size_t bytes) {
    const size_t off = (total_ + 255) & ~(size_t)255;
    items_.push_back({src, bytes, off});
    total_ = off + bytes;
    if (src && bytes) inputEnd_ = total_;
    return off;
  }
  hipError_t commit() {
    hipError_t e = dev_.alloc(std::max<size_t>(total_, 256));
    if (e != hipSuccess) return e;
    if (inputEnd_) {
      uint8_t* h = pinned(inputEnd_);
      if (!h) return hipErrorOutOfMemory;
      for (const Item& it : items_)
        if (it.src && it.bytes) std::memcpy(h + it.off, it.src, it.bytes);
      e = hipMemcpyAsync(dev_.p, h, inputEnd_, hipMemcpyHostToDevice, nullptr);
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
    return h;
  }
  void release() { dev_.free(); }

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
  ScratchBuf<uint8_t> dev_;
};

}  // namespace

// Layout of orbx_extractor::hostResults for the host entry points (one or two images):
//   [0,16)  counts[2], mono[2]   | keypoints 2 x cap | descriptors 2 x cap x 32 | uRight cap | depth cap
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
  DevBuf<uint32_t> d_cand, d_cellCand, d_sel;
  DevBuf<uint16_t> d_knode;
  DevBuf<int> d_rowStart, d_rowItems, d_cellCount, d_cellPrefix, d_candCount, d_selCount, d_slot, d_nOut, d_mono, d_lap, d_xofs, d_yofs, d_sad;
  DevBuf<short> d_xab, d_yab;
  DevBuf<orbx_keypoint> d_kps;
  DevBuf<float> d_uR, d_depth;
  int stagePitch = 0;
  int stereoPairs = 0;
  uint8_t* hostResults = nullptr;  // pinned: results of up to two images land here with async copies and ONE sync
  // hipGraph of the single-image pipeline (host API orbx_extract): index = lapTrivial; valid for (graphW, graphH)
  hipGraphExec_t graphExec[2] = {nullptr, nullptr};
  int graphW = 0, graphH = 0;
  bool graphOff = false;
  // bag of words of the last extraction (orbx_bow_transform_batch): per-feature word / weight / node, assembled vectors
  DevBuf<int> d_bowWord, d_bowNode, d_bowStart, d_bowCounts;
  DevBuf<double> d_bowWeight, d_bowValues;
  DevBuf<uint32_t> d_bowWords, d_bowNodes, d_bowFeats;
  int bowImages = 0;
  DevBuf<int> d_fl2r, d_fr2l, d_fcnt;  // batched fisheye association (orbx_fisheye_stereo_match_batch)
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
  hipStream_t stream2 = nullptr;   // side stream: k_blur overlaps detect / quadtree
  hipEvent_t evPyr = nullptr, evBlur = nullptr, evStart = nullptr, evDet0 = nullptr;
  hipEvent_t next_event() {
    if (evCursor == evPool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      evPool.push_back(e);
    }
    return evPool[evCursor++];
  }
};

namespace {
// Brackets one kernel launch with events on the launch stream when profiling is on.  Consecutive launches on
// the main stream share their boundary event (the end of one is the start of the next), which halves the
// number of event records in the timed region.
struct StageTimer {
  orbx_extractor* ex;
  hipStream_t s;
  int stage;
  size_t i0 = 0;
  bool on;
  StageTimer(orbx_extractor* ex_, hipStream_t s_, int stage_)
      : ex(ex_), s(s_), stage(stage_), on(ex_->profiling && (ex_->profStage < 0 || ex_->profStage == stage_)) {
    if (!on) {
      if (s == ex->stream) ex->lastEvValid = false;  // the chain of shared boundary events is broken here
      return;
    }
    if (s == ex->stream && ex->lastEvValid) {
      i0 = ex->lastEv;
    } else {
      hipEvent_t e = ex->next_event();
      i0 = ex->evCursor - 1;
      if (e) (void)hipEventRecord(e, s);
    }
  }
  ~StageTimer() {
    if (!on) return;
    hipEvent_t e = ex->next_event();
    if (e) (void)hipEventRecord(e, s);
    ex->evLog.push_back({stage, i0, ex->evCursor - 1});
    if (s == ex->stream) {
      ex->lastEv = ex->evCursor - 1;
      ex->lastEvValid = true;
    }
  }
};
}  // namespace

namespace {

// ---- tables: ORBextractor::ORBextractor, src/ORBextractor.cc:408-469 --------------------------------
void build_tables(orbx_extractor* ex) {
  const int L = ex->prm.nlevels;
  const double sf = (double)ex->prm.scale_factor;  // member is double, argument float (:106 of the header)
  ex->scale.assign(L, 1.f);
  ex->sig2.assign(L, 1.f);
  for (int i = 1; i < L; i++) {
    ex->scale[i] = (float)(ex->scale[i - 1] * sf);
    ex->sig2[i] = ex->scale[i] * ex->scale[i];
  }
  ex->inv.resize(L);
  ex->invsig2.resize(L);
  for (int i = 0; i < L; i++) {
    ex->inv[i] = 1.0f / ex->scale[i];
    ex->invsig2[i] = 1.0f / ex->sig2[i];
  }
  ex->nfeat.assign(L, 0);
  const float factor = (float)(1.0f / sf);
  float nDesired = ex->prm.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
  int sum = 0;
  for (int l = 0; l < L - 1; l++) {
    ex->nfeat[l] = cv_round(nDesired);
    sum += ex->nfeat[l];
    nDesired *= factor;
  }
  ex->nfeat[L - 1] = std::max(ex->prm.nfeatures - sum, 0);
  int um[16] = {0};
  const float s2 = std::sqrt(2.f);
  const int vmax = cv_floor(kHalfPatch * s2 / 2 + 1), vmin = cv_ceil(kHalfPatch * s2 / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (int v = 0; v <= vmax; ++v) um[v] = cv_round(std::sqrt(hp2 - v * v));
  for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (um[v0] == um[v0 + 1]) ++v0;
    um[v] = v0;
    ++v0;
  }
  std::memcpy(ex->umax, um, sizeof(um));
}

// ---- geometry for one image size (pyramid sizes :1111-1113, cell grid :901-907) ----------------------
int build_geom(const orbx_extractor* ex, int w, int h, Geom& g, std::string& why) {
  std::memset(&g, 0, sizeof(g));
  const int L = ex->prm.nlevels;
  g.nlevels = L;
  g.iniTh = ex->prm.ini_th_fast;
  g.minTh = ex->prm.min_th_fast;
  long long off = 0, candOff = 0, cellOff = 0;
  int cells = 0, sel = 0, xc = 0, yc = 0;
  int maxCW = 0, maxCH = 0;
  for (int l = 0; l < L; l++) {
    LevelDev& v = g.lv[l];
    v.w = cv_round((float)w * ex->inv[l]);
    v.h = cv_round((float)h * ex->inv[l]);
    v.pitch = align_up(v.w, 64);
    v.off = off;
    off += (long long)v.pitch * v.h;
    off = (off + 255) / 256 * 256;
    const int maxBX = v.w - kBorder, maxBY = v.h - kBorder;
    const float width = (float)(maxBX - kBorder), height = (float)(maxBY - kBorder);
    if (width < 35.f || height < 35.f) {
      why = "level " + std::to_string(l) + " is smaller than one 35 px FAST cell (+32 px border)";
      return ORBX_E_UNSUPPORTED;
    }
    v.nCols = (int)(width / 35.f);
    v.nRows = (int)(height / 35.f);
    v.wCell = (int)std::ceil(width / v.nCols);
    v.hCell = (int)std::ceil(height / v.nRows);
    v.cellStart = cells;
    cells += v.nCols * v.nRows;
    v.quota = ex->nfeat[l];
    const int nIni = (int)std::round(width / height);
    if (nIni < 1 || nIni > kMaxIni) {
      why = "unsupported aspect ratio at level " + std::to_string(l);
      return ORBX_E_UNSUPPORTED;
    }
    v.candCap = (((int)width + v.nCols + 1) / 2 + 1) * (((int)height + v.nRows + 1) / 2 + 1);
    v.candOff = candOff;
    candOff += v.candCap;
    v.cellCap = ((v.wCell + 1) / 2) * ((v.hCell + 1) / 2);  // NMS keeps at most one pixel per 2x2 block
    v.cellOff = cellOff;
    cellOff += (long long)v.nCols * v.nRows * v.cellCap;
    v.selOff = sel;
    v.selCap = v.quota + 4 * kMaxIni + 4;  // independent of the image size: fixed result strides
    sel += v.selCap;
    v.xcoef = xc;
    v.ycoef = yc;
    xc += v.w;
    yc += v.h;
    v.scale = ex->scale[l];
    v.patch = (float)(int)(31 * ex->scale[l]);
    maxCW = std::max(maxCW, v.wCell);
    maxCH = std::max(maxCH, v.hCell);
    if (v.w > 4096 || v.h > 4096) {  // selected keys carry level coordinates (<= w - 1) in 12 bits
      why = "images larger than 4096 px are not supported by the 12-bit key packing";
      return ORBX_E_UNSUPPORTED;
    }
  }
  if (maxCW > 250 || maxCH > 120) {
    why = "FAST cell too large for the packed corner list";
    return ORBX_E_UNSUPPORTED;
  }
  g.totalCells = cells;
  g.tileP = 4 * ((maxCW + 3) / 4 + 3);   // quads per row + 2 dwords of read-ahead + 1
  g.tileH = maxCH + 6;
  g.scoreP = 4 * ((maxCW + 3) / 4 + 2);  // 1 dword zero pad left + quads + 1 dword zero pad right
  g.scoreH = maxCH + 2;
  g.listCap = align_up((long long)maxCW * maxCH, 8);
  g.selImg = sel;
  g.outCap = sel;
  g.pyrImg = off;
  g.candImg = candOff;
  g.cellImg = cellOff;
  if (octree_lds_bytes(g) > 160 * 1024 - 2048) {
    why = "nfeatures too large for the LDS-resident quadtree";
    return ORBX_E_UNSUPPORTED;
  }
  return ORBX_OK;
}

// ---- resize coefficient tables, cv::resize INTER_LINEAR 8U -------------------------------
void build_coefs(const Geom& g, std::vector<int>& xofs, std::vector<short>& xab, std::vector<int>& yofs,
                 std::vector<short>& yab) {
  int nx = 0, ny = 0;
  for (int l = 0; l < g.nlevels; l++) {
    nx += g.lv[l].w;
    ny += g.lv[l].h;
  }
  xofs.assign(nx, 0);
  xab.assign(2 * nx, 0);
  yofs.assign(ny, 0);
  yab.assign(2 * ny, 0);
  for (int l = 1; l < g.nlevels; l++) {
    const LevelDev &D = g.lv[l], &S = g.lv[l - 1];
    const double scale_x = 1.0 / ((double)D.w / S.w), scale_y = 1.0 / ((double)D.h / S.h);
    for (int dx = 0; dx < D.w; dx++) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = cv_floor(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= S.w - 1) { fx = 0; sx = S.w - 1; }
      xofs[D.xcoef + dx] = sx;
      xab[2 * (D.xcoef + dx)] = sat_short((1.f - fx) * 2048.f);
      xab[2 * (D.xcoef + dx) + 1] = sat_short(fx * 2048.f);
    }
    for (int dy = 0; dy < D.h; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      int sy = cv_floor(fy);
      fy -= sy;
      yofs[D.ycoef + dy] = sy;
      yab[2 * (D.ycoef + dy)] = sat_short((1.f - fy) * 2048.f);
      yab[2 * (D.ycoef + dy) + 1] = sat_short(fy * 2048.f);
    }
  }
}

int configure(orbx_extractor* ex, int w, int h) {
  if (w == ex->curW && h == ex->curH) return ORBX_OK;
  Geom g;
  std::string why;
  int rc = build_geom(ex, w, h, g, why);
  if (rc != ORBX_OK) return fail(rc, why);
  const Geom& m = ex->gmax;
  if (g.pyrImg > m.pyrImg || g.candImg > m.candImg || g.selImg > m.selImg || g.cellImg > m.cellImg ||
      g.totalCells > m.totalCells)
    return fail(ORBX_E_CAPACITY, "image larger than the handle's max_width x max_height");
  g.outCap = m.outCap;  // keep result strides fixed for the life of the handle
  g.selImg = m.selImg;
  g.pyrImg = m.pyrImg;
  g.candImg = m.candImg;
  g.cellImg = m.cellImg;
  std::vector<int> xofs, yofs;
  std::vector<short> xab, yab;
  build_coefs(g, xofs, xab, yofs, yab);
  HIPC(hipStreamSynchronize(ex->stream));
  HIPC(hipMemcpy(ex->d_xofs.p, xofs.data(), xofs.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPC(hipMemcpy(ex->d_xab.p, xab.data(), xab.size() * sizeof(short), hipMemcpyHostToDevice));
  HIPC(hipMemcpy(ex->d_yofs.p, yofs.data(), yofs.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPC(hipMemcpy(ex->d_yab.p, yab.data(), yab.size() * sizeof(short), hipMemcpyHostToDevice));
  HIPC(prepare_kernels(g));
  ex->g = g;
  ex->curW = w;
  ex->curH = h;
  return ORBX_OK;
}

struct DetectToken {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  bool valid = false;
  const orbx_extractor* last = nullptr;
};
static DetectToken* detect_token(int device) {  // one per device, created on first use (never destroyed: process lifetime)
  static std::mutex mu;
  static DetectToken* toks[64] = {nullptr};
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!toks[device]) {
    DetectToken* t = new DetectToken();
    if (hipEventCreateWithFlags(&t->ev, hipEventDisableTiming) != hipSuccess) {
      delete t;
      return nullptr;
    }
    toks[device] = t;
  }
  return toks[device];
}

int record_pipeline(orbx_extractor* ex, int n, bool lapTrivial, bool capturing);
static void drop_graphs(orbx_extractor* ex) {
  for (auto& e : ex->graphExec) {
    if (e) (void)hipGraphExecDestroy(e);
    e = nullptr;
  }
}

int enqueue_extract(orbx_extractor* ex, const uint8_t* d_images, int n, int w, int h, ptrdiff_t row_pitch,
                    ptrdiff_t image_pitch, const int32_t* lap) {
  int rc = configure(ex, w, h);
  if (rc != ORBX_OK) return rc;
  ex->pyr.l0 = d_images;
  ex->pyr.l0Row = row_pitch;
  ex->pyr.l0Img = image_pitch;
  ex->pyr.pyr = ex->d_pyr.p;
  ex->pyr.blur = ex->d_blur.p;
  ex->lastN = n;
  ex->lastEvValid = false;
  hipStream_t s = ex->stream;
  // Keypoint x is >= 19 at every level (16-px border + the 3-px FAST ring, scaled by >= 1), so a lapping area that
  // ends below 19 -- the rectified-stereo {0, 0} in particular -- can hold no keypoint: the output order is then
  // just level-major list order, k_slots is skipped and k_describe derives its slot from the level counts.
  bool lapTrivial = true;
  if (lap)
    for (int i = 0; i < n; i++) lapTrivial = lapTrivial && lap[2 * i + 1] < 19;
  if (!lapTrivial) HIPC(hipMemcpyAsync(ex->d_lap.p, lap, (size_t)n * 2 * sizeof(int), hipMemcpyHostToDevice, s));
  // Single images through the host API are launch-bound (12 small kernels on two streams), so the pipeline can be
  // captured once per image size into a hipGraph and replayed (ORBX_GRAPH=1).  Measured on ROCm 7.2 / MI355X it is
  // SLOWER than the plain launches -- one 1280x720 eye 0.487 vs 0.310 ms, a stereo frame 0.823 vs 0.711 ms (640x480:
  // 0.263 vs 0.247 ms per eye) -- so it stays off by default.  Never used while profiling (the stage events are not
  // captured) nor for device batches (their image pointer is the caller's and changes from call to call).
  static const bool useGraph = getenv("ORBX_GRAPH") && atoi(getenv("ORBX_GRAPH")) != 0;
  if (useGraph && n == 1 && d_images == ex->d_stage.p && !ex->profiling && !ex->graphOff) {
    if (ex->graphW != w || ex->graphH != h) {
      drop_graphs(ex);
      ex->graphW = w;
      ex->graphH = h;
    }
    hipGraphExec_t& exec = ex->graphExec[lapTrivial ? 1 : 0];
    if (!exec) {
      hipGraph_t graph = nullptr;
      bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        const int rcCap = record_pipeline(ex, n, lapTrivial, true);
        const hipError_t ee = hipStreamEndCapture(s, &graph);
        ok = rcCap == ORBX_OK && ee == hipSuccess && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
      if (!ok) {  // never fatal: fall back to plain launches for the life of the handle
        (void)hipGetLastError();
        exec = nullptr;
        ex->graphOff = true;
      }
    }
    if (exec) {
      HIPC(hipGraphLaunch(exec, s));
      return ORBX_OK;
    }
  }
  return record_pipeline(ex, n, lapTrivial, false);
}

// The kernel launches of one extraction on the handle's two streams (also the body captured into the hipGraph).
int record_pipeline(orbx_extractor* ex, int n, bool lapTrivial, bool capturing) {
  const Geom& g = ex->g;
  hipStream_t s = ex->stream;
  static const bool serial = getenv("ORBX_SERIAL") != nullptr;  // measurement aid: no side stream
  hipStream_t sb = serial ? s : ex->stream2;
  // Experiment knob (default off): blur level 0 beside the resize chain.  Measured 1.063 vs 1.039 ms/step — the chain
  // slows down more than the quadtree-side blur gains.
  static const int blur0 = getenv("ORBX_BLUR0") ? atoi(getenv("ORBX_BLUR0")) : 0;
  const int nb0 = serial ? 0 : std::min(blur0, 1);  // only level 0 exists before the chain
  if (nb0 > 0) {  // level 0 is the caller's image: its blur runs beside the (latency-bound) resize chain
    HIPC(hipEventRecord(ex->evStart, s));
    HIPC(hipStreamWaitEvent(sb, ex->evStart, 0));
    StageTimer t(ex, sb, ORBX_STAGE_BLUR);
    HIPC(launch_blur(g, ex->pyr, n, 0, nb0, sb));
  }
  static const int splitEnv = getenv("ORBX_SPLIT") ? atoi(getenv("ORBX_SPLIT")) : 0;
  const int split = (splitEnv > 0 && splitEnv < g.nlevels) ? splitEnv : g.nlevels;  // levels [split, L) on the side stream
  for (int l = 1; l < split; l++) {
    StageTimer t(ex, s, ORBX_STAGE_RESIZE);
    HIPC(launch_resize(g, ex->pyr, n, l, ex->d_xofs.p, ex->d_xab.p, ex->d_yofs.p, ex->d_yab.p, s));
  }
  if (split < g.nlevels) {
    HIPC(hipEventRecord(ex->evStart, s));
    HIPC(hipStreamWaitEvent(ex->stream2, ex->evStart, 0));
    for (int l = split; l < g.nlevels; l++) {
      StageTimer t(ex, ex->stream2, ORBX_STAGE_RESIZE);
      HIPC(launch_resize(g, ex->pyr, n, l, ex->d_xofs.p, ex->d_xab.p, ex->d_yofs.p, ex->d_yab.p, ex->stream2));
    }
    {
      StageTimer t(ex, ex->stream2, ORBX_STAGE_DETECT);
      HIPC(launch_detect(g, ex->pyr, n, ex->d_cellCand.p, ex->d_cellCount.p, split, g.nlevels, ex->stream2));
    }
    HIPC(hipEventRecord(ex->evDet0, ex->stream2));
  }
  // k_detect fills every VALU of the chip by itself: two of them side by side (two handles in flight) only stretch
  // each other.  A per-device token orders the k_detect launches of all handles one after the other, while each still
  // overlaps the other handles' quadtree / describe / stereo / resize work.  (ORBX_DETECT_TOKEN=0 disables it.)
  static const bool useToken = !(getenv("ORBX_DETECT_TOKEN") && atoi(getenv("ORBX_DETECT_TOKEN")) == 0);
  DetectToken* tok = (useToken && !capturing) ? detect_token(ex->device) : nullptr;  // (a graph cannot wait on it)
  if (tok) {
    std::lock_guard<std::mutex> lk(tok->mu);
    if (tok->valid && tok->last != ex) HIPC(hipStreamWaitEvent(s, tok->ev, 0));
    ex->lastEvValid = false;
    {
      StageTimer t(ex, s, ORBX_STAGE_DETECT);
      HIPC(launch_detect(g, ex->pyr, n, ex->d_cellCand.p, ex->d_cellCount.p, 0, split, s));
    }
    HIPC(hipEventRecord(tok->ev, s));
    tok->valid = true;
    tok->last = ex;
  } else {
    StageTimer t(ex, s, ORBX_STAGE_DETECT);
    HIPC(launch_detect(g, ex->pyr, n, ex->d_cellCand.p, ex->d_cellCount.p, 0, split, s));
  }
  if (split < g.nlevels) {
    HIPC(hipStreamWaitEvent(s, ex->evDet0, 0));
    ex->lastEvValid = false;
  }
  // The blurred copies only depend on the pyramid.  They run on the side stream, released once k_detect (which
  // fills the chip by itself) is done, so that the streaming blur shares the GPU with the latency-bound quadtree.
  // (Also tried: FAST on level 0 beside the resize chain -- slower, 1.48 vs 1.33 ms/step: both just time-slice.)
  HIPC(hipEventRecord(ex->evPyr, s));
  HIPC(hipStreamWaitEvent(sb, ex->evPyr, 0));
  {
    StageTimer t(ex, sb, ORBX_STAGE_BLUR);
    HIPC(launch_blur(g, ex->pyr, n, nb0, g.nlevels, sb));
  }
  HIPC(hipEventRecord(ex->evBlur, sb));
  {
    StageTimer t(ex, s, ORBX_STAGE_OCTREE);
    HIPC(launch_octree(g, n, ex->d_cellCand.p, ex->d_cellCount.p, ex->d_cellPrefix.p, ex->d_cand.p,
                       ex->d_candCount.p, ex->d_knode.p, ex->d_sel.p, ex->d_selCount.p, s));
  }
  if (!lapTrivial) {
    StageTimer t(ex, s, ORBX_STAGE_SLOTS);
    HIPC(launch_slots(g, n, ex->d_sel.p, ex->d_selCount.p, ex->d_lap.p, ex->d_slot.p, ex->d_nOut.p, ex->d_mono.p, s));
  }
  HIPC(hipStreamWaitEvent(s, ex->evBlur, 0));
  ex->lastEvValid = false;  // fresh start event: do not bill the wait for the side stream to k_describe
  {
    StageTimer t(ex, s, ORBX_STAGE_DESCRIBE);
    HIPC(launch_describe(g, ex->pyr, n, ex->d_sel.p, ex->d_selCount.p, lapTrivial ? nullptr : ex->d_slot.p, ex->d_kps.p,
                         ex->d_desc.p, ex->d_nOut.p, ex->d_mono.p, s));
  }
  return ORBX_OK;
}

int set_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(ORBX_E_NODEVICE, "no HIP device available");
  if (device < 0 || device >= n) return fail(ORBX_E_BADARG, "device index out of range");
  HIPC(hipSetDevice(device));
  return ORBX_OK;
}

}  // namespace

extern "C" {

const char* orbx_last_error(void) { return g_err.c_str(); }
int orbx_abi_version(void) { return 1; }
int orbx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int orbx_extractor_create(const orbx_params* p, int max_width, int max_height, int max_batch, int device,
                          orbx_extractor** out) {
  if (!p || !out) return fail(ORBX_E_BADARG, "null argument");
  *out = nullptr;
  if (p->nlevels < 1 || p->nlevels > ORBX_MAX_LEVELS || p->nfeatures < 1 || !(p->scale_factor > 1.0f) ||
      p->ini_th_fast < p->min_th_fast || p->min_th_fast < 1 || p->ini_th_fast > 254 || max_batch < 1 ||
      max_width < 1 || max_height < 1)
    return fail(ORBX_E_BADARG, "invalid extractor parameters");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  orbx_extractor* ex = new orbx_extractor();
  ex->prm = *p;
  ex->device = device;
  ex->maxW = max_width;
  ex->maxH = max_height;
  ex->maxB = max_batch;
  build_tables(ex);
  std::string why;
  rc = build_geom(ex, max_width, max_height, ex->gmax, why);
  if (rc != ORBX_OK) {
    delete ex;
    return fail(rc, why);
  }
  const Geom& m = ex->gmax;
  const size_t B = (size_t)max_batch;
  ex->stagePitch = align_up(max_width, 64);
  int nx = 0, ny = 0;
  for (int l = 0; l < m.nlevels; l++) {
    nx += m.lv[l].w;
    ny += m.lv[l].h;
  }
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  {
    // ORBX_PRIO (experiment knob): 0 = both default, 1 = side stream (blur) high, 2 = main stream (quadtree) high
    static const int prio = getenv("ORBX_PRIO") ? atoi(getenv("ORBX_PRIO")) : 1;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    ok(hipStreamCreateWithPriority(&ex->stream, hipStreamNonBlocking, prio == 2 ? hi : 0));
    ok(hipStreamCreateWithPriority(&ex->stream2, hipStreamNonBlocking, prio == 1 ? hi : (prio == 2 ? lo : 0)));
  }
  ok(hipEventCreateWithFlags(&ex->done, hipEventDisableTiming));
  ok(hipEventCreateWithFlags(&ex->evPyr, hipEventDisableTiming));
  ok(hipEventCreateWithFlags(&ex->evBlur, hipEventDisableTiming));
  ok(hipEventCreateWithFlags(&ex->evStart, hipEventDisableTiming));
  ok(hipEventCreateWithFlags(&ex->evDet0, hipEventDisableTiming));
  ok(ex->d_pyr.alloc(B * m.pyrImg + 256));
  ok(ex->d_blur.alloc(B * m.pyrImg + 256));
  ok(ex->d_stage.alloc(B * (size_t)ex->stagePitch * max_height + 256));
  ok(ex->d_cand.alloc(B * m.candImg));
  ok(ex->d_cellCand.alloc(B * m.cellImg));
  ok(ex->d_cellCount.alloc(B * m.totalCells));
  ok(ex->d_cellPrefix.alloc(B * m.totalCells));
  ok(ex->d_knode.alloc(B * m.candImg));
  ok(ex->d_candCount.alloc(B * m.nlevels));
  ok(ex->d_sel.alloc(B * m.selImg));
  ok(ex->d_selCount.alloc(B * m.nlevels));
  ok(ex->d_slot.alloc(B * m.selImg));
  ok(ex->d_kps.alloc(B * m.outCap));
  ok(ex->d_desc.alloc(B * m.outCap * 32));
  ok(ex->d_nOut.alloc(B));
  ok(ex->d_mono.alloc(B));
  ok(ex->d_lap.alloc(B * 2));
  ok(ex->d_xofs.alloc(nx + 64));
  ok(ex->d_xab.alloc(2 * nx + 64));
  ok(ex->d_yofs.alloc(ny + 64));
  ok(ex->d_yab.alloc(2 * ny + 64));
  ok(hipHostMalloc(reinterpret_cast<void**>(&ex->hostResults), host_results_bytes(m.outCap), hipHostMallocDefault));
  if (e != hipSuccess) {
    std::string msg = std::string("allocation failed: ") + hipGetErrorString(e);
    orbx_extractor_destroy(ex);
    return fail(ORBX_E_HIP, msg);
  }
  *out = ex;
  return ORBX_OK;
}

void orbx_extractor_destroy(orbx_extractor* ex) {
  if (!ex) return;
  (void)hipSetDevice(ex->device);
  if (ex->stream) (void)hipStreamSynchronize(ex->stream);
  for (auto& e : ex->graphExec) {
    if (e) (void)hipGraphExecDestroy(e);
    e = nullptr;
  }
  if (ex->hostResults) (void)hipHostFree(ex->hostResults);
  ex->hostResults = nullptr;
  ex->d_pyr.free(); ex->d_blur.free(); ex->d_stage.free(); ex->d_desc.free(); ex->d_cand.free(); ex->d_cellCand.free(); ex->d_cellCount.free(); ex->d_cellPrefix.free();
  ex->d_sel.free(); ex->d_knode.free(); ex->d_candCount.free(); ex->d_selCount.free(); ex->d_slot.free();
  ex->d_nOut.free(); ex->d_mono.free(); ex->d_lap.free(); ex->d_fl2r.free(); ex->d_fr2l.free(); ex->d_fcnt.free(); ex->d_bowWord.free(); ex->d_bowNode.free(); ex->d_bowStart.free();
  ex->d_bowCounts.free(); ex->d_bowWeight.free(); ex->d_bowValues.free(); ex->d_bowWords.free(); ex->d_bowNodes.free(); ex->d_bowFeats.free(); ex->d_fdepth.free(); ex->d_fp3d.free(); ex->d_xofs.free(); ex->d_yofs.free();
  ex->d_xab.free(); ex->d_yab.free(); ex->d_kps.free(); ex->d_uR.free(); ex->d_depth.free(); ex->d_sad.free(); ex->d_rowStart.free(); ex->d_rowItems.free();
  for (hipEvent_t e : ex->evPool) (void)hipEventDestroy(e);
  if (ex->done) (void)hipEventDestroy(ex->done);
  if (ex->evPyr) (void)hipEventDestroy(ex->evPyr);
  if (ex->evBlur) (void)hipEventDestroy(ex->evBlur);
  if (ex->evStart) (void)hipEventDestroy(ex->evStart);
  if (ex->evDet0) (void)hipEventDestroy(ex->evDet0);
  if (ex->stream2) (void)hipStreamDestroy(ex->stream2);
  if (ex->stream) (void)hipStreamDestroy(ex->stream);
  delete ex;
}

int orbx_get_tables(const orbx_extractor* ex, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                    int32_t* nfeatures_per_level, int32_t* umax16) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  const int L = ex->prm.nlevels;
  if (scale) std::memcpy(scale, ex->scale.data(), L * sizeof(float));
  if (inv_scale) std::memcpy(inv_scale, ex->inv.data(), L * sizeof(float));
  if (sigma2) std::memcpy(sigma2, ex->sig2.data(), L * sizeof(float));
  if (inv_sigma2) std::memcpy(inv_sigma2, ex->invsig2.data(), L * sizeof(float));
  if (nfeatures_per_level) std::memcpy(nfeatures_per_level, ex->nfeat.data(), L * sizeof(int));
  if (umax16) std::memcpy(umax16, ex->umax, sizeof(ex->umax));
  return ORBX_OK;
}

int orbx_extract_batch_device(orbx_extractor* ex, const uint8_t* d_images, int n_images, int w, int h,
                              ptrdiff_t row_pitch, ptrdiff_t image_pitch, const int32_t* lap) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (!d_images || n_images <= 0 || w <= 0 || h <= 0) return fail(ORBX_E_EMPTY, "empty image");
  if (n_images > ex->maxB) return fail(ORBX_E_CAPACITY, "batch larger than max_batch");
  if (row_pitch < w || ((uintptr_t)d_images & 3) || (row_pitch & 3) || (image_pitch & 3))
    return fail(ORBX_E_BADARG, "images must be 4-byte aligned with 4-byte aligned pitches >= width");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  return enqueue_extract(ex, d_images, n_images, w, h, row_pitch, image_pitch, lap);
}

int orbx_sync(orbx_extractor* ex) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  return ORBX_OK;
}

int orbx_batch_results_device(const orbx_extractor* ex, const orbx_keypoint** d_kps, const uint8_t** d_desc,
                              const int32_t** d_counts, const int32_t** d_mono, int* cap) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (d_kps) *d_kps = ex->d_kps.p;
  if (d_desc) *d_desc = ex->d_desc.p;
  if (d_counts) *d_counts = ex->d_nOut.p;
  if (d_mono) *d_mono = ex->d_mono.p;
  if (cap) *cap = ex->gmax.outCap;
  return ORBX_OK;
}

int orbx_batch_download(orbx_extractor* ex, int image, orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
  if (!ex || !n_out) return fail(ORBX_E_BADARG, "null argument");
  if (image < 0 || image >= ex->lastN) return fail(ORBX_E_BADARG, "image index out of range");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  int n = 0, mono = 0;
  HIPC(hipMemcpy(&n, ex->d_nOut.p + image, sizeof(int), hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(&mono, ex->d_mono.p + image, sizeof(int), hipMemcpyDeviceToHost));
  *n_out = n;
  if (n > cap) return fail(ORBX_E_CAPACITY, "keypoint buffer too small");
  const size_t oc = (size_t)ex->gmax.outCap;
  if (n > 0 && kps) HIPC(hipMemcpy(kps, ex->d_kps.p + image * oc, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost));
  if (n > 0 && desc) HIPC(hipMemcpy(desc, ex->d_desc.p + image * oc * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
  return mono;
}

int orbx_extract(orbx_extractor* ex, const uint8_t* img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                 orbx_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (n_out) *n_out = 0;
  if (!img || w <= 0 || h <= 0) return fail(ORBX_E_EMPTY, "empty image");  // :1021
  if (w > ex->maxW || h > ex->maxH) return fail(ORBX_E_CAPACITY, "image larger than the handle's maximum");
  if (stride < w) return fail(ORBX_E_BADARG, "stride < width");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const int pitch = align_up(w, 64);
  HIPC(hipMemcpy2DAsync(ex->d_stage.p, pitch, img, stride, w, h, hipMemcpyHostToDevice, ex->stream));
  const int32_t lap[2] = {lap0, lap1};
  rc = enqueue_extract(ex, ex->d_stage.p, 1, w, h, pitch, (ptrdiff_t)pitch * h, lap);
  if (rc != ORBX_OK) return rc;
  if (!n_out) return fail(ORBX_E_BADARG, "null argument");
  const size_t oc = (size_t)ex->gmax.outCap;  // results through pinned memory: async copies, one synchronisation
  uint8_t* H = ex->hostResults;
  hipStream_t st = ex->stream;
  HIPC(hipMemcpyAsync(H, ex->d_nOut.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPC(hipMemcpyAsync(H + 8, ex->d_mono.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPC(hipMemcpyAsync(H + hr_kps(oc), ex->d_kps.p, oc * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, st));
  HIPC(hipMemcpyAsync(H + hr_desc(oc), ex->d_desc.p, oc * 32, hipMemcpyDeviceToHost, st));
  HIPC(hipStreamSynchronize(st));
  const int n = *reinterpret_cast<const int*>(H), mono = *reinterpret_cast<const int*>(H + 8);
  *n_out = n;
  if (n > cap) return fail(ORBX_E_CAPACITY, "keypoint buffer too small");
  if (n > 0 && kps) std::memcpy(kps, H + hr_kps(oc), (size_t)n * sizeof(orbx_keypoint));
  if (n > 0 && desc) std::memcpy(desc, H + hr_desc(oc), (size_t)n * 32);
  return mono;
}

int orbx_extract_stereo(orbx_extractor* ex, const uint8_t* img_left, const uint8_t* img_right, int w, int h,
                        ptrdiff_t stride_left, ptrdiff_t stride_right, const int32_t lap_left[2],
                        const int32_t lap_right[2], orbx_keypoint* kps_left, uint8_t* desc_left, int cap_left,
                        int* n_left, int* mono_left, orbx_keypoint* kps_right, uint8_t* desc_right, int cap_right,
                        int* n_right, int* mono_right, float bf, float b, float* uright, float* depth) {
  if (!ex || !n_left || !n_right || !mono_left || !mono_right) return fail(ORBX_E_BADARG, "null argument");
  *n_left = *n_right = 0;
  *mono_left = *mono_right = 0;
  if (!img_left || !img_right || w <= 0 || h <= 0) return fail(ORBX_E_EMPTY, "empty image");
  if (ex->maxB < 2) return fail(ORBX_E_CAPACITY, "orbx_extract_stereo needs a handle created with max_batch >= 2");
  if (w > ex->maxW || h > ex->maxH) return fail(ORBX_E_CAPACITY, "image larger than the handle's maximum");
  if (stride_left < w || stride_right < w) return fail(ORBX_E_BADARG, "stride < width");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const int pitch = align_up(w, 64);
  const size_t imgBytes = (size_t)pitch * h;
  HIPC(hipMemcpy2DAsync(ex->d_stage.p, pitch, img_left, stride_left, w, h, hipMemcpyHostToDevice, ex->stream));
  HIPC(hipMemcpy2DAsync(ex->d_stage.p + imgBytes, pitch, img_right, stride_right, w, h, hipMemcpyHostToDevice, ex->stream));
  const int32_t lap[4] = {lap_left ? lap_left[0] : 0, lap_left ? lap_left[1] : 0, lap_right ? lap_right[0] : 0,
                          lap_right ? lap_right[1] : 0};
  rc = enqueue_extract(ex, ex->d_stage.p, 2, w, h, pitch, (ptrdiff_t)imgBytes, lap);
  if (rc != ORBX_OK) return rc;
  const bool stereo = bf > 0.f && uright && depth;
  if (stereo) {
    rc = orbx_stereo_match_batch(ex, 0, ex, 1, 1, bf, b);
    if (rc != ORBX_OK) return rc;
  }
  // all results travel with asynchronous copies into pinned memory behind the kernels: one synchronisation in total
  const size_t oc = (size_t)ex->gmax.outCap;
  uint8_t* H = ex->hostResults;
  hipStream_t st = ex->stream;
  HIPC(hipMemcpyAsync(H, ex->d_nOut.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPC(hipMemcpyAsync(H + 8, ex->d_mono.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPC(hipMemcpyAsync(H + hr_kps(oc), ex->d_kps.p, 2 * oc * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, st));
  HIPC(hipMemcpyAsync(H + hr_desc(oc), ex->d_desc.p, 2 * oc * 32, hipMemcpyDeviceToHost, st));
  if (stereo) {
    HIPC(hipMemcpyAsync(H + hr_ur(oc), ex->d_uR.p, oc * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(H + hr_depth(oc), ex->d_depth.p, oc * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  HIPC(hipStreamSynchronize(st));
  const int* cnt = reinterpret_cast<const int*>(H);
  const int* mono = reinterpret_cast<const int*>(H + 8);
  *n_left = cnt[0]; *n_right = cnt[1];
  *mono_left = mono[0]; *mono_right = mono[1];
  if (cnt[0] > cap_left || cnt[1] > cap_right) return fail(ORBX_E_CAPACITY, "keypoint buffer too small");
  if (cnt[0] > 0 && kps_left) std::memcpy(kps_left, H + hr_kps(oc), (size_t)cnt[0] * sizeof(orbx_keypoint));
  if (cnt[0] > 0 && desc_left) std::memcpy(desc_left, H + hr_desc(oc), (size_t)cnt[0] * 32);
  if (cnt[1] > 0 && kps_right) std::memcpy(kps_right, H + hr_kps(oc) + oc * sizeof(orbx_keypoint), (size_t)cnt[1] * sizeof(orbx_keypoint));
  if (cnt[1] > 0 && desc_right) std::memcpy(desc_right, H + hr_desc(oc) + oc * 32, (size_t)cnt[1] * 32);
  if (stereo && cnt[0] > 0) {
    std::memcpy(uright, H + hr_ur(oc), (size_t)cnt[0] * sizeof(float));
    std::memcpy(depth, H + hr_depth(oc), (size_t)cnt[0] * sizeof(float));
  }
  return ORBX_OK;
}

int orbx_pyramid_level(orbx_extractor* ex, int image, int level, int blurred, uint8_t* dst, ptrdiff_t dst_stride,
                       int* w, int* h) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (ex->curW == 0 || image < 0 || image >= ex->lastN || level < 0 || level >= ex->g.nlevels)
    return fail(ORBX_E_BADARG, "no such pyramid level");
  const LevelDev& L = ex->g.lv[level];
  if (w) *w = L.w;
  if (h) *h = L.h;
  if (!dst) return ORBX_OK;
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  const uint8_t* src;
  size_t pitch;
  if (blurred) {
    src = ex->pyr.blur + (long long)image * ex->g.pyrImg + L.off;
    pitch = L.pitch;
  } else {
    int p;
    src = level_ptr(ex->g, ex->pyr, image, level, p);
    pitch = p;
  }
  HIPC(hipMemcpy2D(dst, dst_stride, src, pitch, L.w, L.h, hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_debug_candidates(orbx_extractor* ex, int image, int level, int32_t* xys, int cap) {
  if (!ex || image < 0 || image >= ex->lastN || level < 0 || level >= ex->g.nlevels)
    return fail(ORBX_E_BADARG, "bad argument");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  int n = 0;
  HIPC(hipMemcpy(&n, ex->d_candCount.p + image * ex->g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
  const LevelDev& L = ex->g.lv[level];
  if (n > L.candCap) n = L.candCap;
  std::vector<uint32_t> v(n);
  if (n)
    HIPC(hipMemcpy(v.data(), ex->d_cand.p + (long long)image * ex->g.candImg + L.candOff, (size_t)n * 4,
                   hipMemcpyDeviceToHost));
  for (int i = 0; i < n && i < cap; i++) {
    xys[3 * i] = key_x(v[i]);
    xys[3 * i + 1] = key_y(v[i]);
    xys[3 * i + 2] = key_r(v[i]);
  }
  return n;
}

int orbx_hamming256(const void* a, const void* b) {
  const uint32_t* x = (const uint32_t*)a;
  const uint32_t* y = (const uint32_t*)b;
  int d = 0;
  for (int i = 0; i < 8; i++) d += __builtin_popcount(x[i] ^ y[i]);
  return d;
}

int orbx_stereo_match_batch(orbx_extractor* left, int first_left, orbx_extractor* right, int first_right,
                            int n_pairs, float bf, float b) {
  if (!left || !right) return fail(ORBX_E_BADARG, "null handle");
  if (n_pairs <= 0 || first_left < 0 || first_right < 0 || first_left + n_pairs > left->lastN ||
      first_right + n_pairs > right->lastN)
    return fail(ORBX_E_BADARG, "pair range outside the last extraction");
  if (left->device != right->device || left->curW != right->curW || left->curH != right->curH ||
      std::memcmp(&left->prm, &right->prm, sizeof(orbx_params)) != 0)
    return fail(ORBX_E_BADARG, "left and right extractors must share device, image size and parameters");
  if (!(b > 0.f)) return fail(ORBX_E_BADARG, "baseline must be positive");
  HIPC(hipSetDevice(left->device));
  const size_t capL = (size_t)left->gmax.outCap;
  if (left->stereoPairs < n_pairs) {
    HIPC(hipStreamSynchronize(left->stream));
    HIPC(left->d_uR.alloc((size_t)n_pairs * capL));
    HIPC(left->d_depth.alloc((size_t)n_pairs * capL));
    HIPC(left->d_sad.alloc((size_t)n_pairs * capL));
    HIPC(left->d_rowStart.alloc((size_t)n_pairs * (left->maxH + 2)));
    HIPC(left->d_rowItems.alloc((size_t)n_pairs * right->gmax.outCap));
    left->stereoPairs = n_pairs;
  }
  if (right != left) {  // order left's stream after right's extraction
    HIPC(hipEventRecord(right->done, right->stream));
    HIPC(hipStreamWaitEvent(left->stream, right->done, 0));
  }
  StereoArgs a;
  a.kL = left->d_kps.p;
  a.kR = right->d_kps.p;
  a.dL = left->d_desc.p;
  a.dR = right->d_desc.p;
  a.nL = left->d_nOut.p;
  a.nR = right->d_nOut.p;
  a.capL = (int)capL;
  a.capR = right->gmax.outCap;
  a.firstL = first_left;
  a.firstR = first_right;
  a.bf = bf;
  a.b = b;
  a.uRight = left->d_uR.p;
  a.depth = left->d_depth.p;
  a.sad = left->d_sad.p;
  a.rowStart = left->d_rowStart.p;
  a.rowItems = left->d_rowItems.p;
  a.imgH = left->curH;
  a.band = (int)std::ceil(2.0f * left->scale.back()) + 2;
  {
    StageTimer t(left, left->stream, ORBX_STAGE_STEREO_MATCH);
    HIPC(launch_stereo_rows(a, n_pairs, left->stream));
  }
  {
    StageTimer t(left, left->stream, ORBX_STAGE_STEREO_MATCH);
    HIPC(launch_stereo_match(left->g, left->pyr, right->pyr, a, n_pairs, left->stream));
  }
  {
    StageTimer t(left, left->stream, ORBX_STAGE_STEREO_FILTER);
    HIPC(launch_stereo_filter(a, n_pairs, left->stream));
  }
  return ORBX_OK;
}

int orbx_stereo_results_device(const orbx_extractor* left, const float** d_uright, const float** d_depth) {
  if (!left) return fail(ORBX_E_BADARG, "null handle");
  if (d_uright) *d_uright = left->d_uR.p;
  if (d_depth) *d_depth = left->d_depth.p;
  return ORBX_OK;
}

int orbx_stereo_download(orbx_extractor* left, int pair, float* uright, float* depth, int cap) {
  if (!left || pair < 0 || pair >= left->stereoPairs) return fail(ORBX_E_BADARG, "bad pair index");
  HIPC(hipSetDevice(left->device));
  HIPC(hipStreamSynchronize(left->stream));
  const size_t capL = (size_t)left->gmax.outCap;
  const size_t n = std::min((size_t)cap, capL);
  if (uright) HIPC(hipMemcpy(uright, left->d_uR.p + pair * capL, n * sizeof(float), hipMemcpyDeviceToHost));
  if (depth) HIPC(hipMemcpy(depth, left->d_depth.p + pair * capL, n * sizeof(float), hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_bf_knn2(int device, const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, int32_t* idx2,
                 int32_t* dist2, uint8_t* ratio_ok) {
  if (nQ < 0 || nT < 0 || (nQ && (!descQ || !idx2 || !dist2 || !ratio_ok)) || (nT && !descT))
    return fail(ORBX_E_BADARG, "bad argument");
  if (nQ == 0) return ORBX_OK;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  Pack pk;
  const size_t Q = (size_t)nQ;
  const size_t oQ = pk.add(descQ, Q * 32), oT = pk.add(descT, (size_t)std::max(nT, 1) * 32);
  const size_t oOut = pk.add(nullptr, Q * 2 * 4 * 2 + Q);  // idx2 | dist2 | ratio_ok: one copy back
  hipError_t e = pk.commit();
  int* i2 = pk.ptr<int>(oOut);
  int* d2 = i2 + Q * 2;
  uint8_t* ok = reinterpret_cast<uint8_t*>(d2 + Q * 2);
  if (e == hipSuccess) e = launch_bf_knn2(pk.ptr<uint8_t>(oQ), nQ, pk.ptr<uint8_t>(oT), nT, i2, d2, ok, nullptr);
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOut, Q * 17, &e);
    if (e == hipSuccess) {
      std::memcpy(idx2, h, Q * 8);
      std::memcpy(dist2, h + Q * 8, Q * 8);
      std::memcpy(ratio_ok, h + Q * 16, Q);
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

int orbx_fisheye_stereo_match_batch(orbx_extractor* left, int first_left, orbx_extractor* right, int first_right,
                                    int n_pairs, const orbx_kb8_rig* rig) {
  if (!left || !right || !rig) return fail(ORBX_E_BADARG, "null argument");
  if (n_pairs <= 0 || first_left < 0 || first_right < 0 || first_left + n_pairs > left->lastN ||
      first_right + n_pairs > right->lastN)
    return fail(ORBX_E_BADARG, "pair range outside the last extraction");
  if (left->device != right->device || std::memcmp(&left->prm, &right->prm, sizeof(orbx_params)) != 0)
    return fail(ORBX_E_BADARG, "left and right extractors must share device and parameters");
  HIPC(hipSetDevice(left->device));
  const size_t capL = (size_t)left->gmax.outCap, capR = (size_t)right->gmax.outCap;
  if (left->fisheyePairs < n_pairs || left->fisheyeCapR < (int)capR) {
    HIPC(hipStreamSynchronize(left->stream));
    HIPC(left->d_fl2r.alloc((size_t)n_pairs * capL));
    HIPC(left->d_fr2l.alloc((size_t)n_pairs * capR));
    HIPC(left->d_fdepth.alloc((size_t)n_pairs * capL));
    HIPC(left->d_fp3d.alloc((size_t)n_pairs * capL * 3));
    HIPC(left->d_fcnt.alloc((size_t)n_pairs * 2));
    left->fisheyePairs = n_pairs;
    left->fisheyeCapR = (int)capR;
  }
  hipStream_t s = left->stream;
  if (right != left) {  // order left's stream after right's extraction
    HIPC(hipEventRecord(right->done, right->stream));
    HIPC(hipStreamWaitEvent(s, right->done, 0));
  }
  FisheyeBatchArgs a;
  a.kL = left->d_kps.p; a.kR = right->d_kps.p; a.dL = left->d_desc.p; a.dR = right->d_desc.p;
  a.nL = left->d_nOut.p; a.nR = right->d_nOut.p; a.monoL = left->d_mono.p; a.monoR = right->d_mono.p;
  a.capL = (int)capL; a.capR = (int)capR; a.firstL = first_left; a.firstR = first_right;
  a.rig = *rig;
  a.nLevels = left->prm.nlevels;
  for (int l = 0; l < ORBX_MAX_LEVELS; l++) a.sigma2[l] = l < a.nLevels ? left->sig2[l] : 0.f;  // Frame::mvLevelSigma2
  a.leftToRight = left->d_fl2r.p; a.rightToLeft = left->d_fr2l.p; a.depth = left->d_fdepth.p; a.p3D = left->d_fp3d.p;
  a.counters = left->d_fcnt.p;
  {
    StageTimer t(left, s, ORBX_STAGE_STEREO_MATCH);
    HIPC(launch_fisheye_batch(a, n_pairs, s));
  }
  return ORBX_OK;
}

int orbx_fisheye_results_device(const orbx_extractor* left, const int32_t** d_left_to_right, const int32_t** d_right_to_left,
                                const float** d_depth, const float** d_points3d, const int32_t** d_counts) {
  if (!left || left->fisheyePairs == 0) return fail(ORBX_E_BADARG, "no fisheye association has been run on this handle");
  if (d_left_to_right) *d_left_to_right = left->d_fl2r.p;
  if (d_right_to_left) *d_right_to_left = left->d_fr2l.p;
  if (d_depth) *d_depth = left->d_fdepth.p;
  if (d_points3d) *d_points3d = left->d_fp3d.p;
  if (d_counts) *d_counts = left->d_fcnt.p;
  return ORBX_OK;
}

int orbx_fisheye_download(orbx_extractor* left, int pair, int32_t* left_to_right, int32_t* right_to_left, float* depth,
                          float* points3d, int cap_left, int cap_right, int32_t* n_desc_matches) {
  if (!left || pair < 0 || pair >= left->fisheyePairs) return fail(ORBX_E_BADARG, "bad pair index");
  HIPC(hipSetDevice(left->device));
  HIPC(hipStreamSynchronize(left->stream));
  const size_t capL = (size_t)left->gmax.outCap, capR = (size_t)left->fisheyeCapR;
  const size_t nl = std::min(capL, (size_t)std::max(cap_left, 0)), nr = std::min(capR, (size_t)std::max(cap_right, 0));
  if (left_to_right) HIPC(hipMemcpy(left_to_right, left->d_fl2r.p + pair * capL, nl * sizeof(int), hipMemcpyDeviceToHost));
  if (right_to_left) HIPC(hipMemcpy(right_to_left, left->d_fr2l.p + pair * capR, nr * sizeof(int), hipMemcpyDeviceToHost));
  if (depth) HIPC(hipMemcpy(depth, left->d_fdepth.p + pair * capL, nl * sizeof(float), hipMemcpyDeviceToHost));
  if (points3d) HIPC(hipMemcpy(points3d, left->d_fp3d.p + pair * capL * 3, nl * 3 * sizeof(float), hipMemcpyDeviceToHost));
  int cnt[2] = {0, 0};
  HIPC(hipMemcpy(cnt, left->d_fcnt.p + 2 * pair, sizeof(cnt), hipMemcpyDeviceToHost));
  if (n_desc_matches) *n_desc_matches = cnt[1];
  return cnt[0];
}

int orbx_fisheye_stereo_match(int device, const orbx_keypoint* kps_left, const uint8_t* desc_left, int n_left,
                              int mono_left, const orbx_keypoint* kps_right, const uint8_t* desc_right, int n_right,
                              int mono_right, const orbx_kb8_rig* rig, const float* level_sigma2, int n_levels,
                              int32_t* left_to_right, int32_t* right_to_left, float* depth, float* points3d,
                              int32_t* n_desc_matches) {
  if (n_left < 0 || n_right < 0 || mono_left < 0 || mono_left > n_left || mono_right < 0 || mono_right > n_right ||
      !rig || !level_sigma2 || n_levels <= 0 || n_levels > ORBX_MAX_LEVELS ||
      (n_left && (!kps_left || !desc_left || !left_to_right || !depth || !points3d)) ||
      (n_right && (!kps_right || !desc_right || !right_to_left)))
    return fail(ORBX_E_BADARG, "bad argument");
  for (int i = 0; i < n_left; i++) {
    left_to_right[i] = -1;
    depth[i] = -1.0f;
    points3d[3 * i] = points3d[3 * i + 1] = points3d[3 * i + 2] = 0.0f;
  }
  for (int i = 0; i < n_right; i++) right_to_left[i] = -1;
  if (n_desc_matches) *n_desc_matches = 0;
  const int nQ = n_left - mono_left, nT = n_right - mono_right;
  // knnMatch(k = 2) yields pairs only when the train set has two rows (`(*it).size() >= 2`, src/Frame.cc:1302)
  if (nQ == 0 || nT < 2) {
    int rc0 = set_device(device);  // still a device routine: no GPU is an error, never a silent host path
    return rc0 != ORBX_OK ? rc0 : 0;
  }
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  // one packed upload (the -1 / 0 fills of the outputs travel with it); the outputs are contiguous: one copy back
  std::vector<float> zeros((size_t)n_left * 3 + 2, 0.0f);  // points3d fill + the two counters
  Pack pk;
  const size_t NL = (size_t)n_left, NR = (size_t)n_right;
  const size_t oKl = pk.add(kps_left, NL * sizeof(orbx_keypoint)), oKr = pk.add(kps_right, NR * sizeof(orbx_keypoint));
  const size_t oDq = pk.add(desc_left + (size_t)mono_left * 32, (size_t)nQ * 32);
  const size_t oDt = pk.add(desc_right + (size_t)mono_right * 32, (size_t)nT * 32);
  const size_t oSg = pk.add(level_sigma2, (size_t)n_levels * sizeof(float));
  const size_t oL2r = pk.add(left_to_right, NL * 4), oR2l = pk.add(right_to_left, NR * 4), oDep = pk.add(depth, NL * 4);
  const size_t oPts = pk.add(zeros.data(), NL * 12), oCnt = pk.add(zeros.data(), 8);
  const size_t outBytes = oCnt + 8 - oL2r;
  const size_t oOk = pk.add(nullptr, nQ), oI2 = pk.add(nullptr, (size_t)nQ * 8), oD2 = pk.add(nullptr, (size_t)nQ * 8);
  hipError_t e = pk.commit();
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  if (e == hipSuccess)
    chk(launch_bf_knn2(pk.ptr<uint8_t>(oDq), nQ, pk.ptr<uint8_t>(oDt), nT, pk.ptr<int>(oI2), pk.ptr<int>(oD2), pk.ptr<uint8_t>(oOk), nullptr));
  if (e == hipSuccess) {
    FisheyeArgs a;
    a.kL = pk.ptr<orbx_keypoint>(oKl); a.kR = pk.ptr<orbx_keypoint>(oKr); a.nL = n_left; a.nR = n_right; a.monoL = mono_left;
    a.monoR = mono_right;
    a.idx2 = pk.ptr<int>(oI2); a.ratioOk = pk.ptr<uint8_t>(oOk); a.rig = *rig; a.sigma2 = pk.ptr<float>(oSg); a.nLevels = n_levels;
    a.leftToRight = pk.ptr<int>(oL2r); a.rightToLeft = pk.ptr<int>(oR2l); a.depth = pk.ptr<float>(oDep);
    a.p3D = pk.ptr<float>(oPts); a.counters = pk.ptr<int>(oCnt);
    chk(launch_fisheye_triangulate(a, nullptr));
  }
  int counts[2] = {0, 0};
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oL2r, outBytes, &e);
    if (e == hipSuccess) {
      std::memcpy(left_to_right, h, NL * 4);
      std::memcpy(right_to_left, h + (oR2l - oL2r), NR * 4);
      std::memcpy(depth, h + (oDep - oL2r), NL * 4);
      std::memcpy(points3d, h + (oPts - oL2r), NL * 12);
      std::memcpy(counts, h + (oCnt - oL2r), sizeof(counts));
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  if (n_desc_matches) *n_desc_matches = counts[1];
  return counts[0];
}

int orbx_cvt_gray(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, int rgb_order,
                  uint8_t* dst, ptrdiff_t dst_stride) {
  if (!src || !dst || w <= 0 || h <= 0 || (channels != 3 && channels != 4) || src_stride < (ptrdiff_t)w * channels ||
      dst_stride < w)
    return fail(ORBX_E_BADARG, "bad argument");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<uint8_t> ds, dd;
  const size_t sp = (size_t)w * channels, dp = (size_t)w;
  hipError_t e = ds.alloc(sp * h);
  if (e == hipSuccess) e = dd.alloc(dp * h);
  if (e == hipSuccess) e = hipMemcpy2D(ds.p, sp, src, (size_t)src_stride, sp, h, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_cvt_gray(ds.p, w, h, (long long)sp, 0, channels, rgb_order ? 1 : 0, dd.p, (long long)dp, 0, 1, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy2D(dst, (size_t)dst_stride, dd.p, dp, dp, h, hipMemcpyDeviceToHost);
  ds.free(); dd.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

// Coefficient tables exactly as cv::resize builds them (the same arithmetic as the pyramid's build_coefs).
static void build_resize_tables(int w, int h, int dst_w, int dst_h, std::vector<int>& xofs, std::vector<short>& xab,
                                std::vector<int>& yofs, std::vector<short>& yab) {
  xofs.resize(dst_w); yofs.resize(dst_h); xab.resize(2 * (size_t)dst_w); yab.resize(2 * (size_t)dst_h);
  const double scale_x = 1.0 / ((double)dst_w / w), scale_y = 1.0 / ((double)dst_h / h);
  for (int dx = 0; dx < dst_w; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= w - 1) { fx = 0; sx = w - 1; }
    xofs[dx] = sx;
    xab[2 * dx] = sat_short((1.f - fx) * 2048.f);
    xab[2 * dx + 1] = sat_short(fx * 2048.f);
  }
  for (int dy = 0; dy < dst_h; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor(fy);
    fy -= sy;
    yofs[dy] = sy;
    yab[2 * dy] = sat_short((1.f - fy) * 2048.f);
    yab[2 * dy + 1] = sat_short(fy * 2048.f);
  }
}

int orbx_resize_linear(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, uint8_t* dst,
                       int dst_w, int dst_h, ptrdiff_t dst_stride) {
  if (!src || !dst || w <= 0 || h <= 0 || dst_w <= 0 || dst_h <= 0 || (channels != 1 && channels != 3 && channels != 4) ||
      src_stride < (ptrdiff_t)w * channels || dst_stride < (ptrdiff_t)dst_w * channels)
    return fail(ORBX_E_BADARG, "bad argument");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  std::vector<int> xofs, yofs;
  std::vector<short> xab, yab;
  build_resize_tables(w, h, dst_w, dst_h, xofs, xab, yofs, yab);
  ScratchBuf<uint8_t> ds, dd;
  ScratchBuf<int> dxo, dyo;
  ScratchBuf<short> dxa, dya;
  const size_t sp = (size_t)w * channels, dp = (size_t)dst_w * channels;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  chk(ds.alloc(sp * h)); chk(dd.alloc(dp * dst_h)); chk(dxo.alloc(dst_w)); chk(dyo.alloc(dst_h)); chk(dxa.alloc(2 * (size_t)dst_w));
  chk(dya.alloc(2 * (size_t)dst_h));
  if (e == hipSuccess) chk(hipMemcpy2D(ds.p, sp, src, (size_t)src_stride, sp, h, hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(dxo.p, xofs.data(), xofs.size() * sizeof(int), hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(dyo.p, yofs.data(), yofs.size() * sizeof(int), hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(dxa.p, xab.data(), xab.size() * sizeof(short), hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(dya.p, yab.data(), yab.size() * sizeof(short), hipMemcpyHostToDevice));
  if (e == hipSuccess)
    chk(launch_resize_generic(ds.p, w, h, (long long)sp, 0, channels, dd.p, dst_w, dst_h, (long long)dp, 0, dxo.p, dxa.p, dyo.p,
                              dya.p, 1, nullptr));
  if (e == hipSuccess) chk(hipDeviceSynchronize());
  if (e == hipSuccess) chk(hipMemcpy2D(dst, (size_t)dst_stride, dd.p, dp, dp, dst_h, hipMemcpyDeviceToHost));
  ds.free(); dd.free(); dxo.free(); dyo.free(); dxa.free(); dya.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

// ---- cv::remap / cv::CLAHE and the device-resident pre-processing chain ------------------------------------------------------
static int fill_clahe_args(ClaheArgs& a, int w, int h, double clip_limit, int tiles_x, int tiles_y) {
  if (tiles_x <= 0 || tiles_y <= 0 || tiles_x > 64 || tiles_y > 64 || w <= tiles_x || h <= tiles_y || !(clip_limit >= 0.0))
    return fail(ORBX_E_BADARG, "bad CLAHE arguments");
  int ew = w, eh = h;
  if (w % tiles_x != 0 || h % tiles_y != 0) {  // clahe.cpp: both axes are extended as soon as one does not divide
    ew = w + (tiles_x - w % tiles_x);
    eh = h + (tiles_y - h % tiles_y);
  }
  a.w = w; a.h = h; a.tilesX = tiles_x; a.tilesY = tiles_y;
  a.tw = ew / tiles_x; a.th = eh / tiles_y;
  const int area = a.tw * a.th;
  a.lutScale = (float)255 / area;
  a.clip = 0;
  if (clip_limit > 0.0) {
    a.clip = (int)(clip_limit * area / 256);
    if (a.clip < 1) a.clip = 1;
  }
  a.invTw = 1.0f / a.tw;
  a.invTh = 1.0f / a.th;
  return ORBX_OK;
}

int orbx_remap_linear(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, int channels, const float* map_x,
                      const float* map_y, ptrdiff_t map_stride, uint8_t* dst, int dst_w, int dst_h, ptrdiff_t dst_stride) {
  if (!src || !dst || !map_x || !map_y || w <= 0 || h <= 0 || dst_w <= 0 || dst_h <= 0 ||
      (channels != 1 && channels != 3 && channels != 4) || src_stride < (ptrdiff_t)w * channels || map_stride < dst_w ||
      dst_stride < (ptrdiff_t)dst_w * channels)
    return fail(ORBX_E_BADARG, "bad argument");
  if (w > 32767 || h > 32767) return fail(ORBX_E_UNSUPPORTED, "source larger than 32767 (cv::remap's short coordinates)");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<uint8_t> ds, dd;
  ScratchBuf<float> mx, my;
  const size_t sp = ((size_t)w * channels + 3) & ~(size_t)3, dp = ((size_t)dst_w * channels + 3) & ~(size_t)3;
  const size_t mp = ((size_t)dst_w + 3) & ~(size_t)3;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  chk(ds.alloc(sp * h)); chk(dd.alloc(dp * dst_h)); chk(mx.alloc(mp * dst_h)); chk(my.alloc(mp * dst_h));
  if (e == hipSuccess) chk(hipMemcpy2D(ds.p, sp, src, (size_t)src_stride, (size_t)w * channels, h, hipMemcpyHostToDevice));
  if (e == hipSuccess)
    chk(hipMemcpy2D(mx.p, mp * 4, map_x, (size_t)map_stride * 4, (size_t)dst_w * 4, dst_h, hipMemcpyHostToDevice));
  if (e == hipSuccess)
    chk(hipMemcpy2D(my.p, mp * 4, map_y, (size_t)map_stride * 4, (size_t)dst_w * 4, dst_h, hipMemcpyHostToDevice));
  RemapArgs a{};
  a.src = ds.p; a.sw = w; a.sh = h; a.cn = channels; a.srcPitch = (long long)sp; a.srcImgPitch = 0;
  a.mapx = mx.p; a.mapy = my.p; a.mapPitch = (long long)mp; a.mapImgPitch = 0; a.nMaps = 1;
  a.dst = dd.p; a.dw = dst_w; a.dh = dst_h; a.dstPitch = (long long)dp; a.dstImgPitch = 0;
  a.mapVec4 = 1; a.dstVec4 = 1;
  if (e == hipSuccess) chk(launch_remap(a, 1, nullptr));
  if (e == hipSuccess) chk(hipDeviceSynchronize());
  if (e == hipSuccess) chk(hipMemcpy2D(dst, (size_t)dst_stride, dd.p, dp, (size_t)dst_w * channels, dst_h, hipMemcpyDeviceToHost));
  ds.free(); dd.free(); mx.free(); my.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

int orbx_clahe(int device, const uint8_t* src, int w, int h, ptrdiff_t src_stride, double clip_limit, int tiles_x, int tiles_y,
               uint8_t* dst, ptrdiff_t dst_stride) {
  if (!src || !dst || w <= 0 || h <= 0 || src_stride < w || dst_stride < w) return fail(ORBX_E_BADARG, "bad argument");
  ClaheArgs a{};
  int rc = fill_clahe_args(a, w, h, clip_limit, tiles_x, tiles_y);
  if (rc != ORBX_OK) return rc;
  rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<uint8_t> ds, dd, lut;
  ScratchBuf<uint32_t> cells;
  const size_t p = ((size_t)w + 3) & ~(size_t)3;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  chk(ds.alloc(p * h)); chk(dd.alloc(p * h)); chk(lut.alloc((size_t)tiles_x * tiles_y * 256));
  chk(cells.alloc(clahe_cells_bytes(a, 1) / sizeof(uint32_t)));
  if (e == hipSuccess) chk(hipMemcpy2D(ds.p, p, src, (size_t)src_stride, (size_t)w, h, hipMemcpyHostToDevice));
  a.src = ds.p; a.srcPitch = (long long)p; a.srcImgPitch = 0;
  a.dst = dd.p; a.dstPitch = (long long)p; a.dstImgPitch = 0;
  a.lut = lut.p; a.srcVec4 = 1; a.dstVec4 = 1;
  if (e == hipSuccess) chk(launch_clahe(a, 1, cells.p, nullptr));
  if (e == hipSuccess) chk(hipDeviceSynchronize());
  if (e == hipSuccess) chk(hipMemcpy2D(dst, (size_t)dst_stride, dd.p, p, (size_t)w, h, hipMemcpyDeviceToHost));
  ds.free(); dd.free(); lut.free(); cells.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

struct orbx_preproc {
  orbx_preproc_params prm{};
  int device = 0, maxB = 0;
  bool doClahe = false, doRemap = false, doResize = false, doGray = false;
  ClaheArgs clahe{};
  DevBuf<float> d_mapx, d_mapy;
  long long mapPitch = 0;
  DevBuf<int> d_xofs, d_yofs;
  DevBuf<short> d_xab, d_yab;
  DevBuf<uint8_t> d_clahe, d_lut, d_geo, d_gray;
  DevBuf<uint32_t> d_cells;
  long long clahePitch = 0, geoPitch = 0, grayPitch = 0;
  int outW = 0, outH = 0;
  const uint8_t* out = nullptr;  // result of the last run (a stage buffer, or the caller's frames when nothing is enabled)
  long long outPitch = 0, outImgPitch = 0;
  ~orbx_preproc() {
    d_mapx.free(); d_mapy.free(); d_xofs.free(); d_yofs.free(); d_xab.free(); d_yab.free();
    d_clahe.free(); d_lut.free(); d_geo.free(); d_gray.free(); d_cells.free();
  }
};

int orbx_preproc_create(const orbx_preproc_params* p, int max_batch, int device, orbx_preproc** out) {
  if (!p || !out || max_batch <= 0) return fail(ORBX_E_BADARG, "null argument");
  *out = nullptr;
  if (p->src_w <= 0 || p->src_h <= 0 || (p->channels != 1 && p->channels != 3 && p->channels != 4))
    return fail(ORBX_E_BADARG, "bad source geometry");
  if (p->src_w > 32767 || p->src_h > 32767) return fail(ORBX_E_UNSUPPORTED, "source larger than 32767");
  const bool remap = p->map_x != nullptr || p->map_y != nullptr;
  if (remap && (!p->map_x || !p->map_y || p->n_maps <= 0 || p->out_w <= 0 || p->out_h <= 0))
    return fail(ORBX_E_BADARG, "remap needs map_x, map_y, n_maps and the output size");
  const bool resize = !remap && p->out_w > 0 && p->out_h > 0 && (p->out_w != p->src_w || p->out_h != p->src_h);
  if (!remap && ((p->out_w > 0) != (p->out_h > 0))) return fail(ORBX_E_BADARG, "bad output size");
  if (p->clahe_clip_limit < 0.0) return fail(ORBX_E_BADARG, "bad CLAHE clip limit");
  const bool clahe = p->clahe_tiles_x > 0 || p->clahe_tiles_y > 0;
  if (clahe && p->channels != 1) return fail(ORBX_E_UNSUPPORTED, "CLAHE needs single-channel frames (cv::CLAHE: CV_8UC1)");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  std::unique_ptr<orbx_preproc> pp(new (std::nothrow) orbx_preproc());
  if (!pp) return fail(ORBX_E_HIP, "out of memory");
  pp->prm = *p;
  pp->prm.map_x = pp->prm.map_y = nullptr;  // the handle keeps device copies only
  pp->device = device;
  pp->maxB = max_batch;
  pp->doClahe = clahe; pp->doRemap = remap; pp->doResize = resize; pp->doGray = p->channels != 1;
  pp->outW = (remap || resize) ? p->out_w : p->src_w;
  pp->outH = (remap || resize) ? p->out_h : p->src_h;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  const int cn = p->channels;
  if (clahe) {
    rc = fill_clahe_args(pp->clahe, p->src_w, p->src_h, p->clahe_clip_limit, p->clahe_tiles_x, p->clahe_tiles_y);
    if (rc != ORBX_OK) return rc;
    pp->clahePitch = ((long long)p->src_w + 3) & ~3ll;
    chk(pp->d_clahe.alloc((size_t)pp->clahePitch * p->src_h * max_batch));
    chk(pp->d_lut.alloc((size_t)p->clahe_tiles_x * p->clahe_tiles_y * 256 * max_batch));
    chk(pp->d_cells.alloc(clahe_cells_bytes(pp->clahe, max_batch) / sizeof(uint32_t)));
  }
  if (remap) {
    pp->mapPitch = ((long long)p->out_w + 3) & ~3ll;
    const size_t per = (size_t)pp->mapPitch * p->out_h;
    chk(pp->d_mapx.alloc(per * p->n_maps)); chk(pp->d_mapy.alloc(per * p->n_maps));
    const ptrdiff_t ms = p->map_stride > 0 ? p->map_stride : p->out_w;
    if (ms < p->out_w) return fail(ORBX_E_BADARG, "map_stride smaller than the output width");
    for (int m = 0; m < p->n_maps && e == hipSuccess; m++) {
      chk(hipMemcpy2D(pp->d_mapx.p + m * per, (size_t)pp->mapPitch * 4, p->map_x + (size_t)m * ms * p->out_h, (size_t)ms * 4,
                      (size_t)p->out_w * 4, p->out_h, hipMemcpyHostToDevice));
      chk(hipMemcpy2D(pp->d_mapy.p + m * per, (size_t)pp->mapPitch * 4, p->map_y + (size_t)m * ms * p->out_h, (size_t)ms * 4,
                      (size_t)p->out_w * 4, p->out_h, hipMemcpyHostToDevice));
    }
  }
  if (resize) {
    std::vector<int> xofs, yofs;
    std::vector<short> xab, yab;
    build_resize_tables(p->src_w, p->src_h, p->out_w, p->out_h, xofs, xab, yofs, yab);
    chk(pp->d_xofs.alloc(xofs.size())); chk(pp->d_yofs.alloc(yofs.size())); chk(pp->d_xab.alloc(xab.size())); chk(pp->d_yab.alloc(yab.size()));
    if (e == hipSuccess) chk(hipMemcpy(pp->d_xofs.p, xofs.data(), xofs.size() * sizeof(int), hipMemcpyHostToDevice));
    if (e == hipSuccess) chk(hipMemcpy(pp->d_yofs.p, yofs.data(), yofs.size() * sizeof(int), hipMemcpyHostToDevice));
    if (e == hipSuccess) chk(hipMemcpy(pp->d_xab.p, xab.data(), xab.size() * sizeof(short), hipMemcpyHostToDevice));
    if (e == hipSuccess) chk(hipMemcpy(pp->d_yab.p, yab.data(), yab.size() * sizeof(short), hipMemcpyHostToDevice));
  }
  if (remap || resize) {
    pp->geoPitch = ((long long)pp->outW * cn + 3) & ~3ll;
    chk(pp->d_geo.alloc((size_t)pp->geoPitch * pp->outH * max_batch));
  }
  if (pp->doGray) {
    pp->grayPitch = ((long long)pp->outW + 3) & ~3ll;
    chk(pp->d_gray.alloc((size_t)pp->grayPitch * pp->outH * max_batch));
  }
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  *out = pp.release();
  return ORBX_OK;
}

void orbx_preproc_destroy(orbx_preproc* pp) {
  if (!pp) return;
  (void)hipSetDevice(pp->device);
  delete pp;
}

// Enqueue the chain for n frames on stream s: [CLAHE] -> [remap | resize] -> [gray], the order the reference applies them
// (the TUM-VI examples equalise the frames they read, System::TrackStereo rectifies or resizes, Tracking::GrabImage*
// converts to gray last).
static int preproc_enqueue(orbx_preproc* pp, const uint8_t* d_frames, int n, ptrdiff_t row_pitch, ptrdiff_t image_pitch,
                           hipStream_t s) {
  const orbx_preproc_params& p = pp->prm;
  const int cn = p.channels;
  if (!d_frames || n <= 0) return fail(ORBX_E_EMPTY, "empty image");
  if (n > pp->maxB) return fail(ORBX_E_CAPACITY, "batch larger than max_batch");
  if (row_pitch < (ptrdiff_t)p.src_w * cn) return fail(ORBX_E_BADARG, "row pitch smaller than a row");
  const uint8_t* cur = d_frames;
  long long cp = row_pitch, cip = image_pitch;
  hipError_t e = hipSuccess;
  if (pp->doClahe) {
    ClaheArgs a = pp->clahe;
    a.src = cur; a.srcPitch = cp; a.srcImgPitch = cip;
    a.dst = pp->d_clahe.p; a.dstPitch = pp->clahePitch; a.dstImgPitch = pp->clahePitch * p.src_h;
    a.lut = pp->d_lut.p;
    a.srcVec4 = !(((uintptr_t)cur | (uintptr_t)cp | (uintptr_t)cip) & 3);
    a.dstVec4 = 1;
    e = launch_clahe(a, n, pp->d_cells.p, s);
    cur = a.dst; cp = a.dstPitch; cip = a.dstImgPitch;
  }
  if (e == hipSuccess && pp->doRemap) {
    RemapArgs a{};
    a.src = cur; a.sw = p.src_w; a.sh = p.src_h; a.cn = cn; a.srcPitch = cp; a.srcImgPitch = cip;
    a.mapx = pp->d_mapx.p; a.mapy = pp->d_mapy.p; a.mapPitch = pp->mapPitch; a.mapImgPitch = pp->mapPitch * pp->outH;
    a.nMaps = p.n_maps;
    a.dst = pp->d_geo.p; a.dw = pp->outW; a.dh = pp->outH; a.dstPitch = pp->geoPitch; a.dstImgPitch = pp->geoPitch * pp->outH;
    a.mapVec4 = 1; a.dstVec4 = 1;
    e = launch_remap(a, n, s);
    cur = a.dst; cp = a.dstPitch; cip = a.dstImgPitch;
  } else if (e == hipSuccess && pp->doResize) {
    e = launch_resize_generic(cur, p.src_w, p.src_h, cp, cip, cn, pp->d_geo.p, pp->outW, pp->outH, pp->geoPitch,
                              pp->geoPitch * pp->outH, pp->d_xofs.p, pp->d_xab.p, pp->d_yofs.p, pp->d_yab.p, n, s);
    cur = pp->d_geo.p; cp = pp->geoPitch; cip = pp->geoPitch * pp->outH;
  }
  if (e == hipSuccess && pp->doGray) {
    e = launch_cvt_gray(cur, pp->outW, pp->outH, cp, cip, cn, p.rgb_order ? 1 : 0, pp->d_gray.p, pp->grayPitch,
                        pp->grayPitch * pp->outH, n, s);
    cur = pp->d_gray.p; cp = pp->grayPitch; cip = pp->grayPitch * pp->outH;
  }
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  pp->out = cur; pp->outPitch = cp; pp->outImgPitch = cip;
  return ORBX_OK;
}

int orbx_preproc_run_device(orbx_preproc* pp, const uint8_t* d_frames, int n_frames, ptrdiff_t row_pitch,
                            ptrdiff_t image_pitch, const uint8_t** d_out, int* out_w, int* out_h, ptrdiff_t* out_row_pitch,
                            ptrdiff_t* out_image_pitch) {
  if (!pp) return fail(ORBX_E_BADARG, "null handle");
  int rc = set_device(pp->device);
  if (rc != ORBX_OK) return rc;
  rc = preproc_enqueue(pp, d_frames, n_frames, row_pitch, image_pitch, nullptr);
  if (rc != ORBX_OK) return rc;
  HIPC(hipStreamSynchronize(nullptr));
  if (d_out) *d_out = pp->out;
  if (out_w) *out_w = pp->outW;
  if (out_h) *out_h = pp->outH;
  if (out_row_pitch) *out_row_pitch = (ptrdiff_t)pp->outPitch;
  if (out_image_pitch) *out_image_pitch = (ptrdiff_t)pp->outImgPitch;
  return ORBX_OK;
}

int orbx_preproc_run(orbx_preproc* pp, const uint8_t* frame, ptrdiff_t stride, int map_index, uint8_t* dst, ptrdiff_t dst_stride) {
  if (!pp || !frame || !dst) return fail(ORBX_E_BADARG, "null argument");
  const orbx_preproc_params& p = pp->prm;
  if (stride < (ptrdiff_t)p.src_w * p.channels || dst_stride < pp->outW) return fail(ORBX_E_BADARG, "bad stride");
  if (pp->doRemap && (map_index < 0 || map_index >= p.n_maps)) return fail(ORBX_E_BADARG, "map index out of range");
  int rc = set_device(pp->device);
  if (rc != ORBX_OK) return rc;
  // frame slot `map_index` of a staging batch, so that image % n_maps selects the requested map
  const int slot = pp->doRemap ? map_index : 0;
  if (slot >= pp->maxB) return fail(ORBX_E_CAPACITY, "map index needs max_batch > map_index");
  ScratchBuf<uint8_t> ds;
  const size_t sp = ((size_t)p.src_w * p.channels + 3) & ~(size_t)3;
  hipError_t e = ds.alloc(sp * p.src_h * (slot + 1));
  if (e == hipSuccess)
    e = hipMemcpy2D(ds.p + sp * p.src_h * slot, sp, frame, (size_t)stride, (size_t)p.src_w * p.channels, p.src_h, hipMemcpyHostToDevice);
  if (e != hipSuccess) { ds.free(); return fail(ORBX_E_HIP, hipGetErrorString(e)); }
  if (slot > 0) (void)hipMemset(ds.p, 0, sp * p.src_h * slot);
  rc = preproc_enqueue(pp, ds.p, slot + 1, (ptrdiff_t)sp, (ptrdiff_t)(sp * p.src_h), nullptr);
  if (rc == ORBX_OK) {
    e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess)
      e = hipMemcpy2D(dst, (size_t)dst_stride, pp->out + pp->outImgPitch * slot, (size_t)pp->outPitch, (size_t)pp->outW, pp->outH,
                      hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(ORBX_E_HIP, hipGetErrorString(e));
  }
  ds.free();
  return rc;
}

int orbx_extract_batch_raw_device(orbx_extractor* ex, orbx_preproc* pp, const uint8_t* d_frames, int n_frames,
                                  ptrdiff_t row_pitch, ptrdiff_t image_pitch, const int32_t* lap) {
  if (!ex || !pp) return fail(ORBX_E_BADARG, "null handle");
  if (ex->device != pp->device) return fail(ORBX_E_BADARG, "extractor and pre-processor live on different devices");
  if (n_frames > ex->maxB) return fail(ORBX_E_CAPACITY, "batch larger than max_batch");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  rc = preproc_enqueue(pp, d_frames, n_frames, row_pitch, image_pitch, ex->stream);
  if (rc != ORBX_OK) return rc;
  if (((uintptr_t)pp->out & 3) || (pp->outPitch & 3) || (pp->outImgPitch & 3))
    return fail(ORBX_E_BADARG, "pass-through frames must be 4-byte aligned with 4-byte aligned pitches");
  return enqueue_extract(ex, pp->out, n_frames, pp->outW, pp->outH, (ptrdiff_t)pp->outPitch, (ptrdiff_t)pp->outImgPitch, lap);
}

// ---- bag of words -------------------------------------------------------------------------------------------
struct orbx_vocabulary {
  int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, nNodes = 0, nWords = 0;
  DevBuf<int> childStart, children, wordId;
  DevBuf<uint32_t> desc;
  DevBuf<double> weight;
  BowVoc view() const {
    BowVoc v{};
    v.childStart = childStart.p; v.children = children.p; v.desc = desc.p; v.weight = weight.p; v.wordId = wordId.p;
    v.L = L; v.nNodes = nNodes; v.scoring = scoring; v.weighting = weighting;
    return v;
  }
  ~orbx_vocabulary() { childStart.free(); children.free(); wordId.free(); desc.free(); weight.free(); }
};

int orbx_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                           const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights, orbx_vocabulary** out) {
  if (!out) return fail(ORBX_E_BADARG, "null argument");
  *out = nullptr;
  // the limits of TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1359)
  if (!parent || !is_leaf || !descriptors || !weights || n_nodes < 2 || k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 ||
      scoring > 5 || weighting < 0 || weighting > 3)
    return fail(ORBX_E_BADARG, "bad vocabulary arguments");
  std::vector<int> cnt((size_t)n_nodes + 1, 0), start((size_t)n_nodes + 1, 0), children((size_t)n_nodes - 1), word((size_t)n_nodes, -1);
  for (int i = 1; i < n_nodes; i++) {
    if (parent[i] < 0 || parent[i] >= i) return fail(ORBX_E_BADARG, "vocabulary: a node's parent must precede it");
    cnt[parent[i] + 1]++;
  }
  for (int i = 0; i < n_nodes; i++) start[i + 1] = start[i] + cnt[i + 1];
  std::vector<int> fill(start.begin(), start.end() - 1);
  int nWords = 0;
  for (int i = 1; i < n_nodes; i++) {
    children[fill[parent[i]]++] = i;  // file order, as m_nodes[pid].children.push_back(nid)
    const bool structuralLeaf = start[i + 1] == start[i];
    if ((is_leaf[i] != 0) != structuralLeaf) return fail(ORBX_E_BADARG, "vocabulary: leaf flags disagree with the tree");
    if (structuralLeaf) word[i] = nWords++;
    if (start[i + 1] - start[i] > 65535) return fail(ORBX_E_UNSUPPORTED, "vocabulary: more than 65535 children");
  }
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  std::unique_ptr<orbx_vocabulary> v(new (std::nothrow) orbx_vocabulary());
  if (!v) return fail(ORBX_E_HIP, "out of memory");
  v->device = device; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->nNodes = n_nodes; v->nWords = nWords;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  chk(v->childStart.alloc(start.size())); chk(v->children.alloc(children.size())); chk(v->wordId.alloc(word.size()));
  chk(v->desc.alloc((size_t)n_nodes * 8)); chk(v->weight.alloc(n_nodes));
  if (e == hipSuccess) chk(hipMemcpy(v->childStart.p, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(v->children.p, children.data(), children.size() * sizeof(int), hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(v->wordId.p, word.data(), word.size() * sizeof(int), hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(v->desc.p, descriptors, (size_t)n_nodes * 32, hipMemcpyHostToDevice));
  if (e == hipSuccess) chk(hipMemcpy(v->weight.p, weights, (size_t)n_nodes * sizeof(double), hipMemcpyHostToDevice));
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  *out = v.release();
  return ORBX_OK;
}

int orbx_vocabulary_load_text(int device, const char* path, orbx_vocabulary** out) {
  if (!path || !out) return fail(ORBX_E_BADARG, "null argument");
  *out = nullptr;
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(ORBX_E_BADARG, std::string("cannot open ") + path);
  std::string buf;
  {
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) buf.append(chunk, got);
  }
  std::fclose(f);
  const char* p = buf.c_str();
  char* end = nullptr;
  auto next_long = [&](long& v) { v = std::strtol(p, &end, 10); const bool ok = end != p; p = end; return ok; };
  long k, L, n1, n2;
  if (!next_long(k) || !next_long(L) || !next_long(n1) || !next_long(n2))
    return fail(ORBX_E_BADARG, "vocabulary: not a DBoW2 text file");
  std::vector<int32_t> parent(1, 0);
  std::vector<uint8_t> leaf(1, 0), desc(32, 0);
  std::vector<double> weight(1, 0.0);
  for (;;) {  // one node per line: parent isLeaf 32 descriptor bytes weight (TemplatedVocabulary.h:1378-1419); the phantom
    long pid, isLeaf;  // node the reference appends for a trailing empty line (uninitialised descriptor) is not created
    if (!next_long(pid) || !next_long(isLeaf)) break;
    uint8_t row[32];
    bool ok = true;
    for (int i = 0; i < 32 && ok; i++) {
      long b;
      ok = next_long(b);
      row[i] = (uint8_t)b;
    }
    if (!ok) break;
    const double w = std::strtod(p, &end);
    if (end == p) break;
    p = end;
    parent.push_back((int32_t)pid);
    leaf.push_back(isLeaf > 0);
    desc.insert(desc.end(), row, row + 32);
    weight.push_back(w);
  }
  return orbx_vocabulary_create(device, (int)k, (int)L, (int)n1, (int)n2, (int)parent.size(), parent.data(), leaf.data(),
                                desc.data(), weight.data(), out);
}

void orbx_vocabulary_destroy(orbx_vocabulary* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  delete v;
}

int orbx_vocabulary_info(const orbx_vocabulary* v, int32_t info[6]) {
  if (!v || !info) return fail(ORBX_E_BADARG, "null argument");
  info[0] = v->k; info[1] = v->L; info[2] = v->nNodes; info[3] = v->nWords; info[4] = v->scoring; info[5] = v->weighting;
  return ORBX_OK;
}

int orbx_bow_transform(const orbx_vocabulary* voc, const uint8_t* desc, int n, int levelsup, uint32_t* word_ids,
                       double* word_values, int* n_words, uint32_t* node_ids, int32_t* node_start, uint32_t* feature_idx,
                       int* n_nodes) {
  if (!voc || n < 0 || (n && !desc) || !n_words || !n_nodes || !node_start) return fail(ORBX_E_BADARG, "bad argument");
  if (n > kBowMaxFeatures) return fail(ORBX_E_CAPACITY, "more than 8192 features");
  int rc = set_device(voc->device);
  if (rc != ORBX_OK) return rc;
  *n_words = *n_nodes = 0;
  node_start[0] = 0;
  if (n == 0) return 0;
  Pack pk;
  const size_t N = (size_t)n;
  const size_t oD = pk.add(desc, N * 32);
  const size_t oWord = pk.add(nullptr, N * 4), oNode = pk.add(nullptr, N * 4), oWt = pk.add(nullptr, N * 8);
  // outputs in one area: values | words | nodes | feats | nodeStart | counts  -> one copy back
  const size_t oOut = pk.add(nullptr, N * 8 + 3 * N * 4 + (N + 1) * 4 + 3 * 4);
  const size_t rValues = 0, rWords = N * 8, rNodes = rWords + N * 4, rFeats = rNodes + N * 4, rStart = rFeats + N * 4,
               rCounts = rStart + (N + 1) * 4, outBytes = rCounts + 12;
  hipError_t e = pk.commit();
  BowArgs a{};
  a.voc = voc->view();
  a.desc = pk.ptr<uint8_t>(oD); a.descImgPitch = 0; a.counts = nullptr; a.n = n; a.cap = n; a.levelsup = levelsup;
  a.word = pk.ptr<int>(oWord); a.weight = pk.ptr<double>(oWt); a.node = pk.ptr<int>(oNode);
  uint8_t* out = pk.ptr<uint8_t>(oOut);
  a.values = reinterpret_cast<double*>(out + rValues); a.words = reinterpret_cast<uint32_t*>(out + rWords);
  a.nodes = reinterpret_cast<uint32_t*>(out + rNodes); a.feats = reinterpret_cast<uint32_t*>(out + rFeats);
  a.nodeStart = reinterpret_cast<int*>(out + rStart); a.outCounts = reinterpret_cast<int*>(out + rCounts);
  if (e == hipSuccess) e = launch_bow_transform(a, 1, nullptr);
  int cnt[3] = {0, 0, 0};
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOut, outBytes, &e);
    if (e == hipSuccess) {
      std::memcpy(cnt, h + rCounts, sizeof(cnt));
      if (cnt[0] && word_ids) std::memcpy(word_ids, h + rWords, (size_t)cnt[0] * 4);
      if (cnt[0] && word_values) std::memcpy(word_values, h + rValues, (size_t)cnt[0] * 8);
      if (cnt[1] && node_ids) std::memcpy(node_ids, h + rNodes, (size_t)cnt[1] * 4);
      std::memcpy(node_start, h + rStart, (size_t)(cnt[1] + 1) * 4);
      if (cnt[2] && feature_idx) std::memcpy(feature_idx, h + rFeats, (size_t)cnt[2] * 4);
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  *n_words = cnt[0];
  *n_nodes = cnt[1];
  return cnt[2];
}

int orbx_bow_transform_batch(orbx_extractor* ex, const orbx_vocabulary* voc, int levelsup) {
  if (!ex || !voc) return fail(ORBX_E_BADARG, "null handle");
  if (ex->device != voc->device) return fail(ORBX_E_BADARG, "extractor and vocabulary live on different devices");
  if (ex->lastN <= 0) return fail(ORBX_E_BADARG, "no extraction on this handle yet");
  const int cap = ex->gmax.outCap, B = ex->maxB;
  if (cap > kBowMaxFeatures) return fail(ORBX_E_CAPACITY, "more than 8192 features per image");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  if (!ex->d_bowWord.p) {
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    const size_t n = (size_t)cap * B;
    chk(ex->d_bowWord.alloc(n)); chk(ex->d_bowNode.alloc(n)); chk(ex->d_bowWeight.alloc(n)); chk(ex->d_bowValues.alloc(n));
    chk(ex->d_bowWords.alloc(n)); chk(ex->d_bowNodes.alloc(n)); chk(ex->d_bowFeats.alloc(n));
    chk(ex->d_bowStart.alloc((size_t)(cap + 1) * B)); chk(ex->d_bowCounts.alloc((size_t)3 * B));
    if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  }
  BowArgs a{};
  a.voc = voc->view();
  a.desc = ex->d_desc.p; a.descImgPitch = (long long)cap * 32; a.counts = ex->d_nOut.p; a.n = cap; a.cap = cap; a.levelsup = levelsup;
  a.word = ex->d_bowWord.p; a.weight = ex->d_bowWeight.p; a.node = ex->d_bowNode.p;
  a.words = ex->d_bowWords.p; a.values = ex->d_bowValues.p; a.nodes = ex->d_bowNodes.p; a.nodeStart = ex->d_bowStart.p;
  a.feats = ex->d_bowFeats.p; a.outCounts = ex->d_bowCounts.p;
  HIPC(launch_bow_transform(a, ex->lastN, ex->stream));
  ex->bowImages = ex->lastN;
  return ORBX_OK;
}

int orbx_bow_results_device(const orbx_extractor* ex, const uint32_t** d_word_ids, const double** d_word_values,
                            const uint32_t** d_node_ids, const int32_t** d_node_start, const uint32_t** d_feature_idx,
                            const int32_t** d_counts, int* cap) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (!ex->d_bowWord.p) return fail(ORBX_E_BADARG, "no orbx_bow_transform_batch on this handle yet");
  if (d_word_ids) *d_word_ids = ex->d_bowWords.p;
  if (d_word_values) *d_word_values = ex->d_bowValues.p;
  if (d_node_ids) *d_node_ids = ex->d_bowNodes.p;
  if (d_node_start) *d_node_start = ex->d_bowStart.p;
  if (d_feature_idx) *d_feature_idx = ex->d_bowFeats.p;
  if (d_counts) *d_counts = ex->d_bowCounts.p;
  if (cap) *cap = ex->gmax.outCap;
  return ORBX_OK;
}

int orbx_bow_download(orbx_extractor* ex, int image, uint32_t* word_ids, double* word_values, int* n_words, uint32_t* node_ids,
                      int32_t* node_start, uint32_t* feature_idx, int* n_nodes, int cap) {
  if (!ex || !n_words || !n_nodes) return fail(ORBX_E_BADARG, "null argument");
  if (!ex->d_bowWord.p || image < 0 || image >= ex->bowImages) return fail(ORBX_E_BADARG, "image index out of range");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  int cnt[3];
  HIPC(hipMemcpy(cnt, ex->d_bowCounts.p + 3 * image, sizeof(cnt), hipMemcpyDeviceToHost));
  *n_words = cnt[0];
  *n_nodes = cnt[1];
  if (cnt[2] > cap) return fail(ORBX_E_CAPACITY, "output buffers too small");
  const size_t o = (size_t)image * ex->gmax.outCap;
  if (cnt[0] && word_ids) HIPC(hipMemcpy(word_ids, ex->d_bowWords.p + o, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost));
  if (cnt[0] && word_values) HIPC(hipMemcpy(word_values, ex->d_bowValues.p + o, (size_t)cnt[0] * 8, hipMemcpyDeviceToHost));
  if (cnt[1] && node_ids) HIPC(hipMemcpy(node_ids, ex->d_bowNodes.p + o, (size_t)cnt[1] * 4, hipMemcpyDeviceToHost));
  if (node_start)
    HIPC(hipMemcpy(node_start, ex->d_bowStart.p + (size_t)image * (ex->gmax.outCap + 1), (size_t)(cnt[1] + 1) * 4, hipMemcpyDeviceToHost));
  if (cnt[2] && feature_idx) HIPC(hipMemcpy(feature_idx, ex->d_bowFeats.p + o, (size_t)cnt[2] * 4, hipMemcpyDeviceToHost));
  return cnt[2];
}

int orbx_search_by_bow(int device, const uint32_t* kf_node_ids, const int32_t* kf_node_start, const uint32_t* kf_feature_idx,
                       int n_kf_nodes, const orbx_keypoint* kf_kps, const uint8_t* kf_desc, const uint8_t* kf_valid, int n_kf,
                       const uint32_t* f_node_ids, const int32_t* f_node_start, const uint32_t* f_feature_idx, int n_f_nodes,
                       const orbx_keypoint* f_kps, const uint8_t* f_desc, int n_f, int n_left_f, float nnratio,
                       int check_orientation, int32_t* matches) {
  if (n_kf < 0 || n_f < 0 || n_kf_nodes < 0 || n_f_nodes < 0 || (n_f && !matches) ||
      (n_kf_nodes && (!kf_node_ids || !kf_node_start || !kf_feature_idx || !kf_kps || !kf_desc || !kf_valid)) ||
      (n_f_nodes && (!f_node_ids || !f_node_start || !f_feature_idx || !f_kps || !f_desc)))
    return fail(ORBX_E_BADARG, "bad argument");
  const int nkl = n_kf_nodes ? kf_node_start[n_kf_nodes] : 0, nfl = n_f_nodes ? f_node_start[n_f_nodes] : 0;
  if (nkl < 0 || nkl > n_kf || nfl < 0 || nfl > n_f) return fail(ORBX_E_BADARG, "feature vector larger than the frame");
  for (int i = 0; i < nkl; i++)
    if (kf_feature_idx[i] >= (uint32_t)n_kf) return fail(ORBX_E_BADARG, "keyframe feature index out of range");
  for (int i = 0; i < nfl; i++)
    if (f_feature_idx[i] >= (uint32_t)n_f) return fail(ORBX_E_BADARG, "frame feature index out of range");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  for (int i = 0; i < n_f; i++) matches[i] = -1;
  if (n_kf_nodes == 0 || n_f_nodes == 0 || n_f == 0) return 0;
  Pack pk;
  const size_t oKn = pk.add(kf_node_ids, (size_t)n_kf_nodes * 4), oKs = pk.add(kf_node_start, ((size_t)n_kf_nodes + 1) * 4);
  const size_t oKf = pk.add(kf_feature_idx, (size_t)nkl * 4), oKd = pk.add(kf_desc, (size_t)n_kf * 32);
  const size_t oKv = pk.add(kf_valid, n_kf), oKk = pk.add(kf_kps, (size_t)n_kf * sizeof(orbx_keypoint));
  const size_t oFn = pk.add(f_node_ids, (size_t)n_f_nodes * 4), oFs = pk.add(f_node_start, ((size_t)n_f_nodes + 1) * 4);
  const size_t oFf = pk.add(f_feature_idx, (size_t)nfl * 4), oFd = pk.add(f_desc, (size_t)n_f * 32);
  const size_t oFk = pk.add(f_kps, (size_t)n_f * sizeof(orbx_keypoint));
  const size_t oBin = pk.add(nullptr, (size_t)n_f * 4), oFlags = pk.add(nullptr, 33 * 4);
  const size_t oOut = pk.add(nullptr, ((size_t)n_f + 1) * 4);  // result, then the matches: one copy back
  hipError_t e = pk.commit();
  BowMatchArgs a{};
  a.kfNodes = pk.ptr<uint32_t>(oKn); a.kfStart = pk.ptr<int>(oKs); a.kfFeat = pk.ptr<uint32_t>(oKf); a.nKfNodes = n_kf_nodes;
  a.kfDesc = pk.ptr<uint32_t>(oKd); a.kfKps = pk.ptr<orbx_keypoint>(oKk); a.kfValid = pk.ptr<uint8_t>(oKv);
  a.fNodes = pk.ptr<uint32_t>(oFn); a.fStart = pk.ptr<int>(oFs); a.fFeat = pk.ptr<uint32_t>(oFf); a.nFNodes = n_f_nodes;
  a.fDesc = pk.ptr<uint32_t>(oFd); a.fKps = pk.ptr<orbx_keypoint>(oFk); a.nF = n_f; a.nLeftF = n_left_f;
  a.nnratio = nnratio; a.checkOri = check_orientation ? 1 : 0;
  a.result = pk.ptr<int>(oOut); a.match = pk.ptr<int>(oOut) + 1; a.bin = pk.ptr<int>(oBin); a.flags = pk.ptr<int>(oFlags);
  if (e == hipSuccess) e = launch_bow_match(a, nullptr);
  int n = 0;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOut, ((size_t)n_f + 1) * 4, &e);
    if (e == hipSuccess) {
      std::memcpy(&n, h, 4);
      std::memcpy(matches, h + 4, (size_t)n_f * 4);
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  if (n < 0) return fail(ORBX_E_UNSUPPORTED, "a vocabulary node holds more than 4096 frame features");
  return n;
}

int orbx_preproc_output_size(const orbx_preproc* pp, int* out_w, int* out_h) {
  if (!pp) return fail(ORBX_E_BADARG, "null handle");
  if (out_w) *out_w = pp->outW;
  if (out_h) *out_h = pp->outH;
  return ORBX_OK;
}

static int fill_undistort_args(UndistortArgs& a, const float K[4], const float* dist, int n_dist) {
  if (!K || n_dist < 0 || n_dist > 14 || (n_dist && !dist)) return fail(ORBX_E_BADARG, "bad camera arguments");
  if (!(K[0] != 0.f) || !(K[1] != 0.f)) return fail(ORBX_E_BADARG, "fx / fy must be non-zero");
  for (int i = 12; i < n_dist; i++)
    if (dist[i] != 0.f) return fail(ORBX_E_UNSUPPORTED, "tilted-sensor distortion terms are not supported");
  for (int i = 0; i < 4; i++) a.K[i] = K[i];
  for (int i = 0; i < 12; i++) a.k[i] = i < n_dist ? dist[i] : 0.f;
  a.hasDist = n_dist > 0;
  return ORBX_OK;
}

int orbx_undistort_keypoints(int device, const orbx_keypoint* kps, int n, const float K[4], const float* dist, int n_dist,
                             orbx_keypoint* out) {
  if (n < 0 || (n && (!kps || !out))) return fail(ORBX_E_BADARG, "bad argument");
  UndistortArgs a{};
  int rc = fill_undistort_args(a, K, dist, n_dist);
  if (rc != ORBX_OK) return rc;
  rc = set_device(device);  // a device routine even for the identity case: no GPU is an error, never a host path
  if (rc != ORBX_OK) return rc;
  if (out != kps && n) std::memmove(static_cast<void*>(out), kps, (size_t)n * sizeof(orbx_keypoint));
  if (n == 0 || n_dist == 0 || dist[0] == 0.0f) return ORBX_OK;  // src/Frame.cc:854-857
  ScratchBuf<orbx_keypoint> d;
  hipError_t e = d.alloc(n);
  if (e == hipSuccess) e = hipMemcpy(d.p, out, (size_t)n * sizeof(orbx_keypoint), hipMemcpyHostToDevice);
  a.in = reinterpret_cast<const float*>(d.p);
  a.out = reinterpret_cast<float*>(d.p);
  a.n = n;
  a.stride = sizeof(orbx_keypoint) / sizeof(float);
  if (e == hipSuccess) e = launch_undistort(a, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, d.p, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost);
  d.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

int orbx_compute_image_bounds(int device, int cols, int rows, const float K[4], const float* dist, int n_dist,
                              float bounds[4]) {
  if (!bounds || cols <= 0 || rows <= 0) return fail(ORBX_E_BADARG, "bad argument");
  UndistortArgs a{};
  int rc = fill_undistort_args(a, K, dist, n_dist);
  if (rc != ORBX_OK) return rc;
  rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  if (n_dist == 0 || dist[0] == 0.0f) {  // src/Frame.cc:913-918
    bounds[0] = 0.f; bounds[1] = 0.f; bounds[2] = (float)cols; bounds[3] = (float)rows;
    return ORBX_OK;
  }
  float c[8] = {0.f, 0.f, (float)cols, 0.f, 0.f, (float)rows, (float)cols, (float)rows};
  ScratchBuf<float> d;
  hipError_t e = d.alloc(8);
  if (e == hipSuccess) e = hipMemcpy(d.p, c, sizeof(c), hipMemcpyHostToDevice);
  a.in = d.p; a.out = d.p; a.n = 4; a.stride = 2;
  if (e == hipSuccess) e = launch_undistort(a, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(c, d.p, sizeof(c), hipMemcpyDeviceToHost);
  d.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  bounds[0] = std::min(c[0], c[4]);  // mnMinX (:907)
  bounds[2] = std::max(c[2], c[6]);  // mnMaxX
  bounds[1] = std::min(c[1], c[3]);  // mnMinY
  bounds[3] = std::max(c[5], c[7]);  // mnMaxY
  return ORBX_OK;
}

int orbx_search_for_initialization(int device, const orbx_keypoint* kps1, const uint8_t* desc1, int n1,
                                   const orbx_keypoint* kps2, const uint8_t* desc2, int n2, float min_x,
                                   float min_y, float max_x, float max_y, float* prev_matched,
                                   int32_t* matches12, int window_size, float nnratio, int check_orientation) {
  if (n1 < 0 || n2 < 0 || (n1 && (!kps1 || !desc1 || !prev_matched || !matches12)) || (n2 && (!kps2 || !desc2)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n1 == 0) return 0;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<int> cellStart, cellItems, candOff, candIdx, candDist, mdist, m21;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  // one packed upload; vbPrevMatched | result | vnMatches12 are contiguous and come back in one copy
  Pack pk;
  const size_t oK1 = pk.add(kps1, (size_t)n1 * sizeof(orbx_keypoint)), oK2 = pk.add(kps2, (size_t)std::max(n2, 1) * sizeof(orbx_keypoint));
  const size_t oD1 = pk.add(desc1, (size_t)n1 * 32), oD2 = pk.add(desc2, (size_t)std::max(n2, 1) * 32);
  const size_t oPrev = pk.add(prev_matched, (size_t)n1 * 2 * sizeof(float)), oRes = pk.add(nullptr, 2 * sizeof(int));
  const size_t oM12 = pk.add(nullptr, (size_t)n1 * sizeof(int));
  const size_t outBytes = oM12 + (size_t)n1 * sizeof(int) - oPrev;
  chk(pk.commit());
  struct { orbx_keypoint* p; } k1{pk.ptr<orbx_keypoint>(oK1)}, k2{pk.ptr<orbx_keypoint>(oK2)};
  struct { uint8_t* p; } d1{pk.ptr<uint8_t>(oD1)}, d2{pk.ptr<uint8_t>(oD2)};
  struct { float* p; } prev{pk.ptr<float>(oPrev)};
  struct { int* p; } m12{pk.ptr<int>(oM12)}, result{pk.ptr<int>(oRes)};
  chk(cellStart.alloc(64 * 48 + 1)); chk(cellItems.alloc(std::max(n2, 1))); chk(candOff.alloc(n1 + 1));
  chk(mdist.alloc(std::max(n2, 1))); chk(m21.alloc(std::max(n2, 1)));
  InitArgs a{};
  a.k1 = k1.p; a.k2 = k2.p; a.d1 = d1.p; a.d2 = d2.p; a.n1 = n1; a.n2 = n2;
  a.minX = min_x; a.minY = min_y;
  a.invW = 64.f / (max_x - min_x);  // mfGridElementWidthInv, src/Frame.cc:243
  a.invH = 48.f / (max_y - min_y);
  a.prev = prev.p; a.matches12 = m12.p; a.window = window_size; a.nnratio = nnratio;
  a.checkOri = check_orientation;
  a.cellStart = cellStart.p; a.cellItems = cellItems.p; a.candOff = candOff.p;
  a.matchedDist = mdist.p; a.matches21 = m21.p; a.result = result.p;
  a.candCap = 1 << 30;
  int total = 0, res[2] = {0, 0};
  if (e == hipSuccess) chk(launch_search_init(a, nullptr));
  if (e == hipSuccess) chk(hipDeviceSynchronize());
  if (e == hipSuccess) chk(hipMemcpy(&total, candOff.p + n1, sizeof(int), hipMemcpyDeviceToHost));
  if (e == hipSuccess) {
    chk(candIdx.alloc((size_t)std::max(total, 1)));
    chk(candDist.alloc((size_t)std::max(total, 1)));
    a.candIdx = candIdx.p;
    a.candDist = candDist.p;
    a.candCap = std::max(total, 1);
  }
  // resolve: parallel fixed-point rounds (k_init_round), serial walk as fallback / ORBX_PROJ_SERIAL=1 cross-check
  ScratchBuf<int2> cl0, cl1, cr0, cr1;
  ScratchBuf<int> nc0, nc1, fl;
  static const bool forceSerial = getenv("ORBX_PROJ_SERIAL") && atoi(getenv("ORBX_PROJ_SERIAL")) != 0;
  bool done = false;
  int lastRound = 0;
  if (e == hipSuccess) chk(launch_search_init_cands_fill(a, nullptr));
  if (!forceSerial && n2 > 0) {
    chk(cl0.alloc(n1)); chk(cl1.alloc(n1)); chk(cr0.alloc((size_t)n2 * kFeWriters)); chk(cr1.alloc((size_t)n2 * kFeWriters));
    chk(nc0.alloc(n2)); chk(nc1.alloc(n2)); chk(fl.alloc(40));
    a.claim[0] = cl0.p; a.claim[1] = cl1.p; a.claimers[0] = cr0.p; a.claimers[1] = cr1.p;
    a.nclaimers[0] = nc0.p; a.nclaimers[1] = nc1.p; a.flags = fl.p;
    for (int r = 0; r < 48 && e == hipSuccess && !done; r += 4) {
      chk(launch_search_init_rounds(a, r, 4, nullptr));
      int st[2] = {1, 0};
      if (e == hipSuccess) chk(hipMemcpy(st, fl.p, sizeof(st), hipMemcpyDeviceToHost));  // synchronises
      if (st[1]) break;
      done = st[0] == 0;
      lastRound = r + 3;
    }
  }
  if (e == hipSuccess) chk(done ? launch_search_init_finish(a, lastRound, nullptr) : launch_search_init_resolve_serial(a, nullptr));
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oPrev, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      std::memcpy(prev_matched, h, (size_t)n1 * 2 * sizeof(float));
      std::memcpy(res, h + (oRes - oPrev), sizeof(res));
      std::memcpy(matches12, h + (oM12 - oPrev), (size_t)n1 * sizeof(int));
    }
  }
  cl0.free(); cl1.free(); cr0.free(); cr1.free(); nc0.free(); nc1.free(); fl.free();
  pk.release(); cellStart.free(); cellItems.free();
  candOff.free(); candIdx.free(); candDist.free(); mdist.free(); m21.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return res[0];
}

int orbx_features_in_area(int device, const orbx_keypoint* kps, int n, float min_x, float min_y, float max_x,
                          float max_y, const float* queries, int n_queries, int32_t* offsets, int32_t* indices,
                          int indices_cap, int32_t* grid_cell_start, int32_t* grid_items) {
  if (n < 0 || n_queries < 0 || (n && !kps) || (n_queries && (!queries || !offsets)))
    return fail(ORBX_E_BADARG, "bad argument");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<orbx_keypoint> k;
  ScratchBuf<float> q;
  ScratchBuf<int> cellStart, cellItems, qOff, out, mdist, m21, m12, result;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  const int nn = std::max(n, 1), nq = std::max(n_queries, 1);
  chk(k.alloc(nn)); chk(q.alloc((size_t)nq * 5)); chk(cellStart.alloc(64 * 48 + 1)); chk(cellItems.alloc(nn));
  chk(qOff.alloc(nq + 1)); chk(mdist.alloc(nn)); chk(m21.alloc(nn)); chk(m12.alloc(1)); chk(result.alloc(2));
  if (e == hipSuccess && n) chk(hipMemcpy(k.p, kps, (size_t)n * sizeof(orbx_keypoint), hipMemcpyHostToDevice));
  if (e == hipSuccess && n_queries) chk(hipMemcpy(q.p, queries, (size_t)n_queries * 5 * sizeof(float), hipMemcpyHostToDevice));
  InitArgs a{};
  a.k2 = k.p; a.n2 = n; a.n1 = 0;
  a.minX = min_x; a.minY = min_y;
  a.invW = 64.f / (max_x - min_x);
  a.invH = 48.f / (max_y - min_y);
  a.cellStart = cellStart.p; a.cellItems = cellItems.p; a.matchedDist = mdist.p; a.matches21 = m21.p;
  a.matches12 = m12.p; a.result = result.p; a.candOff = qOff.p; a.candCap = 1 << 30;
  int total = 0;
  if (e == hipSuccess) chk(launch_grid_build(a, nullptr));
  if (e == hipSuccess && n_queries) {
    chk(launch_area_query(a, q.p, n_queries, qOff.p, nullptr, 0, nullptr));
    a.n1 = n_queries;  // k_init_scan scans candOff[0..n1)
    if (e == hipSuccess) chk(launch_scan_offsets(a, nullptr));
    if (e == hipSuccess) chk(hipDeviceSynchronize());
    if (e == hipSuccess) chk(hipMemcpy(&total, qOff.p + n_queries, sizeof(int), hipMemcpyDeviceToHost));
    if (e == hipSuccess) chk(hipMemcpy(offsets, qOff.p, (size_t)(n_queries + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (e == hipSuccess && total > 0 && indices && total <= indices_cap) {
      chk(out.alloc(total));
      if (e == hipSuccess) chk(launch_area_query(a, q.p, n_queries, qOff.p, out.p, 1, nullptr));
      if (e == hipSuccess) chk(hipDeviceSynchronize());
      if (e == hipSuccess) chk(hipMemcpy(indices, out.p, (size_t)total * sizeof(int), hipMemcpyDeviceToHost));
    }
  }
  if (e == hipSuccess) chk(hipDeviceSynchronize());
  if (e == hipSuccess && grid_cell_start)
    chk(hipMemcpy(grid_cell_start, cellStart.p, (64 * 48 + 1) * sizeof(int), hipMemcpyDeviceToHost));
  if (e == hipSuccess && grid_items && n) chk(hipMemcpy(grid_items, cellItems.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  k.free(); q.free(); cellStart.free(); cellItems.free(); qOff.free(); out.free(); mdist.free(); m21.free(); m12.free();
  result.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  if (indices && total > indices_cap) return fail(ORBX_E_CAPACITY, "indices buffer too small");
  return total;
}

namespace {
int search_by_projection_impl(int device, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right, int n,
                              float min_x, float min_y, float max_x, float max_y, const float* scale_factors, int nlevels,
                              const orbx_map_point_view* map_points, const orbx_projected_point* points, int n_points,
                              float th, int far_points, float th_far_points, float nnratio, int check_ori,
                              uint8_t* occupied, int32_t* match) {
  const int mode = points ? 1 : 0;
  if (n == 0) return 0;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<int> cellStart, cellItems, candOff, candIdx, candDist, mdist, m21, m12, taker0, taker1, choice, flags;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  const int nm = std::max(n_points, 1);
  // inputs in one packed upload; occupied | result | match are contiguous so that they come back in one copy
  Pack pk;
  const size_t oK = pk.add(kps_un, (size_t)n * sizeof(orbx_keypoint)), oD = pk.add(desc, (size_t)n * 32);
  const size_t oUr = pk.add(u_right, (size_t)n * sizeof(float));
  const size_t oSf = pk.add(scale_factors, (size_t)std::max(nlevels, 1) * sizeof(float));
  const size_t oMp = pk.add(mode == 0 && n_points ? map_points : nullptr, (size_t)nm * sizeof(orbx_map_point_view));
  const size_t oPp = pk.add(mode == 1 && n_points ? points : nullptr, (size_t)nm * sizeof(orbx_projected_point));
  const size_t oOcc = pk.add(occupied, n), oRes = pk.add(nullptr, 2 * sizeof(int)), oMt = pk.add(nullptr, (size_t)n * sizeof(int));
  const size_t outBytes = oMt + (size_t)n * sizeof(int) - oOcc;
  chk(pk.commit());
  struct { orbx_keypoint* p; } k{pk.ptr<orbx_keypoint>(oK)};
  struct { uint8_t* p; } d{pk.ptr<uint8_t>(oD)}, occ{pk.ptr<uint8_t>(oOcc)};
  struct { float* p; } ur{pk.ptr<float>(oUr)}, sf{pk.ptr<float>(oSf)};
  struct { orbx_map_point_view* p; } mp{pk.ptr<orbx_map_point_view>(oMp)};
  struct { orbx_projected_point* p; } pp{pk.ptr<orbx_projected_point>(oPp)};
  struct { int* p; } mt{pk.ptr<int>(oMt)}, result{pk.ptr<int>(oRes)};
  chk(taker0.alloc(n)); chk(taker1.alloc(n)); chk(choice.alloc(nm)); chk(flags.alloc(40));
  chk(cellStart.alloc(64 * 48 + 1)); chk(cellItems.alloc(n));
  chk(candOff.alloc(nm + 1)); chk(mdist.alloc(n)); chk(m21.alloc(n)); chk(m12.alloc(1));
  ProjArgs a{};
  a.grid.k2 = k.p; a.grid.n2 = n; a.grid.n1 = 0;
  a.grid.minX = min_x; a.grid.minY = min_y;
  a.grid.invW = 64.f / (max_x - min_x);
  a.grid.invH = 48.f / (max_y - min_y);
  a.grid.cellStart = cellStart.p; a.grid.cellItems = cellItems.p; a.grid.matchedDist = mdist.p; a.grid.matches21 = m21.p;
  a.grid.matches12 = m12.p; a.grid.result = result.p; a.grid.candOff = candOff.p; a.grid.candCap = 1 << 30;
  a.desc = d.p; a.uRight = u_right ? ur.p : nullptr; a.scale = sf.p; a.mps = mp.p; a.pts = pp.p; a.nmp = n_points;
  a.mode = mode; a.checkOri = check_ori;
  a.th = th; a.thFar = th_far_points; a.nnratio = nnratio; a.far = far_points;
  a.occupied = occ.p; a.match = mt.p; a.candOff = candOff.p; a.result = result.p; a.candCap = 1 << 30;
  int total = 0, res[2] = {0, 0};
  if (e == hipSuccess) chk(launch_proj_count(a, nullptr));
  if (e == hipSuccess) chk(hipDeviceSynchronize());
  if (e == hipSuccess && n_points) chk(hipMemcpy(&total, candOff.p + n_points, sizeof(int), hipMemcpyDeviceToHost));
  if (e == hipSuccess) {
    chk(candIdx.alloc((size_t)std::max(total, 1)));
    chk(candDist.alloc((size_t)std::max(total, 1)));
    a.candIdx = candIdx.p; a.candDist = candDist.p; a.candCap = std::max(total, 1);
  }
  a.taker[0] = taker0.p; a.taker[1] = taker1.p; a.choice = choice.p; a.flags = flags.p;
  // Resolve: rounds of the parallel fixed-point iteration (k_proj_round) until a round changes nothing; the one-wave
  // serial walk stays as the fallback for a pathological claim chain (ORBX_PROJ_SERIAL=1 forces it, for the tests).
  static const bool forceSerial = getenv("ORBX_PROJ_SERIAL") && atoi(getenv("ORBX_PROJ_SERIAL")) != 0;
  if (e == hipSuccess) chk(launch_proj_cands_fill(a, nullptr));
  bool done = false;
  if (!forceSerial && n_points > 0) {
    for (int r = 0; r < 48 && e == hipSuccess && !done; r += 4) {
      chk(launch_proj_rounds(a, r, 4, nullptr));
      int changed = 1;
      if (e == hipSuccess) chk(hipMemcpy(&changed, flags.p, sizeof(int), hipMemcpyDeviceToHost));  // synchronises
      done = changed == 0;
    }
  }
  if (e == hipSuccess) chk(done ? launch_proj_finish(a, 0, nullptr) : launch_proj_resolve_serial(a, nullptr));
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOcc, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      std::memcpy(occupied, h, n);
      std::memcpy(res, h + (oRes - oOcc), sizeof(res));
      std::memcpy(match, h + (oMt - oOcc), (size_t)n * sizeof(int));
    }
  }
  pk.release(); cellStart.free(); cellItems.free();
  candOff.free(); candIdx.free(); candDist.free(); mdist.free(); m21.free(); m12.free();
  taker0.free(); taker1.free(); choice.free(); flags.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return res[0];
}
}  // namespace

int orbx_search_by_projection(int device, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right,
                              int n, float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                              int nlevels, const orbx_map_point_view* map_points, int n_map_points, float th,
                              int far_points, float th_far_points, float nnratio, uint8_t* occupied, int32_t* match) {
  if (n < 0 || n_map_points < 0 || nlevels < 1 || !scale_factors || (n && (!kps_un || !desc || !occupied || !match)) ||
      (n_map_points && !map_points))
    return fail(ORBX_E_BADARG, "bad argument");
  for (int i = 0; i < n_map_points; i++)
    if (map_points[i].predicted_level < 0 || map_points[i].predicted_level >= nlevels)
      return fail(ORBX_E_BADARG, "map point with a predicted level outside [0, nlevels)");
  static const orbx_map_point_view dummy{};
  return search_by_projection_impl(device, kps_un, desc, u_right, n, min_x, min_y, max_x, max_y, scale_factors, nlevels,
                                   map_points ? map_points : &dummy, nullptr, n_map_points, th, far_points,
                                   th_far_points, nnratio, 0, occupied, match);
}

int orbx_search_by_projection_frame(int device, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right,
                                    int n, float min_x, float min_y, float max_x, float max_y,
                                    const orbx_projected_point* points, int n_points, int check_orientation,
                                    uint8_t* occupied, int32_t* match) {
  if (n < 0 || n_points < 0 || (n && (!kps_un || !desc || !occupied || !match)) || (n_points && !points))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n_points > 15000) return fail(ORBX_E_CAPACITY, "more than 15000 projected points");
  static const orbx_projected_point dummy{};
  return search_by_projection_impl(device, kps_un, desc, u_right, n, min_x, min_y, max_x, max_y, nullptr, 0, nullptr,
                                   points ? points : &dummy, n_points, 1.0f, 0, 0.f, 0.f, check_orientation, occupied, match);
}

namespace {
// One side (left or right camera) of a stereo-fisheye projection search: grid + candidate lists on the device.
struct ProjSide {
  ScratchBuf<int> cellStart, cellItems, candOff, candIdx, candDist, mdist, m21, m12, result;
  orbx_map_point_view* mpp = nullptr;  // views of this camera inside the call's packed upload
  orbx_projected_point* ppp = nullptr;
  ProjArgs a{};
  void release() {
    cellStart.free(); cellItems.free(); candOff.free(); candIdx.free(); candDist.free(); mdist.free(); m21.free();
    m12.free(); result.free();
  }
};

int search_by_projection_fisheye_impl(int device, const orbx_keypoint* kps, const uint8_t* desc, int n_left, int n_right,
                                      float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                                      int nlevels, const orbx_map_point_view* viewsL, const orbx_map_point_view* viewsR,
                                      const orbx_projected_point* ptsL, const orbx_projected_point* ptsR, int n_points,
                                      float th, int far_points, float th_far_points, float nnratio, int check_ori,
                                      const int32_t* l2r, const int32_t* r2l, uint8_t* occupied, int32_t* match) {
  const int mode = ptsL ? 1 : 0, n = n_left + n_right;
  if (n == 0) return 0;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ProjSide S[2];
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  const int nm = std::max(n_points, 1);
  // one packed upload of every input (both cameras' views included); occupied | result | match come back in one copy
  Pack pk;
  const size_t oK = pk.add(kps, (size_t)n * sizeof(orbx_keypoint)), oD = pk.add(desc, (size_t)n * 32);
  const size_t oSf = pk.add(scale_factors, (size_t)std::max(nlevels, 1) * sizeof(float));
  const size_t oA12 = pk.add(n_left ? l2r : nullptr, (size_t)std::max(n_left, 1) * 4);
  const size_t oA21 = pk.add(n_right ? r2l : nullptr, (size_t)std::max(n_right, 1) * 4);
  size_t oMp[2], oPp[2];
  for (int side = 0; side < 2; side++) {
    oMp[side] = pk.add(n_points && mode == 0 ? (side ? viewsR : viewsL) : nullptr, (size_t)nm * sizeof(orbx_map_point_view));
    oPp[side] = pk.add(n_points && mode == 1 ? (side ? ptsR : ptsL) : nullptr, (size_t)nm * sizeof(orbx_projected_point));
  }
  const size_t oOcc = pk.add(occupied, n), oRes = pk.add(nullptr, 2 * sizeof(int)), oMt = pk.add(nullptr, (size_t)n * sizeof(int));
  const size_t outBytes = oMt + (size_t)n * sizeof(int) - oOcc;
  chk(pk.commit());
  struct { orbx_keypoint* p; } k{pk.ptr<orbx_keypoint>(oK)};
  struct { uint8_t* p; } d{pk.ptr<uint8_t>(oD)}, occ{pk.ptr<uint8_t>(oOcc)};
  struct { float* p; } sf{pk.ptr<float>(oSf)};
  struct { int* p; } a12{pk.ptr<int>(oA12)}, a21{pk.ptr<int>(oA21)}, mt{pk.ptr<int>(oMt)}, res{pk.ptr<int>(oRes)};
  for (int side = 0; side < 2 && e == hipSuccess; side++) {
    ProjSide& P = S[side];
    const int ns = side ? n_right : n_left, first = side ? n_left : 0;
    chk(P.cellStart.alloc(64 * 48 + 1)); chk(P.cellItems.alloc(std::max(ns, 1))); chk(P.candOff.alloc(nm + 1));
    chk(P.mdist.alloc(std::max(ns, 1))); chk(P.m21.alloc(std::max(ns, 1))); chk(P.m12.alloc(1)); chk(P.result.alloc(2));
    P.mpp = pk.ptr<orbx_map_point_view>(oMp[side]);
    P.ppp = pk.ptr<orbx_projected_point>(oPp[side]);
    ProjArgs& a = P.a;
    a.grid.k2 = k.p + first; a.grid.n2 = ns; a.grid.n1 = 0;
    a.grid.minX = min_x; a.grid.minY = min_y;
    a.grid.invW = 64.f / (max_x - min_x);
    a.grid.invH = 48.f / (max_y - min_y);
    a.grid.cellStart = P.cellStart.p; a.grid.cellItems = P.cellItems.p; a.grid.matchedDist = P.mdist.p;
    a.grid.matches21 = P.m21.p; a.grid.matches12 = P.m12.p; a.grid.result = P.result.p; a.grid.candOff = P.candOff.p;
    a.grid.candCap = 1 << 30;
    a.desc = d.p + (size_t)first * 32; a.uRight = nullptr;  // no mvuRight gate when F.Nleft != -1 (:90, :1667)
    a.scale = sf.p; a.mps = P.mpp; a.pts = P.ppp; a.nmp = n_points; a.mode = mode; a.checkOri = check_ori;
    a.th = side ? 1.0f : th;  // the right-camera radius is not scaled by th (:144)
    a.thFar = th_far_points; a.nnratio = nnratio; a.far = far_points;
    a.occupied = occ.p + first; a.match = mt.p + first; a.candOff = P.candOff.p; a.result = P.result.p; a.candCap = 1 << 30;
    if (ns > 0) chk(launch_proj_count(a, nullptr));
    else chk(hipMemset(P.candOff.p, 0, (size_t)(nm + 1) * sizeof(int)));
  }
  if (e == hipSuccess) chk(hipDeviceSynchronize());
  for (int side = 0; side < 2 && e == hipSuccess; side++) {
    ProjSide& P = S[side];
    int total = 0;
    if (n_points) chk(hipMemcpy(&total, P.candOff.p + n_points, sizeof(int), hipMemcpyDeviceToHost));
    chk(P.candIdx.alloc((size_t)std::max(total, 1)));
    chk(P.candDist.alloc((size_t)std::max(total, 1)));
    P.a.candIdx = P.candIdx.p; P.a.candDist = P.candDist.p; P.a.candCap = std::max(total, 1);
    if (e == hipSuccess && (side ? n_right : n_left) > 0) chk(launch_proj_cands_fill(P.a, nullptr));
  }
  ProjFeArgs f{};
  f.offL = S[0].candOff.p; f.idxL = S[0].candIdx.p; f.distL = S[0].candDist.p;
  f.offR = S[1].candOff.p; f.idxR = S[1].candIdx.p; f.distR = S[1].candDist.p;
  f.nLeft = n_left; f.n = n; f.nmp = n_points; f.mode = mode; f.checkOri = check_ori; f.nnratio = nnratio;
  f.mps = S[0].mpp; f.pts = S[0].ppp; f.kps = k.p; f.l2r = a12.p; f.r2l = a21.p;
  f.occupied = occ.p; f.match = mt.p; f.result = res.p;
  int result[2] = {0, 0};
  // parallel fixed-point rounds (k_proj_round_fe); the serial walk is the fallback (writer-list overflow, no convergence
  // within 48 rounds) and the ORBX_PROJ_SERIAL=1 cross-check
  ScratchBuf<int4> wr0, wr1;
  ScratchBuf<int> wl0, wl1, wc0, wc1, fl;
  static const bool forceSerial = getenv("ORBX_PROJ_SERIAL") && atoi(getenv("ORBX_PROJ_SERIAL")) != 0;
  bool done = false;
  int lastRound = 0;
  if (!forceSerial && n_points > 0) {
    chk(wr0.alloc(nm)); chk(wr1.alloc(nm)); chk(wl0.alloc((size_t)n * kFeWriters)); chk(wl1.alloc((size_t)n * kFeWriters));
    chk(wc0.alloc(n)); chk(wc1.alloc(n)); chk(fl.alloc(40));
    f.writes[0] = wr0.p; f.writes[1] = wr1.p; f.writers[0] = wl0.p; f.writers[1] = wl1.p;
    f.nwriters[0] = wc0.p; f.nwriters[1] = wc1.p; f.flags = fl.p;
    for (int r = 0; r < 48 && e == hipSuccess && !done; r += 4) {
      chk(launch_proj_rounds_fisheye(f, r, 4, nullptr));
      int st[2] = {1, 0};
      if (e == hipSuccess) chk(hipMemcpy(st, fl.p, sizeof(st), hipMemcpyDeviceToHost));  // synchronises
      if (st[1]) break;  // a slot collected more than kFeWriters writers in one round
      done = st[0] == 0;
      lastRound = r + 3;
    }
  }
  if (e == hipSuccess) chk(done ? launch_proj_finish_fisheye(f, lastRound, nullptr) : launch_proj_resolve_fisheye(f, nullptr));
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOcc, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      std::memcpy(occupied, h, n);
      std::memcpy(result, h + (oRes - oOcc), sizeof(int));
      std::memcpy(match, h + (oMt - oOcc), (size_t)n * sizeof(int));
    }
  }
  wr0.free(); wr1.free(); wl0.free(); wl1.free(); wc0.free(); wc1.free(); fl.free();
  pk.release();
  S[0].release(); S[1].release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return result[0];
}
}  // namespace

int orbx_search_by_projection_fisheye(int device, const orbx_keypoint* kps, const uint8_t* desc, int n_left, int n_right,
                                      float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                                      int nlevels, const orbx_map_point_view* map_points,
                                      const orbx_map_point_right* map_points_right, int n_map_points, float th,
                                      int far_points, float th_far_points, float nnratio, const int32_t* left_to_right,
                                      const int32_t* right_to_left, uint8_t* occupied, int32_t* match) {
  const int n = n_left + n_right;
  if (n_left < 0 || n_right < 0 || n_map_points < 0 || nlevels < 1 || !scale_factors ||
      (n && (!kps || !desc || !occupied || !match)) || (n_map_points && (!map_points || !map_points_right)) ||
      (n_left && !left_to_right) || (n_right && !right_to_left))
    return fail(ORBX_E_BADARG, "bad argument");
  for (int i = 0; i < n_left; i++)
    if (left_to_right[i] < -1 || left_to_right[i] >= n_right) return fail(ORBX_E_BADARG, "left_to_right entry out of range");
  for (int i = 0; i < n_right; i++)
    if (right_to_left[i] < -1 || right_to_left[i] >= n_left) return fail(ORBX_E_BADARG, "right_to_left entry out of range");
  // the right camera as a second list of views: (mTrackProjXR, mTrackProjYR), mTrackViewCosR, mnTrackScaleLevelR
  std::vector<orbx_map_point_view> left(map_points, map_points + n_map_points), right(map_points, map_points + n_map_points);
  for (int i = 0; i < n_map_points; i++) {
    const orbx_map_point_right& r = map_points_right[i];
    if ((left[i].in_view && (left[i].predicted_level < 0 || left[i].predicted_level >= nlevels)) ||
        (r.in_view_r && (r.predicted_level_r < -1 || r.predicted_level_r >= nlevels)))
      return fail(ORBX_E_BADARG, "map point with a predicted level outside [0, nlevels)");
    if (!left[i].in_view) left[i].predicted_level = 0;
    right[i].proj_x = map_points[i].proj_xr;
    right[i].proj_y = r.proj_yr;
    right[i].view_cos = r.view_cos_r;
    right[i].predicted_level = r.predicted_level_r < 0 ? 0 : r.predicted_level_r;
    right[i].in_view = (r.in_view_r && r.predicted_level_r != -1) ? 1 : 0;  // :141-143
    // `if (!mbTrackInView && !mbTrackInViewR) continue` (:54) is implied: both lists stay empty
  }
  static const orbx_map_point_view dummy{};
  return search_by_projection_fisheye_impl(device, kps, desc, n_left, n_right, min_x, min_y, max_x, max_y, scale_factors, nlevels,
                                           n_map_points ? left.data() : &dummy, n_map_points ? right.data() : &dummy, nullptr,
                                           nullptr, n_map_points, th, far_points, th_far_points, nnratio, 0, left_to_right,
                                           right_to_left, occupied, match);
}

int orbx_search_by_projection_frame_fisheye(int device, const orbx_keypoint* kps, const uint8_t* desc, int n_left,
                                            int n_right, float min_x, float min_y, float max_x, float max_y,
                                            const orbx_projected_point* points, const float* uv_right, int n_points,
                                            int check_orientation, uint8_t* occupied, int32_t* match) {
  const int n = n_left + n_right;
  if (n_left < 0 || n_right < 0 || n_points < 0 || (n && (!kps || !desc || !occupied || !match)) ||
      (n_points && (!points || !uv_right)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n_points > 15000) return fail(ORBX_E_CAPACITY, "more than 15000 projected points");
  std::vector<orbx_projected_point> right(points, points + n_points);
  for (int i = 0; i < n_points; i++) {
    right[i].u = uv_right[2 * i];
    right[i].v = uv_right[2 * i + 1];
  }
  static const orbx_projected_point dummy{};
  return search_by_projection_fisheye_impl(device, kps, desc, n_left, n_right, min_x, min_y, max_x, max_y, nullptr, 0, nullptr,
                                           nullptr, n_points ? points : &dummy, n_points ? right.data() : &dummy, n_points, 1.0f,
                                           0, 0.f, 0.f, check_orientation, nullptr, nullptr, occupied, match);
}

int orbx_profile_enable(orbx_extractor* ex, int on) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  // on: 0 = off, 1 = every kernel launch, 2 + s = only launches of stage s (ORBX_STAGE_*)
  ex->profiling = on != 0;
  ex->profStage = on >= 2 ? on - 2 : -1;
  ex->lastEvValid = false;
  return ORBX_OK;
}

int orbx_profile_collect(orbx_extractor* ex, double* ms, int32_t* launches) {
  if (!ex || !ms || !launches) return fail(ORBX_E_BADARG, "null argument");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  for (int i = 0; i < ORBX_NUM_STAGES; i++) {
    ms[i] = 0;
    launches[i] = 0;
  }
  for (const auto& r : ex->evLog) {
    float t = 0;
    HIPC(hipEventElapsedTime(&t, ex->evPool[r.e0], ex->evPool[r.e1]));
    ms[r.stage] += t;
    launches[r.stage]++;
  }
  ex->evLog.clear();
  ex->evCursor = 0;
  ex->lastEvValid = false;
  return ORBX_OK;
}

const char* orbx_stage_name(int stage) {
  static const char* names[ORBX_NUM_STAGES] = {"k_resize", "k_detect", "k_octree", "k_blur",
                                                "k_slots",  "k_describe", "k_stereo_match", "k_stereo_filter"};
  return stage >= 0 && stage < ORBX_NUM_STAGES ? names[stage] : "?";
}

int orbx_level_stats(orbx_extractor* ex, int image, int32_t* w, int32_t* h, int32_t* n_candidates,
                     int32_t* n_selected) {
  if (!ex || ex->curW == 0 || image < 0 || image >= ex->lastN) return fail(ORBX_E_BADARG, "bad argument");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  const int L = ex->g.nlevels;
  for (int l = 0; l < L; l++) {
    if (w) w[l] = ex->g.lv[l].w;
    if (h) h[l] = ex->g.lv[l].h;
  }
  if (n_candidates)
    HIPC(hipMemcpy(n_candidates, ex->d_candCount.p + image * L, L * sizeof(int), hipMemcpyDeviceToHost));
  if (n_selected)
    HIPC(hipMemcpy(n_selected, ex->d_selCount.p + image * L, L * sizeof(int), hipMemcpyDeviceToHost));
  return ORBX_OK;
}

// Test hook: the device quadtree's introsort replica, run on the host (compared with std::sort in tests).
void orbx_debug_introsort(uint64_t* v, int n) { debug_introsort_host(v, n); }
void orbx_debug_set_detect_list_cap(int cap) { debug_set_detect_list_cap(cap); }
void orbx_debug_set_octree_global(int on) { debug_set_octree_global(on); }
int orbx_debug_introsort_device(int device, uint64_t* v, int n) {
  if (!v || n < 0 || n > 4000) return fail(ORBX_E_BADARG, "bad argument");
  if (n == 0) return ORBX_OK;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<uint64_t> d;
  HIPC(d.alloc(n));
  hipError_t e = hipMemcpy(d.p, v, (size_t)n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_debug_sort(d.p, n, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(v, d.p, (size_t)n * 8, hipMemcpyDeviceToHost);
  d.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

}  // extern "C"
