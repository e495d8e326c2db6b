// orbx_api.hip — C ABI of liborbx (include/orbx.h): handles, buffers, geometry, launch sequencing.
// Host-side restatement of the ORBextractor constructor tables (src/ORBextractor.cc:408-469) and of the
// OpenCV resize coefficient tables; all pixel/bit work is in orbx_kernels.hip.
#include "orbx_host.h"

namespace orbx_host {
thread_local std::string g_err;
}

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

namespace orbx_host {

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
  for (int l = 0; l < ORBX_MAX_LEVELS; l++) g.levelCell[l] = INT_MAX;
  g.nlevels = L;
  g.iniTh = ex->prm.ini_th_fast;
  g.minTh = ex->prm.min_th_fast;
  g.cv440 = ex->cv440;
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
    g.levelCell[l] = cells;
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
    xc += align_up(v.w, 256);  // k_resize reads whole 256-column blocks of the x table
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
  g.tileP = 4 * ((maxCW + 3) / 4 + 2);   // quads per row + 2 dwords (the 6 px halo = the 2 dwords a lane reads past its quad)
  g.tileH = maxCH + 6;
  g.scoreP = 4 * ((maxCW + 3) / 4 + 2);  // 1 dword zero pad left + quads + 1 dword zero pad right
  g.scoreH = maxCH + 2;
  g.listCap = align_up((long long)maxCW * maxCH, 8);
  g.selImg = sel;
  g.outCap = sel;
  g.pyrImg = off;
  g.candImg = candOff;
  g.cellImg = cellOff;
  if (resize_lds_bytes(g) > 160 * 1024 - 2048) {
    why = "scale factor too large for the LDS-staged resize footprint";
    return ORBX_E_UNSUPPORTED;
  }
  if (octree_lds_bytes(g) > 160 * 1024 - 2048) {
    why = "nfeatures too large for the LDS-resident quadtree";
    return ORBX_E_UNSUPPORTED;
  }
  return ORBX_OK;
}

// ---- resize coefficient tables, cv::resize INTER_LINEAR 8U -------------------------------
// x table: one uint4 per dst column {sx, sx & ~3, v_perm selector of the byte pair, a0 | a1 << 16}; every level is padded
// to a multiple of 256 entries with copies of its last column.
void build_coefs(const Geom& g, std::vector<uint4>& xtab, std::vector<int>& yofs, std::vector<short>& yab) {
  int nx = 0, ny = 0;
  for (int l = 0; l < g.nlevels; l++) {
    nx += align_up(g.lv[l].w, 256);
    ny += g.lv[l].h;
  }
  xtab.assign(nx, make_uint4(0, 0, 0, 0));
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
      const uint32_t a0 = (uint16_t)sat_short((1.f - fx) * 2048.f), a1 = (uint16_t)sat_short(fx * 2048.f);
      const uint32_t sh = (uint32_t)sx & 3u;
      xtab[D.xcoef + dx] = make_uint4((uint32_t)sx, (uint32_t)sx & ~3u, sh | 0x0c000c00u | ((sh + 1u) << 16), a0 | (a1 << 16));
    }
    for (int dx = D.w; dx < align_up(D.w, 256); dx++) xtab[D.xcoef + dx] = xtab[D.xcoef + D.w - 1];
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


// ---- fused small levels of the resize chain (orbx_resize_tail.hip) ----------------------------------
// The last levels of the pyramid -- at most kTailLevels of them, and only levels whose SOURCE has at most kTailSrcPx pixels
// (levels 5-7 of a 1280x720 pyramid: level 4 has 214 k pixels; levels 4-7 of a 640x480 one) -- are resized in ONE fused
// launch; a workgroup owns a band of kTailRows rows of the last level.  A segment shrinks (and a level falls back to
// k_resize) when the cascade's LDS tiles do not fit (large scale factors).  The hook / ORBX_RESIZE_TAIL can ask for any
// chain of segments.  Measured, step time of bench.py with three handles: 1280x720 x 64 images: no fusion 0.518 ms; levels
// 5-7 with bands of 13 / 20 / 26 / 34 rows 0.505 / 0.503 / 0.499 / 0.500 ms; levels 4-7 0.509 ms at 13 rows, slower above;
// levels 6-7 0.514 ms.  640x480 x 64: no fusion 0.2434 ms; levels 2-4 + 5-7 0.2484; 4-7 0.2400; 5-7 0.2415; 2-7 0.2453.
constexpr int kTailSrcPx = 230 * 1000, kTailLevels = 4, kTailRows = 26, kTailLdsMax = 96 * 1024;
static int g_tail_first = -1, g_tail_levels = kTailLevels, g_tail_rows = kTailRows;  // test hook (orbx_debug_set_resize_tail)

// One segment: levels lA .. lA+nT-1 from level lA-1.  Returns false when it cannot be built (LDS, quad limit).
static bool build_tail_segment(const Geom& g, const std::vector<int>& yofs, int lA, int nT, int R, TailPlan& tp,
                               std::vector<TailBand>& nd, int minLA = 2, int minT = 2, size_t ldsMax = kTailLdsMax) {
  std::memset(&tp, 0, sizeof(tp));
  nd.clear();
  if (lA < minLA || nT < minT || lA + nT > g.nlevels || nT > ORBX_MAX_LEVELS) return false;
  const int hTop = g.lv[lA + nT - 1].h;
  const int nB = (hTop + R - 1) / R;
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  // nd[b][t + 1] = rows of level lA + t band b computes, nd[b][0] = rows of level lA - 1 it stages
  nd.assign((size_t)nB * (nT + 1), TailBand{0, 0, 0, 0});
  for (int b = 0; b < nB; b++) nd[(size_t)b * (nT + 1) + nT] = {b * R, std::min((b + 1) * R, hTop) - 1, 0, 0};
  for (int t = nT - 1; t >= 0; t--) {  // rows of cascade position t (level lA + t - 1) from those of position t + 1
    const LevelDev &D = g.lv[lA + t], &S = g.lv[lA + t - 1];
    for (int b = 0; b < nB; b++) {
      const TailBand& d = nd[(size_t)b * (nT + 1) + t + 1];
      TailBand& s = nd[(size_t)b * (nT + 1) + t];
      s.first = clampi(yofs[D.ycoef + d.first], 0, S.h - 1);
      s.last = clampi(yofs[D.ycoef + d.last] + 1, 0, S.h - 1);
      if (t > 0) {  // an output level: every row must be produced by somebody
        if (b == 0) s.first = 0;
        if (b == nB - 1) s.last = S.h - 1;
      }
    }
    if (t > 0)
      for (int b = 0; b + 1 < nB; b++) {  // no gaps between consecutive bands (scale factors > 2 skip source rows)
        TailBand& s = nd[(size_t)b * (nT + 1) + t];
        s.last = std::max(s.last, nd[(size_t)(b + 1) * (nT + 1) + t].first - 1);
      }
  }
  int maxRows[ORBX_MAX_LEVELS + 1] = {0};
  for (int b = 0; b < nB; b++)
    for (int t = 0; t <= nT; t++) {
      TailBand& e = nd[(size_t)b * (nT + 1) + t];
      e.ownEnd = t == 0 ? 0 : (b + 1 < nB ? nd[(size_t)(b + 1) * (nT + 1) + t].first : g.lv[lA + t - 1].h);
      maxRows[t] = std::max(maxRows[t], e.last - e.first + 1);
    }
  size_t tile[2] = {0, 0};
  int rtab = 0;
  for (int t = 0; t <= nT; t++) {
    const LevelDev& V = g.lv[lA + t - 1];
    tp.pitch[t] = align_up(V.w + 8, 16);
    if (t < nT) tile[t & 1] = std::max(tile[t & 1], (size_t)maxRows[t] * tp.pitch[t]);  // the last level is not kept
    if (t > 0) {
      if ((V.w + 3) / 4 > 512) return false;  // a thread of k_resize_tail owns one quad of columns
      tp.rtOff[t - 1] = rtab;
      rtab += maxRows[t];
    }
    if ((long long)maxRows[t] * (V.w + 16) >= (1 << 20)) return false;  // (the kernel's float index split)
  }
  const size_t lds = (size_t)align_up(tile[0], 16) + align_up(tile[1], 16) + (size_t)rtab * 16;
  if (lds > ldsMax) return false;
  tp.rtTotal = rtab;
  tp.lA = lA;
  tp.nT = nT;
  tp.nBands = nB;
  tp.tileBytes[0] = align_up(tile[0], 16);
  tp.tileBytes[1] = align_up(tile[1], 16);
  tp.ldsBytes = (unsigned)lds;
  return true;
}

// All segments of the current geometry, in level order; tp.bandOff = first entry of the segment in `bands`.
void build_tail_plans(const Geom& g, const std::vector<int>& yofs, std::vector<TailPlan>& plans, std::vector<TailBand>& bands) {
  plans.clear();
  bands.clear();
  static const bool envRead = [] {  // diagnostic knob, same meaning as the hook: ORBX_RESIZE_TAIL=first[,levels[,rows]]
    if (const char* e = getenv("ORBX_RESIZE_TAIL")) {
      int a = -1, b = 0, c = 0;
      if (sscanf(e, "%d,%d,%d", &a, &b, &c) >= 1) {
        g_tail_first = a;
        if (b > 0) g_tail_levels = b;
        if (c > 0) g_tail_rows = c;
      }
    }
    return true;
  }();
  (void)envRead;
  if (g_tail_first == 0) return;  // (fusion off)
  const int L = g.nlevels;
  int l = 2;
  if (g_tail_first > 0) {
    l = std::max(2, g_tail_first);
  } else {  // the library's policy: one segment, the last (at most kTailLevels) small levels
    while (l < L && (long long)g.lv[l - 1].w * g.lv[l - 1].h > kTailSrcPx) l++;
    l = std::max(l, L - kTailLevels);
  }
  while (L - l >= 2) {
    TailPlan tp;
    std::vector<TailBand> nd;
    int nT = std::min(g_tail_levels, L - l);
    if (L - l - nT == 1) nT = (L - l >= 4 && g_tail_levels >= 2) ? nT - 1 : nT + 1;  // never leave a single level behind
    while (nT >= 2 && !build_tail_segment(g, yofs, l, nT, g_tail_rows, tp, nd)) nT--;
    if (nT < 2) {  // this level through k_resize, try again from the next one
      l++;
      continue;
    }
    tp.bandOff = (int)bands.size();
    bands.insert(bands.end(), nd.begin(), nd.end());
    plans.push_back(tp);
    l += nT;
  }
}

// The single-frame plans (orbx_extract / orbx_extract_stereo, and device batches of <= kLatMaxImages images): the WHOLE chain
// from level 0 as one or two cascade launches.  Five dependent launches cost 37 us of a 1280x720 stereo frame's ~210 us of GPU
// time for ~3 us of arithmetic (profiles/r5a_frame_trace.txt); with a couple of images the chip is empty, so the halo rows a
// deep cascade recomputes (a band of 2 rows of level 7 needs ~38 rows of level 0) cost nothing that matters.  Segments are
// as long as the LDS admits (kLatLdsMax); ORBX_LAT_TAIL=levels,rows overrides (levels = 0 switches the plans off).
constexpr int kLatLdsMax = 156 * 1024, kLatMaxImages = 2;
static int g_lat_max_images = kLatMaxImages;   // ORBX_LAT_MAX_IMAGES (measurement aid)
static int g_lat_levels = 4, g_lat_rows = 2;  // 1280x720: levels 1-4 + 5-7, 14.3 + 7.6 us (one launch for 1-7: 24; 2 + 2 + 2 + 1: 27)
void build_latency_plans(const Geom& g, const std::vector<int>& yofs, std::vector<TailPlan>& plans, std::vector<TailBand>& bands) {
  plans.clear();
  bands.clear();
  static const bool envRead = [] {
    if (const char* e = getenv("ORBX_LAT_TAIL")) {
      int a = -1, b = 0;
      if (sscanf(e, "%d,%d", &a, &b) >= 1) {
        g_lat_levels = a;
        if (b > 0) g_lat_rows = b;
      }
    }
    if (const char* e = getenv("ORBX_LAT_MAX_IMAGES")) g_lat_max_images = atoi(e);
    return true;
  }();
  (void)envRead;
  if (g_lat_levels <= 0) return;
  const int L = g.nlevels;
  int l = 1;
  while (l < L) {
    TailPlan tp;
    std::vector<TailBand> nd;
    int nT = std::min(g_lat_levels, L - l);
    while (nT >= 1 && !build_tail_segment(g, yofs, l, nT, g_lat_rows, tp, nd, 1, 1, (size_t)kLatLdsMax)) nT--;
    if (nT < 1) {  // (a level the cascade kernel cannot take: no single-frame plan, the level kernels serve every batch size)
      plans.clear();
      bands.clear();
      return;
    }
    tp.bandOff = (int)bands.size();
    bands.insert(bands.end(), nd.begin(), nd.end());
    plans.push_back(tp);
    l += nT;
  }
}

int configure(orbx_extractor* ex, int w, int h) {
  if (w == ex->curW && h == ex->curH) return ORBX_OK;
  // every per-axis table (resize coefficients, row tables) is sized for max_width x max_height: a wide-and-short image
  // can pass the aggregate buffer checks below and still overrun them
  if (w > ex->maxW || h > ex->maxH)
    return fail(ORBX_E_CAPACITY, "image larger than the handle's max_width x max_height");
  Geom g;
  std::string why;
  int rc = build_geom(ex, w, h, g, why);
  if (rc != ORBX_OK) return fail(rc, why);
  const Geom& m = ex->gmax;
  if (g.pyrImg > m.pyrImg || g.candImg > m.candImg || g.selImg > m.selImg || g.cellImg > m.cellImg ||
      g.totalCells > m.totalCells)
    return fail(ORBX_E_CAPACITY, "image within max_width x max_height but its pyramid / cell-grid rounding needs more "
                                 "buffer than the handle's maximum size: create the handle with this size");
  g.outCap = m.outCap;  // keep result strides fixed for the life of the handle
  g.selImg = m.selImg;
  g.pyrImg = m.pyrImg;
  g.candImg = m.candImg;
  g.cellImg = m.cellImg;
  std::vector<uint4> xtab;
  std::vector<int> yofs;
  std::vector<short> yab;
  build_coefs(g, xtab, yofs, yab);
  HIPC(hipStreamSynchronize(ex->stream));
  HIPC(hipMemcpy(ex->d_xtab.p, xtab.data(), xtab.size() * sizeof(uint4), hipMemcpyHostToDevice));
  HIPC(hipMemcpy(ex->d_yofs.p, yofs.data(), yofs.size() * sizeof(int), hipMemcpyHostToDevice));
  {  // k_resize's row table: the two source rows of every destination row, clamped to the source level, as u16 halves
    std::vector<uint32_t> yrow(yofs.size(), 0);
    for (int l = 1; l < g.nlevels; l++) {
      const int Sh = g.lv[l - 1].h;
      for (int dy = 0; dy < g.lv[l].h; dy++) {
        const int sy = yofs[g.lv[l].ycoef + dy];
        const int a = std::min(std::max(sy, 0), Sh - 1), b = std::min(std::max(sy + 1, 0), Sh - 1);
        yrow[g.lv[l].ycoef + dy] = (uint32_t)a | ((uint32_t)b << 16);
      }
    }
    HIPC(hipMemcpy(ex->d_yrow.p, yrow.data(), yrow.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  HIPC(hipMemcpy(ex->d_yab.p, yab.data(), yab.size() * sizeof(short), hipMemcpyHostToDevice));
  HIPC(prepare_kernels(g));
  std::vector<TailPlan> plans;
  std::vector<TailBand> bands;
  build_tail_plans(g, yofs, plans, bands);
  if (!plans.empty()) {
    if (ex->d_tailBands.n < bands.size()) HIPC(ex->d_tailBands.alloc(bands.size()));
    HIPC(hipMemcpy(ex->d_tailBands.p, bands.data(), bands.size() * sizeof(TailBand), hipMemcpyHostToDevice));
    unsigned lds = 0;
    for (const TailPlan& tp : plans) lds = std::max(lds, tp.ldsBytes);
    HIPC(prepare_resize_tail(lds));
  }
  ex->tails = plans;
  {
    std::vector<TailPlan> lp;
    std::vector<TailBand> lb;
    build_latency_plans(g, yofs, lp, lb);
    if (!lp.empty()) {
      if (ex->d_latBands.n < lb.size()) HIPC(ex->d_latBands.alloc(lb.size()));
      HIPC(hipMemcpy(ex->d_latBands.p, lb.data(), lb.size() * sizeof(TailBand), hipMemcpyHostToDevice));
      unsigned lds = 0;
      for (const TailPlan& tp : lp) lds = std::max(lds, tp.ldsBytes);
      HIPC(prepare_resize_tail(lds));
    }
    ex->latTails = lp;
  }
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

int record_pipeline(orbx_extractor* ex, int n, bool lapTrivial, bool capturing, bool pyrDone = false);
static void drop_graphs(orbx_extractor* ex) {
  for (auto& e : ex->graphExec) {
    if (e) (void)hipGraphExecDestroy(e);
    e = nullptr;
  }
}

// May the images at `d_images` go through the single-frame cascade plans?  k_resize_tail stages level 0 with 16-byte loads of
// whole rows: 16-byte aligned rows, and the last row's last load must stay inside the buffer (the handle's own staging buffer
// is padded; a caller's buffer is only trusted when the width is a multiple of 16).
static bool latency_plan_ok(const orbx_extractor* ex, const uint8_t* d_images, int n, int w, ptrdiff_t row_pitch,
                            ptrdiff_t image_pitch) {
  if (ex->latTails.empty() || n > g_lat_max_images) return false;
  if (((uintptr_t)d_images & 15) || (row_pitch & 15) || (image_pitch & 15)) return false;
  return (w & 15) == 0 || d_images == ex->d_stage.p;
}

// Everything of an extraction that precedes its kernels: geometry, per-extraction state, lapping areas.
static int prepare_extract(orbx_extractor* ex, const uint8_t* d_images, int n, int w, int h, ptrdiff_t row_pitch,
                           ptrdiff_t image_pitch, const int32_t* lap, bool& lapTrivial) {
  int rc = configure(ex, w, h);
  if (rc != ORBX_OK) return rc;
  ex->pyr.l0 = d_images;
  ex->pyr.l0Row = row_pitch;
  ex->pyr.l0Img = image_pitch;
  ex->pyr.pyr = ex->d_pyr.p;
  ex->lastN = n;
  ex->lastEvValid = false;
  ex->blurValid = false;       // per-extraction state is reset HERE: a hipGraph replay never runs record_pipeline
  ex->lastStereoPairs = 0;     // results of an earlier batch's stereo association are not this batch's
  ex->hostPyrImages = 0;       // (the host copy of the pyramid, orbx_set_host_pyramid, belongs to the previous extraction)
  ex->useLat = latency_plan_ok(ex, d_images, n, w, row_pitch, image_pitch);
  hipStream_t s = ex->stream;
  // Keypoint x is >= 19 at every level (16-px border + the 3-px FAST ring, scaled by >= 1), so a lapping area that
  // ends below 19 -- the rectified-stereo {0, 0} in particular -- can hold no keypoint: the output order is then
  // just level-major list order, k_slots is skipped and k_describe derives its slot from the level counts.
  lapTrivial = true;
  if (lap)
    for (int i = 0; i < n; i++) lapTrivial = lapTrivial && lap[2 * i + 1] < 19;
  if (!lapTrivial) {
    // The lapping areas of a rig do not change from frame to frame: they are uploaded only when they differ from what the
    // device holds, and then from page-locked memory (no per-step copy from the caller's pageable array).
    const size_t nb = (size_t)n * 2 * sizeof(int);
    if (ex->lapN != n || std::memcmp(ex->h_lap, lap, nb) != 0) {
      HIPC(hipStreamSynchronize(s));  // an earlier upload from h_lap may still be queued
      std::memcpy(ex->h_lap, lap, nb);
      ex->lapN = n;
      HIPC(hipMemcpyAsync(ex->d_lap.p, ex->h_lap, nb, hipMemcpyHostToDevice, s));
    }
  }
  return ORBX_OK;
}

// The resize chain of images [img0, img0 + n) on stream s: the single-frame cascade plans (ex->useLat) or, for batches, the
// level kernels + the fused small levels.  (The level kernels take whole batches only: img0 = 0.)
static int enqueue_pyramid(orbx_extractor* ex, hipStream_t s, int img0, int n) {
  const Geom& g = ex->g;
  if (ex->useLat) {
    for (const TailPlan& tp : ex->latTails) {
      StageTimer t(ex, s, ORBX_STAGE_RESIZE);
      HIPC(launch_resize_tail(g, ex->pyr, tp, ex->d_latBands.p + tp.bandOff, img0, n, ex->d_xtab.p, ex->d_yofs.p, ex->d_yab.p, s));
    }
    return ORBX_OK;
  }
  size_t seg = 0;  // next fused segment of small levels (ex->tails, in level order)
  for (int l = 1; l < g.nlevels;) {
    StageTimer t(ex, s, ORBX_STAGE_RESIZE);
    if (seg < ex->tails.size() && ex->tails[seg].lA == l) {
      const TailPlan& tp = ex->tails[seg++];
      HIPC(launch_resize_tail(g, ex->pyr, tp, ex->d_tailBands.p + tp.bandOff, 0, n, ex->d_xtab.p, ex->d_yofs.p, ex->d_yab.p, s));
      l += tp.nT;
    } else {
      HIPC(launch_resize(g, ex->pyr, n, l, ex->d_xtab.p, ex->d_yrow.p, ex->d_yab.p, s));
      l++;
    }
  }
  return ORBX_OK;
}

int enqueue_extract(orbx_extractor* ex, const uint8_t* d_images, int n, int w, int h, ptrdiff_t row_pitch,
                    ptrdiff_t image_pitch, const int32_t* lap) {
  bool lapTrivial = true;
  int rc = prepare_extract(ex, d_images, n, w, h, row_pitch, image_pitch, lap, lapTrivial);
  if (rc != ORBX_OK) return rc;
  hipStream_t s = ex->stream;
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

// The kernel launches of one extraction on the handle's stream (also the body captured into the hipGraph):
// resize chain -> k_detect -> k_octree -> (k_slots) -> k_describe.  The 7x7 Gaussian of :1074-1076 is evaluated inside
// k_describe on the keypoints' own windows (no blurred pyramid, no side stream); the round-1 schedule (k_blur over every
// level on a side stream beside the quadtree) and the alternatives measured around it are in DESIGN.md 4.
int record_pipeline(orbx_extractor* ex, int n, bool lapTrivial, bool capturing, bool pyrDone) {
  const Geom& g = ex->g;
  hipStream_t s = ex->stream;
  ex->blurValid = false;  // the blurred levels (orbx_pyramid_level blurred = 1) are produced on demand
  if (!pyrDone) {         // (orbx_extract_stereo builds the two eyes' pyramids itself, each behind its own upload)
    const int rc = enqueue_pyramid(ex, s, 0, n);
    if (rc != ORBX_OK) return rc;
  }
  if (ex->d_dbgScore.p) HIPC(hipMemsetAsync(ex->d_dbgScore.p, 0, ex->d_dbgScore.n, s));  // test tap only
  // k_detect fills every VALU of the chip by itself: two of them side by side (two handles in flight) only stretch
  // each other.  A per-device token orders the k_detect launches of all handles one after the other, while each still
  // overlaps the other handles' quadtree / describe / stereo / resize work.
  static const bool noToken = getenv("ORBX_NO_DETECT_TOKEN") && atoi(getenv("ORBX_NO_DETECT_TOKEN")) != 0;   // measurement aid
  // (a graph cannot wait on it; a launch of one or two images does not fill the chip: no ordering, no event record between
  // k_detect and k_octree of a single frame)
  DetectToken* tok = (!capturing && !noToken && n > g_lat_max_images) ? detect_token(ex->device) : nullptr;
  if (tok) {
    std::lock_guard<std::mutex> lk(tok->mu);
    if (tok->valid && tok->last != ex) HIPC(hipStreamWaitEvent(s, tok->ev, 0));
    ex->lastEvValid = false;
    {
      StageTimer t(ex, s, ORBX_STAGE_DETECT);
      HIPC(launch_detect(g, ex->pyr, n, ex->d_cellCand.p, ex->d_cellCount.p, 0, g.nlevels, ex->d_dbgScore.p, s));
    }
    HIPC(hipEventRecord(tok->ev, s));
    tok->valid = true;
    tok->last = ex;
  } else {
    StageTimer t(ex, s, ORBX_STAGE_DETECT);
    HIPC(launch_detect(g, ex->pyr, n, ex->d_cellCand.p, ex->d_cellCount.p, 0, g.nlevels, ex->d_dbgScore.p, s));
  }
  {
    StageTimer t(ex, s, ORBX_STAGE_OCTREE);
    HIPC(launch_octree(g, n, ex->d_cellCand.p, ex->d_cellCount.p, ex->d_cellPrefix.p, ex->d_cand.p,
                       ex->d_candCount.p, ex->d_knode.p, ex->d_sel.p, ex->d_selCount.p, 0, g.nlevels, s));
  }
  if (!lapTrivial) {
    StageTimer t(ex, s, ORBX_STAGE_SLOTS);
    HIPC(launch_slots(g, n, ex->d_sel.p, ex->d_selCount.p, ex->d_lap.p, ex->d_slot.p, ex->d_nOut.p, ex->d_mono.p, s));
  }
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

}  // namespace orbx_host

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
    nx += align_up(m.lv[l].w, 256);
    ny += m.lv[l].h;
  }
  hipError_t e = hipSuccess;
  auto ok = [&](hipError_t r) {
    if (e == hipSuccess) e = r;
  };
  ok(hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking));
  ok(hipEventCreateWithFlags(&ex->done, hipEventDisableTiming));
  ok(ex->d_pyr.alloc(B * m.pyrImg + 256));
  ok(ex->d_stage.alloc(B * (size_t)ex->stagePitch * max_height + 256));
  ok(ex->d_cand.alloc(B * m.candImg));
  ok(ex->d_cellCand.alloc(B * m.cellImg));
  ok(ex->d_cellCount.alloc(B * m.totalCells));
  ok(ex->d_cellPrefix.alloc(B * m.totalCells));
  ok(ex->d_knode.alloc(B * m.candImg));
  ok(ex->d_candCount.alloc(B * m.nlevels));
  ok(ex->d_sel.alloc(B * m.selImg));
  ok(ex->d_selCount.alloc(B * m.nlevels + ORBX_MAX_LEVELS));   // (k_describe2 reads ORBX_MAX_LEVELS counts of an image unconditionally)
  ok(ex->d_slot.alloc(B * m.selImg));
  ok(ex->d_kps.alloc(B * m.outCap));
  ok(ex->d_desc.alloc(B * m.outCap * 32));
  ok(ex->d_nOut.alloc(B));
  ok(ex->d_mono.alloc(B));
  ok(ex->d_lap.alloc(B * 2));
  ok(hipHostMalloc(reinterpret_cast<void**>(&ex->h_lap), (size_t)B * 2 * sizeof(int), hipHostMallocDefault));
  ok(ex->d_xtab.alloc(nx + 64));
  ok(ex->d_yofs.alloc(ny + 64));
  ok(ex->d_yrow.alloc(ny + 64));
  ok(ex->d_packCtr.alloc(4));
  if (e == hipSuccess) e = hipMemset(ex->d_packCtr.p, 0, 4 * sizeof(int));
  ok(ex->d_yab.alloc(2 * ny + 64));
  ok(hipHostMalloc(reinterpret_cast<void**>(&ex->hostResults), host_results_bytes(m.outCap), hipHostMallocDefault));
  if (e != hipSuccess) {
    std::string msg = std::string("allocation failed: ") + hipGetErrorString(e);
    orbx_extractor_destroy(ex);
    return fail(ORBX_E_HIP, msg);
  }
  std::memset(ex->hostResults, 0, 64);   // counts and the gather's sequence word
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
  if (ex->hostPyr) (void)hipHostFree(ex->hostPyr);
  ex->hostPyr = nullptr;
  if (ex->streamPyr) (void)hipStreamSynchronize(ex->streamPyr);
  if (ex->hostPyrAll) (void)hipHostFree(ex->hostPyrAll);
  ex->hostPyrAll = nullptr;
  if (ex->evPyr) (void)hipEventDestroy(ex->evPyr);
  if (ex->streamPyr) (void)hipStreamDestroy(ex->streamPyr);
  if (ex->h_lap) (void)hipHostFree(ex->h_lap);
  ex->h_lap = nullptr;
  ex->d_dbgScore.free(); ex->d_pyr.free(); ex->d_blur.free(); ex->d_stage.free(); ex->d_desc.free(); ex->d_cand.free(); ex->d_cellCand.free(); ex->d_cellCount.free(); ex->d_cellPrefix.free();
  ex->d_sel.free(); ex->d_knode.free(); ex->d_candCount.free(); ex->d_selCount.free(); ex->d_slot.free();
  ex->d_nOut.free(); ex->d_mono.free(); ex->d_lap.free(); ex->d_fl2r.free(); ex->d_fr2l.free(); ex->d_fcnt.free(); ex->d_fcand.free(); ex->d_bowWord.free(); ex->d_bowNode.free(); ex->d_bowStart.free();
  ex->d_bowCounts.free(); ex->d_bowWeight.free(); ex->d_bowValues.free(); ex->d_bowWords.free(); ex->d_bowNodes.free(); ex->d_bowFeats.free(); ex->d_fdepth.free(); ex->d_fp3d.free(); ex->d_xtab.free(); ex->d_tailBands.free(); ex->d_yofs.free(); ex->d_yrow.free(); ex->d_packCtr.free();
  ex->d_mapPos.free(); ex->d_mapNormal.free(); ex->d_mapMinD.free(); ex->d_mapMaxD.free(); ex->d_mapDesc.free(); ex->d_mapFlags.free();
  ex->d_mapSkip.free(); ex->d_poses.free(); ex->d_views.free();
  ex->d_lfPos.free(); ex->d_lfAngle.free(); ex->d_lfOct.free(); ex->d_lfN.free(); ex->d_lfDesc.free(); ex->d_lfFlags.free();
  ex->d_posesQ.free(); ex->d_pviews.free(); ex->d_scaleF.free(); ex->d_posesK.free(); ex->d_fviewsL.free(); ex->d_fviewsR.free();
  ex->d_latBands.free();
  ex->d_yab.free(); ex->d_kps.free(); ex->d_uR.free(); ex->d_depth.free(); ex->d_sad.free(); ex->d_rowStart.free(); ex->d_srec.free(); ex->d_sdesc.free();
  for (hipEvent_t e : ex->evPool) (void)hipEventDestroy(e);
  if (ex->done) (void)hipEventDestroy(ex->done);
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

int orbx_set_opencv_compat(orbx_extractor* ex, int opencv_version) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (opencv_version != 440 && opencv_version != 451 && opencv_version != 44016 && opencv_version != 44032)
    return fail(ORBX_E_BADARG, "opencv_version: 440 / 44016 / 44032 (OpenCV 4.0 .. 4.5.0: scalar model / 16- / 32-lane vector body) or "
                               "451 (OpenCV >= 4.5.1)");
  const int v = opencv_version == 440 ? 1 : opencv_version == 44016 ? 16 : opencv_version == 44032 ? 32 : 0;
  if (v == ex->cv440) return ORBX_OK;
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  HIPC(hipStreamSynchronize(ex->stream));  // queued work keeps the arithmetic it was enqueued with
  ex->cv440 = v;
  ex->g.cv440 = v;
  ex->gmax.cv440 = v;
  ex->blurValid = false;
  drop_graphs(ex);  // (a captured pipeline holds the other k_describe instantiation)
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

int orbx_stream_handle(const orbx_extractor* ex, void** stream) {
  if (!ex || !stream) return fail(ORBX_E_BADARG, "null argument");
  *stream = reinterpret_cast<void*>(ex->stream);
  return ORBX_OK;
}

int orbx_extract_batch(orbx_extractor* ex, const uint8_t* images, int n_images, int w, int h, ptrdiff_t row_pitch,
                       ptrdiff_t image_pitch, const int32_t* lap) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (!images || n_images <= 0 || w <= 0 || h <= 0) return fail(ORBX_E_EMPTY, "empty image");
  if (n_images > ex->maxB) return fail(ORBX_E_CAPACITY, "batch larger than max_batch");
  if (w > ex->maxW || h > ex->maxH) return fail(ORBX_E_CAPACITY, "image larger than the handle's max_width x max_height");
  if (row_pitch < w || image_pitch < row_pitch * (ptrdiff_t)(h - 1) + w) return fail(ORBX_E_BADARG, "pitch too small");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const int pitch = align_up(w, 64);
  const size_t imgBytes = (size_t)pitch * h;
  if (image_pitch == row_pitch * (ptrdiff_t)h) {  // frames back to back: one tall 2-D copy
    HIPC(hipMemcpy2DAsync(ex->d_stage.p, pitch, images, row_pitch, w, (size_t)h * n_images, hipMemcpyHostToDevice, ex->stream));
  } else {
    for (int i = 0; i < n_images; i++)
      HIPC(hipMemcpy2DAsync(ex->d_stage.p + i * imgBytes, pitch, images + (size_t)i * image_pitch, row_pitch, w, h,
                            hipMemcpyHostToDevice, ex->stream));
  }
  return enqueue_extract(ex, ex->d_stage.p, n_images, w, h, pitch, (ptrdiff_t)imgBytes, lap);
}

int orbx_batch_download_async(orbx_extractor* ex, int32_t* counts, int32_t* mono, orbx_keypoint* kps, uint8_t* desc,
                              float* uright, float* depth, int n_pairs) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (ex->lastN <= 0) return fail(ORBX_E_BADARG, "no batch has been extracted on this handle");
  if (n_pairs < 0 || n_pairs > ex->lastStereoPairs)
    return fail(ORBX_E_BADARG, "n_pairs exceeds the stereo association run on this handle since its last extraction");
  HIPC(hipSetDevice(ex->device));
  const size_t n = (size_t)ex->lastN, oc = (size_t)ex->gmax.outCap;
  hipStream_t st = ex->stream;
  if (counts) HIPC(hipMemcpyAsync(counts, ex->d_nOut.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
  if (mono) HIPC(hipMemcpyAsync(mono, ex->d_mono.p, n * sizeof(int), hipMemcpyDeviceToHost, st));
  if (kps) HIPC(hipMemcpyAsync(kps, ex->d_kps.p, n * oc * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, st));
  if (desc) HIPC(hipMemcpyAsync(desc, ex->d_desc.p, n * oc * 32, hipMemcpyDeviceToHost, st));
  if (uright && n_pairs) HIPC(hipMemcpyAsync(uright, ex->d_uR.p, (size_t)n_pairs * oc * sizeof(float), hipMemcpyDeviceToHost, st));
  if (depth && n_pairs) HIPC(hipMemcpyAsync(depth, ex->d_depth.p, (size_t)n_pairs * oc * sizeof(float), hipMemcpyDeviceToHost, st));
  return ORBX_OK;
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

// The single-frame host entries fetch their results with ONE kernel that writes the handle's pinned host block over PCIe
// (hipHostMalloc memory is device-visible and host-coherent): a single-frame trace showed the six D2H copies of the results
// taking 57 us of a 270 us frame, against ~10 us for the gather.
static int enqueue_stereo_match(orbx_extractor* left, int first_left, orbx_extractor* right, int first_right,
                                int n_pairs, float bf, float b, bool withFilter, StereoArgs* argsOut,
                                const ResultPack* packWithBand = nullptr, bool* packedOut = nullptr);
static hipError_t make_result_pack(orbx_extractor* ex, int nimg, bool stereo, ResultPack* out) {
  const size_t oc = (size_t)ex->gmax.outCap;
  uint8_t* hd = nullptr;
  hipError_t e = hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), ex->hostResults, 0);
  if (e != hipSuccess) return e;
  ResultPack a{};
  a.nOut = ex->d_nOut.p; a.mono = ex->d_mono.p;
  a.kps = reinterpret_cast<const uint32_t*>(ex->d_kps.p); a.desc = reinterpret_cast<const uint32_t*>(ex->d_desc.p);
  a.uR = reinterpret_cast<const uint32_t*>(ex->d_uR.p); a.depth = reinterpret_cast<const uint32_t*>(ex->d_depth.p);
  a.hCnt = reinterpret_cast<uint32_t*>(hd);
  a.hKps = reinterpret_cast<uint32_t*>(hd + hr_kps(oc)); a.hDesc = reinterpret_cast<uint32_t*>(hd + hr_desc(oc));
  a.hUr = reinterpret_cast<uint32_t*>(hd + hr_ur(oc)); a.hDepth = reinterpret_cast<uint32_t*>(hd + hr_depth(oc));
  a.nimg = nimg; a.cap = (int)oc; a.stereo = stereo ? 1 : 0;
  a.mask = 0x7F; a.fixedN = -1;
  a.packCtr = nullptr; a.hFlag = nullptr; a.seq = 0;
  *out = a;
  return hipSuccess;
}
// fuseFilter != nullptr: the stereo association's median cut (pair 0) runs inside the gather launch (k_stereo_filter_pack);
// kpsPacked: keypoints and descriptors have already left with the association's launch
static hipError_t enqueue_result_pack(orbx_extractor* ex, int nimg, bool stereo, const StereoArgs* fuseFilter = nullptr,
                                      bool kpsPacked = false) {
  ResultPack a{};
  hipError_t e = make_result_pack(ex, nimg, stereo, &a);
  if (e != hipSuccess) return e;
  if (fuseFilter && stereo && nimg == 2) return launch_stereo_filter_pack(*fuseFilter, a, ex->stream, kpsPacked);
  return launch_result_pack(a, ex->stream);
}

// ---- host copy of the pyramid (the reference's public mvImagePyramid) -----------------------------------------------
static size_t host_pyr_image_bytes(const orbx_extractor* ex) {
  return (((size_t)ex->stagePitch * ex->maxH + 255) & ~(size_t)255) + (((size_t)ex->gmax.pyrImg + 255) & ~(size_t)255);
}
// The kernels of a single-frame host entry behind its upload(s), on the handle's stream: resize chain, then -- when the host
// copy of the pyramid is kept -- the event its copies wait for, then k_detect .. k_describe.  (The hipGraph replay of
// ORBX_GRAPH=1 keeps the plain order of enqueue_extract; its host copies start behind the frame's last kernel.)
static int enqueue_frame(orbx_extractor* ex, int n, bool lapTrivial, const int32_t* lap) {
  static const bool useGraph = getenv("ORBX_GRAPH") && atoi(getenv("ORBX_GRAPH")) != 0;
  hipStream_t st = ex->stream;
  if (ex->keepHostPyr && !ex->evPyr) HIPC(hipEventCreateWithFlags(&ex->evPyr, hipEventDisableTiming));
  if (useGraph && n == 1) {
    int rc = enqueue_extract(ex, ex->pyr.l0, n, ex->curW, ex->curH, (ptrdiff_t)ex->pyr.l0Row, (ptrdiff_t)ex->pyr.l0Img, lap);
    if (rc != ORBX_OK) return rc;
    if (ex->keepHostPyr) HIPC(hipEventRecord(ex->evPyr, st));
    return ORBX_OK;
  }
  int rc = enqueue_pyramid(ex, st, 0, n);
  if (rc != ORBX_OK) return rc;
  if (ex->keepHostPyr) HIPC(hipEventRecord(ex->evPyr, st));
  return record_pipeline(ex, n, lapTrivial, false, true);
}

// Copies of every level of images [0, nimg) of the current extraction into hostPyrAll, on streamPyr behind `after` (an event
// recorded once the pyramids are complete): the copies run on the DMA engines beside k_detect .. k_describe.
static int enqueue_host_pyramid(orbx_extractor* ex, int nimg, hipEvent_t after, const uint8_t* const* hostImg,
                                const ptrdiff_t* hostStride) {
  if (!ex->streamPyr) HIPC(hipStreamCreateWithFlags(&ex->streamPyr, hipStreamNonBlocking));
  const size_t per = host_pyr_image_bytes(ex);
  if (ex->hostPyrAllBytes < 2 * per) {
    if (ex->hostPyrAll) (void)hipHostFree(ex->hostPyrAll);
    ex->hostPyrAll = nullptr;
    ex->hostPyrAllBytes = 0;
    HIPC(hipHostMalloc(reinterpret_cast<void**>(&ex->hostPyrAll), 2 * per, hipHostMallocDefault));
    ex->hostPyrAllBytes = 2 * per;
  }
  const Geom& g = ex->g;
  const size_t restDst = ((size_t)ex->stagePitch * ex->maxH + 255) & ~(size_t)255;
  const LevelDev& LL = g.lv[g.nlevels - 1];
  const size_t restOff = g.nlevels > 1 ? (size_t)g.lv[1].off : 0;
  const size_t restBytes = g.nlevels > 1 ? (size_t)LL.off + (size_t)LL.pitch * LL.h - restOff : 0;
  const int p0 = (int)ex->pyr.l0Row;
  if ((size_t)p0 * g.lv[0].h > restDst) return fail(ORBX_E_CAPACITY, "level-0 pitch larger than the handle's staging pitch");
  // level 0 IS the caller's image: the host thread copies it itself while the GPU works (it would only wait otherwise) -- a
  // third of the bytes never cross the link a second time
  for (int i = 0; i < nimg; i++) {
    uint8_t* H = ex->hostPyrAll + (size_t)i * per;
    const uint8_t* src = hostImg[i];
    const size_t w = (size_t)g.lv[0].w;
    if ((size_t)hostStride[i] == (size_t)p0 && (size_t)p0 == w) {
      std::memcpy(H, src, w * g.lv[0].h);
    } else {
      for (int y = 0; y < g.lv[0].h; y++) std::memcpy(H + (size_t)y * p0, src + (size_t)y * hostStride[i], w);
    }
  }
  // levels 1.. : one DMA copy per image of its block of the device pyramid (4 MB per 1280x720 stereo frame: the ~100 us
  // between the pyramid and the frame's last kernel are just enough for them on a PCIe 5 x16 link).  The HOST waits for the
  // pyramid event (polling: it has nothing else to do until the frame's synchronisation) and then issues the copies --
  // a hipStreamWaitEvent on the copy stream was resolved ~90 us late by this runtime (the copies started when the frame's
  // kernels were over: +82 us per frame, profiles/r5b_host_pyramid.txt).
  static const bool streamWait = getenv("ORBX_PYR_STREAM_WAIT") != nullptr;  // (the measured alternative)
  if (streamWait) {
    HIPC(hipStreamWaitEvent(ex->streamPyr, after, 0));
  } else {
    // bounded poll: ~200 us of polling with a pause between the queries -- four frames' worth of kernels --, then
    // the blocking wait (a wedged stream must not spin a core forever; two eye threads poll at once in the reference's flow)
    hipError_t q = hipErrorNotReady;
    const auto tp0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1; (q = hipEventQuery(after)) == hipErrorNotReady; spin++) {
      cpu_pause();
      if ((spin & 63u) == 0 && std::chrono::steady_clock::now() - tp0 > std::chrono::microseconds(200)) {
        q = hipEventSynchronize(after);
        break;
      }
    }
    if (q != hipSuccess) return fail(ORBX_E_HIP, std::string("pyramid event: ") + hipGetErrorString(q));
  }
  for (int i = 0; i < nimg; i++)
    if (restBytes)
      HIPC(hipMemcpyAsync(ex->hostPyrAll + (size_t)i * per + restDst, ex->pyr.pyr + (long long)i * g.pyrImg + restOff, restBytes,
                          hipMemcpyDeviceToHost, ex->streamPyr));
  ex->hostPyrImages = nimg;
  ex->hostPyrL0Pitch = p0;
  return ORBX_OK;
}

int orbx_set_host_pyramid(orbx_extractor* ex, int enable) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  ex->keepHostPyr = enable != 0;
  if (!ex->keepHostPyr) ex->hostPyrImages = 0;
  return ORBX_OK;
}

int orbx_host_pyramid_level(const orbx_extractor* ex, int image, int level, const uint8_t** data, int* w, int* h,
                            ptrdiff_t* stride) {
  if (!ex || !data) return fail(ORBX_E_BADARG, "null argument");
  *data = nullptr;
  if (!ex->keepHostPyr) return fail(ORBX_E_BADARG, "orbx_set_host_pyramid(handle, 1) first");
  if (image < 0 || image >= ex->hostPyrImages || level < 0 || level >= ex->g.nlevels)
    return fail(ORBX_E_BADARG, "no such level in the host copy of the last single-frame extraction");
  const LevelDev& L = ex->g.lv[level];
  const size_t per = host_pyr_image_bytes(ex);
  const size_t restDst = ((size_t)ex->stagePitch * ex->maxH + 255) & ~(size_t)255;
  const uint8_t* H = ex->hostPyrAll + (size_t)image * per;
  if (level == 0) {
    *data = H;
    if (stride) *stride = ex->hostPyrL0Pitch;
  } else {
    *data = H + restDst + ((size_t)L.off - (size_t)ex->g.lv[1].off);
    if (stride) *stride = L.pitch;
  }
  if (w) *w = L.w;
  if (h) *h = L.h;
  return ORBX_OK;
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
  const int32_t lap[2] = {lap0, lap1};
  if (!n_out) return fail(ORBX_E_BADARG, "null argument");
  hipStream_t st = ex->stream;
  bool lapTrivial = true;
  rc = prepare_extract(ex, ex->d_stage.p, 1, w, h, pitch, (ptrdiff_t)pitch * h, lap, lapTrivial);
  if (rc != ORBX_OK) return rc;
  HIPC(hipMemcpy2DAsync(ex->d_stage.p, pitch, img, stride, w, h, hipMemcpyHostToDevice, st));
  rc = enqueue_frame(ex, 1, lapTrivial, lap);
  if (rc != ORBX_OK) return rc;
  const size_t oc = (size_t)ex->gmax.outCap;  // results through pinned memory: async copies, one synchronisation
  uint8_t* H = ex->hostResults;
  HIPC(enqueue_result_pack(ex, 1, false));   // one gather kernel writes the pinned block (count-trimmed), no D2H copies
  if (ex->keepHostPyr) {
    const ptrdiff_t hs[1] = {stride};
    const uint8_t* const hi[1] = {img};
    rc = enqueue_host_pyramid(ex, 1, ex->evPyr, hi, hs);
    if (rc != ORBX_OK) return rc;
  }
  HIPC(hipStreamSynchronize(st));
  if (ex->keepHostPyr) HIPC(hipStreamSynchronize(ex->streamPyr));
  const int n = *reinterpret_cast<const int*>(H), mono = *reinterpret_cast<const int*>(H + 8);
  *n_out = n;
  ex->hostResImages = 1;
  ex->hostResStereo = false;
  if ((kps || desc) && n > cap) return fail(ORBX_E_CAPACITY, "keypoint buffer too small");
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
  const auto tq0 = std::chrono::steady_clock::now();
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const int pitch = align_up(w, 64);
  const size_t imgBytes = (size_t)pitch * h;
  const int32_t lap[4] = {lap_left ? lap_left[0] : 0, lap_left ? lap_left[1] : 0, lap_right ? lap_right[0] : 0,
                          lap_right ? lap_right[1] : 0};
  hipStream_t st = ex->stream;
  // ONE stream.  (Measured: the right eye's upload + pyramid on a second stream beside the left eye's -- 0.243 ms per frame
  // against 0.220 ms for this order, profiles/r5b_frame_trace*.txt: the two uploads do not overlap on this runtime, the right
  // eye's chain is the critical one either way, and the cross-stream join costs 10 us.)
  bool lapTrivial = true;
  rc = prepare_extract(ex, ex->d_stage.p, 2, w, h, pitch, (ptrdiff_t)imgBytes, lap, lapTrivial);
  if (rc != ORBX_OK) return rc;
  static const bool copy1d = getenv("ORBX_UPLOAD_1D") != nullptr;   // measurement aid
  if (copy1d && stride_left == w && stride_right == w && pitch == w) {
    HIPC(hipMemcpyAsync(ex->d_stage.p, img_left, imgBytes, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(ex->d_stage.p + imgBytes, img_right, imgBytes, hipMemcpyHostToDevice, st));
  } else {
    HIPC(hipMemcpy2DAsync(ex->d_stage.p, pitch, img_left, stride_left, w, h, hipMemcpyHostToDevice, st));
    HIPC(hipMemcpy2DAsync(ex->d_stage.p + imgBytes, pitch, img_right, stride_right, w, h, hipMemcpyHostToDevice, st));
  }
  rc = enqueue_frame(ex, 2, lapTrivial, lap);
  if (rc != ORBX_OK) return rc;
  const bool stereo = bf > 0.f;   // (uright / depth NULL: the results stay in the host block, orbx_host_results)
  StereoArgs sargs;
  const bool fuse = stereo && !ex->profiling;  // (the stage table keeps the filter as its own launch)
  bool kpsPacked = false;
  if (stereo) {
    ResultPack rp{};
    if (fuse) {
      HIPC(make_result_pack(ex, 2, true, &rp));
      // the gather workgroups count their arrivals in d_packCtr and the last one resets it: a frame that failed between launch
      // and synchronisation may have left it non-zero -- cleared on the stream before the next fused launch
      if (ex->packCtrDirty) HIPC(hipMemsetAsync(ex->d_packCtr.p, 0, 4 * sizeof(int), st));
      ex->packCtrDirty = true;
      rp.packCtr = ex->d_packCtr.p;
      rp.hFlag = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(rp.hCnt) + hr_flag());
      rp.seq = ++ex->packSeq;
      if (rp.seq == 0) rp.seq = ++ex->packSeq;   // (0 is the block's initial value)
    }
    rc = enqueue_stereo_match(ex, 0, ex, 1, 1, bf, b, !fuse, &sargs, fuse ? &rp : nullptr, &kpsPacked);
    if (rc != ORBX_OK) return rc;
  }
  // all results travel with asynchronous copies into pinned memory behind the kernels: one synchronisation in total
  const size_t oc = (size_t)ex->gmax.outCap;
  uint8_t* H = ex->hostResults;
  HIPC(enqueue_result_pack(ex, 2, stereo, fuse ? &sargs : nullptr, kpsPacked));   // one gather kernel writes the pinned block (count-trimmed)
  const auto tqp = std::chrono::steady_clock::now();
  if (ex->keepHostPyr) {
    const ptrdiff_t hs[2] = {stride_left, stride_right};
    const uint8_t* const hi[2] = {img_left, img_right};
    rc = enqueue_host_pyramid(ex, 2, ex->evPyr, hi, hs);
    if (rc != ORBX_OK) return rc;
  }
  static const bool latTimes = getenv("ORBX_LAT_TIMES") != nullptr;   // measurement aid: host-side phases of the call
  const auto tq1 = std::chrono::steady_clock::now();
  // Keypoints and descriptors reach the host block with the association's launch (launch_stereo_match's gather workgroups), two
  // launches before the frame ends: a caller that wants them in its own arrays gets them copied WHILE the association and the
  // median cut run (7 us of reads from memory the GPU has just written, off the end of the call).  The block's sequence word is
  // the gather's last write; the stream is polled beside it so that a failed launch cannot leave this thread spinning.
  bool copiedEarly = false;
  static const bool earlyOut = !(getenv("ORBX_EARLY_COPYOUT") && atoi(getenv("ORBX_EARLY_COPYOUT")) == 0);
  if (kpsPacked && earlyOut && (kps_left || desc_left || kps_right || desc_right)) {
    const volatile uint32_t* flag = reinterpret_cast<const volatile uint32_t*>(H + hr_flag());
    const uint32_t want = ex->packSeq;
    bool seen = false;
    const auto ts0 = std::chrono::steady_clock::now();
    for (unsigned spin = 1;; spin++) {   // bounded: a pause per poll, the stream's state every 1024 polls, at most ~300 us in all
      if (*flag == want) { seen = true; break; }
      cpu_pause();
      if ((spin & 1023u) == 0) {
        if (hipStreamQuery(st) != hipErrorNotReady) { seen = *flag == want; break; }
        if (std::chrono::steady_clock::now() - ts0 > std::chrono::microseconds(300)) break;   // (the copy-after path below takes over)
      }
    }
    if (seen) {
      std::atomic_thread_fence(std::memory_order_acquire);
      const int* c0 = reinterpret_cast<const int*>(H);
      const int nl = c0[0], nr = c0[1];
      if (nl <= cap_left && nr <= cap_right) {
        if (nl > 0 && kps_left) std::memcpy(kps_left, H + hr_kps(oc), (size_t)nl * sizeof(orbx_keypoint));
        if (nl > 0 && desc_left) std::memcpy(desc_left, H + hr_desc(oc), (size_t)nl * 32);
        if (nr > 0 && kps_right) std::memcpy(kps_right, H + hr_kps(oc) + oc * sizeof(orbx_keypoint), (size_t)nr * sizeof(orbx_keypoint));
        if (nr > 0 && desc_right) std::memcpy(desc_right, H + hr_desc(oc) + oc * 32, (size_t)nr * 32);
        copiedEarly = true;
      }
    }
  }
  HIPC(hipStreamSynchronize(st));
  ex->packCtrDirty = false;   // the frame completed: the arrival counter is back at zero
  const auto tq2 = std::chrono::steady_clock::now();
  if (ex->keepHostPyr) HIPC(hipStreamSynchronize(ex->streamPyr));
  if (latTimes) {
    const auto tq3 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::micro>(b - a).count();
    };
    std::fprintf(stderr, "orbx_extract_stereo: enqueue %.1f us (of which host pyramid %.1f), sync %.1f, pyramid sync %.1f\n",
                 us(tq0, tq1), us(tqp, tq1), us(tq1, tq2), us(tq2, tq3));
  }
  const int* cnt = reinterpret_cast<const int*>(H);
  const int* mono = reinterpret_cast<const int*>(H + 8);
  *n_left = cnt[0]; *n_right = cnt[1];
  *mono_left = mono[0]; *mono_right = mono[1];
  ex->hostResImages = 2;
  ex->hostResStereo = stereo;
  if (((kps_left || desc_left || uright || depth) && cnt[0] > cap_left) || ((kps_right || desc_right) && cnt[1] > cap_right))
    return fail(ORBX_E_CAPACITY, "keypoint buffer too small");
  if (!copiedEarly) {
    if (cnt[0] > 0 && kps_left) std::memcpy(kps_left, H + hr_kps(oc), (size_t)cnt[0] * sizeof(orbx_keypoint));
    if (cnt[0] > 0 && desc_left) std::memcpy(desc_left, H + hr_desc(oc), (size_t)cnt[0] * 32);
    if (cnt[1] > 0 && kps_right) std::memcpy(kps_right, H + hr_kps(oc) + oc * sizeof(orbx_keypoint), (size_t)cnt[1] * sizeof(orbx_keypoint));
    if (cnt[1] > 0 && desc_right) std::memcpy(desc_right, H + hr_desc(oc) + oc * 32, (size_t)cnt[1] * 32);
  }
  if (stereo && cnt[0] > 0) {
    if (uright) std::memcpy(uright, H + hr_ur(oc), (size_t)cnt[0] * sizeof(float));
    if (depth) std::memcpy(depth, H + hr_depth(oc), (size_t)cnt[0] * sizeof(float));
  }
  return ORBX_OK;
}

int orbx_host_results(const orbx_extractor* ex, int image, const orbx_keypoint** kps, const uint8_t** desc, int* n, int* mono,
                      const float** uright, const float** depth) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  if (image < 0 || image >= ex->hostResImages)
    return fail(ORBX_E_BADARG, "no such image in the result block of the last orbx_extract / orbx_extract_stereo call");
  const size_t oc = (size_t)ex->gmax.outCap;
  const uint8_t* H = ex->hostResults;
  if (n) *n = reinterpret_cast<const int*>(H)[image];
  if (mono) *mono = reinterpret_cast<const int*>(H + 8)[image];
  if (kps) *kps = reinterpret_cast<const orbx_keypoint*>(H + hr_kps(oc)) + (size_t)image * oc;
  if (desc) *desc = H + hr_desc(oc) + (size_t)image * oc * 32;
  const bool st = image == 0 && ex->hostResStereo;
  if (uright) *uright = st ? reinterpret_cast<const float*>(H + hr_ur(oc)) : nullptr;
  if (depth) *depth = st ? reinterpret_cast<const float*>(H + hr_depth(oc)) : nullptr;
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
    // The extraction never materialises blurred levels (k_describe blurs the keypoints' own windows).  This reader of
    // "the level as GaussianBlur leaves it" (:1074-1076) runs the full-image k_blur once per extraction, on demand.
    if (!ex->blurValid) {
      if (!ex->d_blur.p) HIPC(ex->d_blur.alloc((size_t)ex->maxB * ex->gmax.pyrImg + 256));
      ex->pyr.blur = ex->d_blur.p;
      HIPC(launch_blur(ex->g, ex->pyr, ex->lastN, 0, ex->g.nlevels, ex->stream));
      HIPC(hipStreamSynchronize(ex->stream));
      ex->blurValid = true;
    }
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

int orbx_pyramid_download(orbx_extractor* ex, int image, int n_levels, uint8_t* const* dst, const ptrdiff_t* dst_stride) {
  if (!ex || !dst || !dst_stride) return fail(ORBX_E_BADARG, "null argument");
  if (ex->curW == 0 || image < 0 || image >= ex->lastN || n_levels < 0 || n_levels > ex->g.nlevels)
    return fail(ORBX_E_BADARG, "no such pyramid");
  for (int l = 0; l < n_levels; l++)
    if (dst[l] && dst_stride[l] < ex->g.lv[l].w) return fail(ORBX_E_BADARG, "destination stride smaller than the level width");
  HIPC(hipSetDevice(ex->device));
  // A 2-D device-to-host copy into pageable memory is executed row by row by the runtime (2 800 rows per 1280x720 eye:
  // 17 ms).  The pyramid is therefore fetched as TWO contiguous blocks -- level 0 (pitch x h) and the image's block of
  // levels 1.. (pyrImg bytes) -- into a page-locked staging area with asynchronous 1-D copies, one synchronisation, and
  // the rows are then laid out in the caller's arrays by the CPU (3 MB: ~0.2 ms).
  const Geom& g = ex->g;
  int p0 = 0;
  const uint8_t* l0 = level_ptr(g, ex->pyr, image, 0, p0);
  const size_t l0Bytes = (size_t)p0 * (g.lv[0].h - 1) + g.lv[0].w;
  // the image's block holds an unused level-0 slot first: fetch [start of level 1, end of the last level)
  const LevelDev& LL = g.lv[g.nlevels - 1];
  const size_t restOff = g.nlevels > 1 ? (size_t)g.lv[1].off : 0;
  const size_t restBytes = g.nlevels > 1 ? (size_t)LL.off + (size_t)LL.pitch * LL.h - restOff : 0;
  const size_t need = l0Bytes + restBytes + 64;
  if (ex->hostPyrBytes < need) {
    if (ex->hostPyr) (void)hipHostFree(ex->hostPyr);
    ex->hostPyr = nullptr;
    ex->hostPyrBytes = 0;
    HIPC(hipHostMalloc(reinterpret_cast<void**>(&ex->hostPyr), need, hipHostMallocDefault));
    ex->hostPyrBytes = need;
  }
  uint8_t* H = ex->hostPyr;
  const bool want0 = n_levels > 0 && dst[0];
  bool wantRest = false;
  for (int l = 1; l < n_levels; l++) wantRest = wantRest || dst[l];
  if (want0) HIPC(hipMemcpyAsync(H, l0, l0Bytes, hipMemcpyDeviceToHost, ex->stream));
  if (wantRest)
    HIPC(hipMemcpyAsync(H + l0Bytes, ex->pyr.pyr + (long long)image * g.pyrImg + restOff, restBytes, hipMemcpyDeviceToHost, ex->stream));
  HIPC(hipStreamSynchronize(ex->stream));
  for (int l = 0; l < n_levels; l++) {
    if (!dst[l]) continue;
    const LevelDev& L = g.lv[l];
    const uint8_t* src = l == 0 ? H : H + l0Bytes + ((size_t)L.off - restOff);
    const size_t sp = l == 0 ? (size_t)p0 : (size_t)L.pitch;
    if ((size_t)dst_stride[l] == sp && sp == (size_t)L.w) {
      std::memcpy(dst[l], src, sp * L.h);
    } else {
      for (int y = 0; y < L.h; y++) std::memcpy(dst[l] + (size_t)y * dst_stride[l], src + (size_t)y * sp, (size_t)L.w);
    }
  }
  return ORBX_OK;
}

int orbx_debug_candidates(orbx_extractor* ex, int image, int level, int32_t* xys, int cap) {
  if (!ex || image < 0 || image >= ex->lastN || level < 0 || level >= ex->g.nlevels)
    return fail(ORBX_E_BADARG, "bad argument");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  // read from k_detect's per-cell slots (the quadtree keeps its dense copy in registers: it is not written back), in the cell
  // order the quadtree gathers them, truncated like it (candCap)
  const LevelDev& L = ex->g.lv[level];
  const int cells = L.nCols * L.nRows;
  std::vector<int> cnt(cells);
  HIPC(hipMemcpy(cnt.data(), ex->d_cellCount.p + (long long)image * ex->g.totalCells + L.cellStart, (size_t)cells * sizeof(int),
                 hipMemcpyDeviceToHost));
  std::vector<uint32_t> v((size_t)cells * L.cellCap);
  if (!v.empty())
    HIPC(hipMemcpy(v.data(), ex->d_cellCand.p + (long long)image * ex->g.cellImg + L.cellOff, v.size() * 4, hipMemcpyDeviceToHost));
  int n = 0;
  for (int c = 0; c < cells; c++)
    for (int i = 0; i < cnt[c] && n < L.candCap; i++, n++) {
      if (n >= cap) continue;
      const uint32_t k = v[(size_t)c * L.cellCap + i];
      xys[3 * n] = key_x(k);
      xys[3 * n + 1] = key_y(k);
      xys[3 * n + 2] = key_r(k);
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

// withFilter = false: the caller runs the median cut itself (orbx_extract_stereo: fused with the result gather); *argsOut
// receives the kernels' argument block
static int enqueue_stereo_match(orbx_extractor* left, int first_left, orbx_extractor* right, int first_right,
                                int n_pairs, float bf, float b, bool withFilter, StereoArgs* argsOut,
                                const ResultPack* packWithBand, bool* packedOut) {
  if (packedOut) *packedOut = false;
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
    const size_t capS = std::max(capL, (size_t)right->gmax.outCap);
    HIPC(left->d_rowStart.alloc((size_t)n_pairs * 2 * (left->maxH + 2)));
    HIPC(left->d_srec.alloc((size_t)n_pairs * 2 * capS));
    HIPC(left->d_sdesc.alloc((size_t)n_pairs * 2 * capS * 2));
    left->stereoPairs = n_pairs;
  }
  left->lastStereoPairs = n_pairs;
  if (right != left && hipStreamQuery(right->stream) != hipSuccess) {
    // order left's stream after right's extraction -- unless right's stream has drained (the single-frame entries return
    // synchronised: the reference's two threaded operator() calls, src/Frame.cc:200-203, are both complete here), which saves the
    // event and the cross-stream wait (~10 us of the ~50 us ComputeStereoMatches call)
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
  a.srec = left->d_srec.p;
  a.sdesc = left->d_sdesc.p;
  a.cap = (int)std::max(capL, (size_t)right->gmax.outCap);
  a.imgH = left->curH;
  a.band = (int)std::ceil(2.0f * left->scale.back()) + 2;
  const bool direct = stereo_direct_ok(a, n_pairs);  // a single pair: the band workgroups select their keypoints themselves
  if (!direct) {
    StageTimer t(left, left->stream, ORBX_STAGE_STEREO_MATCH);
    HIPC(launch_stereo_sort(left->g, a, n_pairs, left->stream));
  }
  {
    StageTimer t(left, left->stream, ORBX_STAGE_STEREO_MATCH);
    const bool pack = direct && packWithBand && n_pairs == 1;
    HIPC(launch_stereo_match(left->g, left->pyr, right->pyr, a, n_pairs, left->stream, direct, pack ? packWithBand : nullptr));
    if (pack && packedOut) *packedOut = true;
  }
  if (withFilter) {
    StageTimer t(left, left->stream, ORBX_STAGE_STEREO_FILTER);
    HIPC(launch_stereo_filter(a, n_pairs, left->stream));
  }
  if (argsOut) *argsOut = a;
  return ORBX_OK;
}

int orbx_stereo_match_batch(orbx_extractor* left, int first_left, orbx_extractor* right, int first_right,
                            int n_pairs, float bf, float b) {
  return enqueue_stereo_match(left, first_left, right, first_right, n_pairs, bf, b, true, nullptr);
}

int orbx_stereo_results_device(const orbx_extractor* left, const float** d_uright, const float** d_depth) {
  if (!left) return fail(ORBX_E_BADARG, "null handle");
  if (d_uright) *d_uright = left->d_uR.p;
  if (d_depth) *d_depth = left->d_depth.p;
  return ORBX_OK;
}

int orbx_stereo_download(orbx_extractor* left, int pair, float* uright, float* depth, int cap) {
  if (!left || pair < 0 || pair >= left->lastStereoPairs)
    return fail(ORBX_E_BADARG, "no such pair in the stereo association run since the handle's last extraction");
  HIPC(hipSetDevice(left->device));
  const size_t capL = (size_t)left->gmax.outCap;
  const size_t n = std::min((size_t)std::max(cap, 0), capL);
  // one gather kernel into the handle's pinned block + one synchronisation (two blocking copies into pageable memory cost 2 x ~20 us)
  uint8_t* hd = nullptr;
  HIPC(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), left->hostResults, 0));
  ResultPack a{};
  a.uR = reinterpret_cast<const uint32_t*>(left->d_uR.p + pair * capL);
  a.depth = reinterpret_cast<const uint32_t*>(left->d_depth.p + pair * capL);
  a.hUr = reinterpret_cast<uint32_t*>(hd + hr_ur(capL)); a.hDepth = reinterpret_cast<uint32_t*>(hd + hr_depth(capL));
  a.nimg = 1; a.cap = (int)capL; a.stereo = 1; a.mask = (uright ? 0x10 : 0) | (depth ? 0x20 : 0); a.fixedN = (int)n;
  a.nOut = left->d_nOut.p; a.mono = left->d_mono.p;
  if (a.mask && n) HIPC(launch_result_pack(a, left->stream));
  HIPC(hipStreamSynchronize(left->stream));
  if (uright && n) std::memcpy(uright, left->hostResults + hr_ur(capL), n * sizeof(float));
  if (depth && n) std::memcpy(depth, left->hostResults + hr_depth(capL), n * sizeof(float));
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
    HIPC(left->d_fcnt.alloc((size_t)n_pairs * 2 + 1));   // + the length of the accepted-pair list
    HIPC(left->d_fcand.alloc((size_t)n_pairs * capL * 2));
    left->fisheyePairs = n_pairs;
    left->fisheyeCapR = (int)capR;
  }
  hipStream_t s = left->stream;
  if (right != left && hipStreamQuery(right->stream) != hipSuccess) {  // order left's stream after right's extraction (unless it has drained)
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
  a.cand = reinterpret_cast<uint2*>(left->d_fcand.p); a.candCount = left->d_fcnt.p + 2 * (size_t)n_pairs;
  {
    StageTimer t(left, s, ORBX_STAGE_STEREO_MATCH);
    HIPC(launch_fisheye_batch(a, n_pairs, s));
  }
  return ORBX_OK;
}

int orbx_debug_upload_results(orbx_extractor* ex, int image, const orbx_keypoint* kps, const uint8_t* desc, int n, int mono_index) {
  if (!ex || (!kps && n > 0) || (!desc && n > 0)) return fail(ORBX_E_BADARG, "null argument");
  if (image < 0 || image >= ex->lastN) return fail(ORBX_E_BADARG, "image outside the last extraction");
  const int cap = ex->gmax.outCap;
  if (n < 0 || n > cap || mono_index < 0 || mono_index > n) return fail(ORBX_E_BADARG, "counts outside the handle's capacity");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  if (n > 0) {
    HIPC(hipMemcpy(ex->d_kps.p + (size_t)image * cap, kps, (size_t)n * sizeof(orbx_keypoint), hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_desc.p + (size_t)image * cap * 32, desc, (size_t)n * 32, hipMemcpyHostToDevice));
  }
  HIPC(hipMemcpy(ex->d_nOut.p + image, &n, sizeof(int), hipMemcpyHostToDevice));
  HIPC(hipMemcpy(ex->d_mono.p + image, &mono_index, sizeof(int), hipMemcpyHostToDevice));
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
void orbx_debug_set_clahe_cell_kernel(int on) { debug_set_clahe_cell_kernel(on); }
void orbx_debug_set_remap_lds(int on) { debug_set_remap_lds(on); }
void orbx_debug_set_octree_global(int on) { debug_set_octree_global(on); }
void orbx_debug_set_stereo_direct(int max_pairs) { debug_set_stereo_direct(max_pairs); }
void orbx_debug_set_resize_tail(int first_level, int max_levels, int band_rows) {
  orbx_host::g_tail_first = first_level;
  orbx_host::g_tail_levels = max_levels > 0 ? max_levels : orbx_host::kTailLevels;
  orbx_host::g_tail_rows = band_rows > 0 ? band_rows : orbx_host::kTailRows;
}
int orbx_debug_resize_plan(const orbx_extractor* ex, int32_t* first_level, int32_t* n_levels, int32_t* n_bands, int cap) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  int n = 0;
  for (const TailPlan& tp : ex->tails) {
    if (n < cap) {
      if (first_level) first_level[n] = tp.lA;
      if (n_levels) n_levels[n] = tp.nT;
      if (n_bands) n_bands[n] = tp.nBands;
    }
    n++;
  }
  return n;
}
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

int orbx_debug_score_map(orbx_extractor* ex, int enable) {
  if (!ex) return fail(ORBX_E_BADARG, "null handle");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  // the tap's pointer and its memset are part of a captured pipeline: graphs recorded with the other setting are dropped
  // (a replay would keep writing a freed buffer, or never write a new one)
  drop_graphs(ex);
  if (enable && !ex->d_dbgScore.p) HIPC(ex->d_dbgScore.alloc((size_t)ex->maxB * ex->gmax.pyrImg + 256));
  if (!enable) ex->d_dbgScore.free();
  return ORBX_OK;
}

int orbx_debug_score_level(orbx_extractor* ex, int image, int level, uint8_t* dst, ptrdiff_t dst_stride) {
  if (!ex || !dst || !ex->d_dbgScore.p || ex->curW == 0 || image < 0 || image >= ex->lastN || level < 0 ||
      level >= ex->g.nlevels)
    return fail(ORBX_E_BADARG, "score map not enabled or no such level");
  HIPC(hipSetDevice(ex->device));
  HIPC(hipStreamSynchronize(ex->stream));
  const LevelDev& L = ex->g.lv[level];
  HIPC(hipMemcpy2D(dst, dst_stride, ex->d_dbgScore.p + (long long)image * ex->g.pyrImg + L.off, L.pitch, L.w, L.h,
                   hipMemcpyDeviceToHost));
  return ORBX_OK;
}

int orbx_debug_sincos(int device, const float* angles, int n, int fused, float* sin_out, float* cos_out) {
  if (!angles || !sin_out || !cos_out || n < 0) return fail(ORBX_E_BADARG, "bad argument");
  if (n == 0) return ORBX_OK;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  ScratchBuf<float> d;
  HIPC(d.alloc((size_t)3 * n));
  hipError_t e = hipMemcpy(d.p, angles, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = launch_debug_sincos(d.p, n, fused, d.p + n, d.p + 2 * (size_t)n, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(sin_out, d.p + n, (size_t)n * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(cos_out, d.p + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost);
  d.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return ORBX_OK;
}

}  // extern "C"
