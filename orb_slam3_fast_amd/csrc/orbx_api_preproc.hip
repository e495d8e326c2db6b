// orbx_api_preproc.hip — C ABI of the image pre-processing (gray, resize, cv::remap, CLAHE, the device-resident orbx_preproc chain) and of
// Frame::UndistortKeyPoints / ComputeImageBounds.
#include "orbx_host.h"

extern "C" {

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
  DevBuf<int> d_remapTab;          // k_remap_lds footprints (remap_tile_table), when every tile of the plan fits
  int remapTilesX = 0, remapTilesY = 0;
  bool remapLds = false;
  DevBuf<int> d_xofs, d_yofs;
  DevBuf<short> d_xab, d_yab;
  DevBuf<uint4> d_rxtab;           // single-channel resize through the pyramid's kernel (launch_resize_plain): its tables
  DevBuf<uint32_t> d_ryrow;
  DevBuf<short> d_ryab;
  bool resizeFast = false;
  DevBuf<uint8_t> d_clahe, d_lut, d_geo, d_gray;
  DevBuf<uint32_t> d_cells;
  long long clahePitch = 0, geoPitch = 0, grayPitch = 0;
  int outW = 0, outH = 0;
  const uint8_t* out = nullptr;  // result of the last run (a stage buffer, or the caller's frames when nothing is enabled)
  long long outPitch = 0, outImgPitch = 0;
  ~orbx_preproc() {
    d_mapx.free(); d_mapy.free(); d_remapTab.free(); d_xofs.free(); d_yofs.free(); d_xab.free(); d_yab.free(); d_rxtab.free(); d_ryrow.free(); d_ryab.free();
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
    if (cn == 1) {   // the LDS-staged kernel needs every output tile's source footprint (orbx_preproc.hip remap_tile_table)
      std::vector<int> tab;
      pp->remapLds = remap_tile_table(p->map_x, p->map_y, (long long)ms, p->out_w, p->out_h, p->src_w, p->src_h, p->n_maps, tab,
                                      pp->remapTilesX, pp->remapTilesY);
      if (pp->remapLds) {
        chk(pp->d_remapTab.alloc(tab.size()));
        if (e == hipSuccess) chk(hipMemcpy(pp->d_remapTab.p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
      }
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
    // single-channel frames: the same cv::resize through the pyramid's tiled kernel (tables of a two-level geometry)
    static const bool plainOff = getenv("ORBX_RESIZE_PLAIN") && atoi(getenv("ORBX_RESIZE_PLAIN")) == 0;   // A / B switch
    if (cn == 1 && !plainOff && p->out_w >= 4 && resize_plain_lds(p->src_w, p->src_h, p->out_w, p->out_h) <= 60 * 1024) {
      Geom g2{};
      g2.nlevels = 2;
      g2.lv[0].w = p->src_w; g2.lv[0].h = p->src_h;
      g2.lv[1].w = p->out_w; g2.lv[1].h = p->out_h;
      std::vector<uint4> xt;
      std::vector<int> yo;
      std::vector<short> ya;
      build_coefs(g2, xt, yo, ya);
      std::vector<uint32_t> yrow(yo.size(), 0);
      for (int dy = 0; dy < p->out_h; dy++) {
        const int a = std::min(std::max(yo[dy], 0), p->src_h - 1), b = std::min(std::max(yo[dy] + 1, 0), p->src_h - 1);
        yrow[dy] = (uint32_t)a | ((uint32_t)b << 16);
      }
      chk(pp->d_rxtab.alloc(xt.size() + 64)); chk(pp->d_ryrow.alloc(yrow.size() + 64)); chk(pp->d_ryab.alloc(ya.size() + 128));
      if (e == hipSuccess) chk(hipMemcpy(pp->d_rxtab.p, xt.data(), xt.size() * sizeof(uint4), hipMemcpyHostToDevice));
      if (e == hipSuccess) chk(hipMemcpy(pp->d_ryrow.p, yrow.data(), yrow.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      if (e == hipSuccess) chk(hipMemcpy(pp->d_ryab.p, ya.data(), ya.size() * sizeof(short), hipMemcpyHostToDevice));
      if (e == hipSuccess) chk(prepare_resize_plain(p->src_w, p->src_h, p->out_w, p->out_h));
      pp->resizeFast = e == hipSuccess;
    }
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
  if ((long long)row_pitch * p.src_h > 0x7fffffffll) return fail(ORBX_E_UNSUPPORTED, "frames larger than 2 GiB (32-bit tap offsets)");
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
    if (pp->remapLds && !(((uintptr_t)cur | (uintptr_t)cp | (uintptr_t)cip) & 15)) {   // 16-byte staging pieces: aligned rows
      a.tileTab = pp->d_remapTab.p; a.tilesX = pp->remapTilesX; a.tilesY = pp->remapTilesY;
    }
    e = launch_remap(a, n, s);
    cur = a.dst; cp = a.dstPitch; cip = a.dstImgPitch;
  } else if (e == hipSuccess && pp->doResize) {
    if (pp->resizeFast && cp < (1ll << 31))
      e = launch_resize_plain(cur, p.src_w, p.src_h, cp, cip, pp->d_geo.p, pp->outW, pp->outH, pp->geoPitch, pp->geoPitch * pp->outH,
                              pp->d_rxtab.p, pp->d_ryrow.p, pp->d_ryab.p, n, s);
    else
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

}  // extern "C"
