// Runs the oracle's whole path once under -fsanitize=address,undefined (tests/test_oracle_sanitize.py): pyramid,
// per-cell FAST, quadtree, orientation, blur, descriptors, stereo association, brute-force kNN, the guided matchers,
// KB8 triangulation, undistortion and the pre-processing helpers on a deterministic synthetic pair.
// TEST INFRASTRUCTURE ONLY.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orb_oracle.h"

using namespace orbo;

static uint64_t s_state = 0x20220131ull;
static uint32_t rnd() {  // splitmix64
  uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

static void scene(int w, int h, int shift, std::vector<uint8_t>& img) {
  img.assign((size_t)w * h, 110);
  uint64_t keep = s_state;
  s_state = 0x5eed;
  for (int o = 0; o < 260; o++) {
    const int cx = rnd() % w, cy = rnd() % h, rw = 6 + rnd() % 50, rh = 6 + rnd() % 40, v = 20 + rnd() % 215;
    const int dx = 2 + (o * 7) % 40;
    for (int y = cy; y < cy + rh && y < h; y++)
      for (int x = cx; x < cx + rw; x++) {
        const int xx = x - (shift ? dx : 0);
        if (xx >= 0 && xx < w) img[(size_t)y * w + xx] = (uint8_t)v;
      }
  }
  for (auto& p : img) p = (uint8_t)std::min(255, std::max(0, (int)p + (int)(rnd() % 9) - 4));
  s_state = keep;
}

int main() {
  const int w = 400, h = 300, nf = 500;
  std::vector<uint8_t> L, R;
  scene(w, h, 0, L);
  scene(w, h, 1, R);
  Extractor eL(nf, 1.2f, 8, 20, 7), eR(nf, 1.2f, 8, 20, 7);
  std::vector<KeyPoint> kL, kR;
  std::vector<uint8_t> dL, dR;
  const int mL = eL.extract(L.data(), w, h, w, 100, 399, kL, dL);
  const int mR = eR.extract(R.data(), w, h, w, 0, 300, kR, dR);
  std::vector<float> u, dep;
  compute_stereo_matches(eL.pyramid, eR.pyramid, kL, dL.data(), kR, dR.data(), eL.t.scale, eL.t.inv_scale, 0.12f * 532.03f, 0.12f, u,
                         dep);
  std::vector<int> idx2, dist2;
  std::vector<uint8_t> ok;
  bf_knn2(dL.data(), (int)kL.size(), dR.data(), (int)kR.size(), idx2, dist2, ok);
  FrameGrid g;
  g.build(kR, 0, 0, (float)w, (float)h);
  std::vector<float> prev(2 * kL.size());
  for (size_t i = 0; i < kL.size(); i++) {
    prev[2 * i] = kL[i].x;
    prev[2 * i + 1] = kL[i].y;
  }
  std::vector<int> m12;
  const int ni = search_for_initialization(kL, dL.data(), kR, dR.data(), g, prev, m12, 100, 0.9f, true);
  // guided searches with the left keypoints as map points / last-frame points
  std::vector<MapPointView> mps(kL.size());
  std::vector<MapPointRight> mpr(kL.size());
  std::vector<ProjectedPoint> pts(kL.size());
  for (size_t i = 0; i < kL.size(); i++) {
    MapPointView& m = mps[i];
    m.proj_x = kL[i].x + 1.5f; m.proj_y = kL[i].y - 1.0f; m.proj_xr = kL[i].x - 10.f; m.view_cos = (i & 1) ? 0.9f : 0.9985f;
    m.track_depth = 5.f + (float)(i % 70); m.predicted_level = kL[i].octave; m.in_view = (i % 9) != 0; m.bad = (i % 23) == 0;
    m.has_observations = (i % 4) != 0; m.pad_ = 0;
    for (int b = 0; b < 32; b++) m.desc[b] = dL[i * 32 + b];
    mpr[i].proj_yr = kL[i].y; mpr[i].view_cos_r = 0.9f; mpr[i].predicted_level_r = (i % 11) ? kL[i].octave : -1;
    mpr[i].in_view_r = (i % 5) != 0; mpr[i].pad_[0] = mpr[i].pad_[1] = mpr[i].pad_[2] = 0;
    ProjectedPoint& p = pts[i];
    p.u = m.proj_x; p.v = m.proj_y; p.ur = m.proj_xr; p.radius = 7.f * eL.t.scale[kL[i].octave]; p.angle = kL[i].angle;
    p.min_level = kL[i].octave - 1; p.max_level = kL[i].octave + 1; p.valid = m.in_view; p.has_observations = m.has_observations;
    p.pad_[0] = p.pad_[1] = 0;
    for (int b = 0; b < 32; b++) p.desc[b] = dL[i * 32 + b];
  }
  std::vector<uint8_t> occ(kR.size(), 0);
  std::vector<int> match;
  std::vector<float> uRv(kR.size(), -1.f);
  const int np1 = search_by_projection_map(kR, dR.data(), uRv.data(), g, eL.t.scale, mps, 3.f, true, 60.f, 0.8f, occ, match);
  occ.assign(kR.size(), 0);
  const int np2 = search_by_projection_frame(kR, dR.data(), uRv.data(), g, pts, true, occ, match);
  // relocalisation flavour + SearchForTriangulation (feature vectors: node = first descriptor byte / 8)
  occ.assign(kR.size(), 0);
  for (size_t i = 0; i < occ.size(); i += 5) occ[i] = 1;
  const int np5 = search_by_projection_keyframe(kR, dR.data(), g, pts, 100, true, occ, match);
  auto make_fv = [](const std::vector<uint8_t>& d, std::vector<uint32_t>& nodes, std::vector<int>& start, std::vector<uint32_t>& feat) {
    start.assign(1, 0);
    for (uint32_t nid = 0; nid < 32; nid++) {
      const size_t before = feat.size();
      for (size_t i = 0; i < d.size() / 32; i++)
        if ((uint32_t)(d[i * 32] >> 3) == nid && (i % 29) != 0) feat.push_back((uint32_t)i);
      if (feat.size() > before && nid != 7) { nodes.push_back(nid * 3 + 1); start.push_back((int)feat.size()); }
      else feat.resize(before);
    }
  };
  std::vector<uint32_t> n1, f1, n2, f2;
  std::vector<int> s1, s2, tri12;
  make_fv(dL, n1, s1, f1);
  make_fv(dR, n2, s2, f2);
  std::vector<uint8_t> hm1(kL.size(), 0), hm2(kR.size(), 0);
  for (size_t i = 0; i < hm1.size(); i += 4) hm1[i] = 1;
  for (size_t i = 1; i < hm2.size(); i += 4) hm2[i] = 1;
  const float epi[2] = {0.5f * w, 0.5f * h}, F12[9] = {1e-6f, 2e-6f, 0.f, -2e-6f, 1e-6f, -1.f, 1e-4f, 1.f, -3.f};
  const int np6 = search_for_triangulation(n1, s1, f1, kL, dL.data(), hm1.data(), nullptr, n2, s2, f2, kR, dR.data(), hm2.data(),
                                           uRv.data(), eL.t.scale, eL.t.sigma2, epi, F12, false, false, true, tri12) +
                  search_for_triangulation(n1, s1, f1, kL, dL.data(), hm1.data(), nullptr, n2, s2, f2, kR, dR.data(), hm2.data(),
                                           nullptr, eL.t.scale, eL.t.sigma2, epi, F12, false, true, true, tri12);
  // SearchByBoW(KeyFrame*, KeyFrame*) and the Fuse / SearchBySim3 search on the same feature vectors / points
  std::vector<float> angL(kL.size()), angR(kR.size());
  for (size_t i = 0; i < kL.size(); i++) angL[i] = kL[i].angle;
  for (size_t i = 0; i < kR.size(); i++) angR[i] = kR[i].angle;
  std::vector<uint8_t> good1(kL.size(), 1), good2(kR.size(), 1);
  for (size_t i = 0; i < good1.size(); i += 6) good1[i] = 0;
  for (size_t i = 2; i < good2.size(); i += 6) good2[i] = 0;
  std::vector<int> bow12;
  const int np7 = search_by_bow_keyframes(n1, s1, f1, dL.data(), angL.data(), good1.data(), (int)kL.size(), n2, s2, f2, dR.data(),
                                          angR.data(), good2.data(), (int)kR.size(), 0.75f, true, bow12);
  std::vector<FusePoint> fps(kL.size());
  for (size_t i = 0; i < kL.size(); i++) {
    FusePoint& p = fps[i];
    p.u = kL[i].x + 0.5f; p.v = kL[i].y - 0.5f; p.ur = p.u - 8.f; p.predicted_level = kL[i].octave;
    p.radius = 3.f * eL.t.scale[kL[i].octave]; p.valid = (i % 7) != 0; p.pad_[0] = p.pad_[1] = p.pad_[2] = 0;
    for (int b = 0; b < 32; b++) p.desc[b] = dL[i * 32 + b];
  }
  std::vector<int> fbi, fbd;
  const int np8 = fuse_search(kR, dR.data(), uRv.data(), g, eL.t.inv_sigma2, fps, 50, fbi, fbd) +
                  fuse_search(kR, dR.data(), nullptr, g, std::vector<float>(8, 0.f), fps, 100, fbi, fbd);
  int np6r = 0;
  {  // two-camera-rig branch of SearchForTriangulation on the concatenated (left | right) features
    TriRig tr;
    const float camp[8] = {190.97f, 190.97f, 200.f, 150.f, 0.0035f, 0.0007f, -0.002f, 0.0002f};
    for (int c = 0; c < 4; c++)
      for (int i = 0; i < 8; i++) tr.cam[c][i] = camp[i];
    tr.precision = 1e-6f;
    const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int c = 0; c < 4; c++) {
      for (int i = 0; i < 9; i++) tr.R[c][i] = I9[i];
      tr.t[c][0] = (c == 1) ? 0.1f : (c == 2 ? -0.1f : 0.f); tr.t[c][1] = 0.f; tr.t[c][2] = 0.f;
    }
    std::vector<KeyPoint> kk2(kL);
    kk2.insert(kk2.end(), kR.begin(), kR.end());
    std::vector<uint8_t> dd2(dL), hm(kk2.size(), 0);
    dd2.insert(dd2.end(), dR.begin(), dR.end());
    std::vector<uint32_t> nn, ff;
    std::vector<int> ss, rigm;
    make_fv(dd2, nn, ss, ff);
    std::vector<uint8_t> bl;
    np6r = search_for_triangulation_rig(nn, ss, ff, kk2, dd2.data(), hm.data(), (int)kL.size(), nn, ss, ff, kk2, dd2.data(), hm.data(),
                                            (int)kL.size(), eL.t.sigma2, eL.t.sigma2, tr, false, false, true, rigm, &bl);
  }
  // stereo-fisheye flavours on the concatenated frame
  std::vector<KeyPoint> kk(kR);
  kk.insert(kk.end(), kL.begin(), kL.end());
  std::vector<uint8_t> dd(dR);
  dd.insert(dd.end(), dL.begin(), dL.end());
  FrameGrid gl, gr;
  gl.build(kR, 0, 0, (float)w, (float)h);
  gr.build(kL, 0, 0, (float)w, (float)h);
  std::vector<int> l2r(kR.size(), -1), r2l(kL.size(), -1);
  for (size_t i = 0; i < kR.size() && i < kL.size(); i += 3) {
    l2r[i] = (int)i;
    r2l[i] = (int)i;
  }
  std::vector<uint8_t> occ2(kk.size(), 0);
  for (auto& m : mps) m.proj_xr = m.proj_x;
  const int np3 = search_by_projection_map_fisheye(kk, dd.data(), (int)kR.size(), gl, gr, eL.t.scale, mps, mpr, 3.f, true, 60.f, 0.8f, l2r,
                                                   r2l, occ2, match);
  std::vector<float> uvr(2 * pts.size());
  for (size_t i = 0; i < pts.size(); i++) {
    uvr[2 * i] = pts[i].u;
    uvr[2 * i + 1] = pts[i].v;
  }
  occ2.assign(kk.size(), 0);
  const int np4 = search_by_projection_frame_fisheye(kk, dd.data(), (int)kR.size(), gl, gr, pts, uvr.data(), true, occ2, match);
  // fisheye association
  KB8 c1, c2;
  const float cam[8] = {190.97f, 190.97f, 200.f, 150.f, 0.0035f, 0.0007f, -0.002f, 0.0002f};
  for (int i = 0; i < 8; i++) c1.p[i] = c2.p[i] = cam[i];
  const float R12[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t12[3] = {0.1f, 0.f, 0.f};
  std::vector<int> a12, a21;
  std::vector<float> fdep, p3d;
  int nd = 0;
  const int nfm = compute_stereo_fisheye_matches(kL, dL.data(), mL, kR, dR.data(), mR, c1, c2, R12, t12, eL.t.sigma2, a12, a21, fdep,
                                                 p3d, &nd, nullptr);
  int nfm_bow = 0;
  // undistortion + pre-processing
  const float K[4] = {458.654f, 457.296f, 200.f, 150.f}, D[4] = {-0.2834f, 0.0740f, 0.00019f, 1.76e-05f};
  std::vector<KeyPoint> un;
  undistort_keypoints(kL, K, D, 4, un);
  float bounds[4];
  compute_image_bounds(w, h, K, D, 4, bounds);
  std::vector<uint8_t> rgb((size_t)w * h * 3), gray((size_t)w * h), small((size_t)300 * 225 * 3);
  for (size_t i = 0; i < rgb.size(); i++) rgb[i] = L[i / 3];
  cvt_gray_u8(rgb.data(), w, h, (ptrdiff_t)w * 3, 3, true, gray.data(), w);
  resize_linear_u8c(rgb.data(), w, h, (ptrdiff_t)w * 3, 3, small.data(), 300, 225, 300 * 3);
  // rectification maps that also point outside the source / hold NaN, and CLAHE on a size that does not divide into 8 x 8 tiles
  const int rw = w - 13, rh = h - 7;
  std::vector<float> mapx((size_t)rw * rh), mapy((size_t)rw * rh);
  for (int y = 0; y < rh; y++)
    for (int x = 0; x < rw; x++) {
      mapx[(size_t)y * rw + x] = 1.04f * x - 9.3f + 0.01f * y;
      mapy[(size_t)y * rw + x] = 1.03f * y - 6.7f - 0.02f * x;
    }
  mapx[0] = std::nanf(""); mapx[1] = 3e9f; mapy[2] = -3e9f; mapx[3] = -1.f; mapy[3] = (float)h;
  std::vector<uint8_t> rect((size_t)rw * rh), eq((size_t)rw * rh);
  remap_linear_u8(L.data(), w, h, w, mapx.data(), mapy.data(), rw, rect.data(), rw, rh, rw);
  clahe_u8(rect.data(), rw, rh, rw, 3.0, 8, 8, eq.data(), rw);
  // bag of words: a small irregular tree (leaves at two depths, one stopped word), transform + SearchByBoW on the frame's own
  // descriptors (keyframe = left eye, frame = left | right eyes)
  {
    std::vector<int> par(1, 0);
    std::vector<uint8_t> leaf(1, 0), nd(32, 0);
    std::vector<double> wts(1, 0.0);
    auto add = [&](int p, bool lf, const uint8_t* d, double wgt) {
      par.push_back(p); leaf.push_back(lf); nd.insert(nd.end(), d, d + 32); wts.push_back(wgt);
      return (int)par.size() - 1;
    };
    for (int c = 0; c < 5; c++) {
      const int id = add(0, c == 4, dL.data() + (size_t)(c * 37 % std::max(1, (int)kL.size())) * 32, c == 4 ? 2.5 : 0.0);
      if (c < 4)
        for (int e = 0; e < 3; e++) add(id, true, dR.data() + (size_t)((c * 3 + e) * 11 % std::max(1, (int)kR.size())) * 32, e == 2 && c == 1 ? 0.0 : 1.0 + e);
    }
    Vocabulary voc;
    voc.build(5, 2, 0, 0, (int)par.size(), par.data(), leaf.data(), nd.data(), wts.data());
    std::vector<uint32_t> w1, n1, f1, w2, n2, f2;
    std::vector<double> v1, v2;
    std::vector<int> s1, s2, bm;
    std::vector<uint8_t> both(dL);
    both.insert(both.end(), dR.begin(), dR.end());
    std::vector<float> angK, angF;
    for (auto& k : kL) { angK.push_back(k.angle); angF.push_back(k.angle); }
    for (auto& k : kR) angF.push_back(k.angle);
    bow_transform(voc, dL.data(), (int)kL.size(), 1, w1, v1, n1, s1, f1);
    bow_transform(voc, both.data(), (int)(kL.size() + kR.size()), 1, w2, v2, n2, s2, f2);
    std::vector<uint8_t> valid(kL.size(), 1);
    nfm_bow = search_by_bow(n1, s1, f1, dL.data(), angK.data(), valid.data(), n2, s2, f2, both.data(), angF.data(),
                            (int)(kL.size() + kR.size()), (int)kL.size(), 0.7f, true, bm);
  }
  std::printf("ok %d %d %zu %zu stereo %d knn %d init %d proj %d %d fe %d %d fisheye %d/%d un %.2f b %.1f g %d\n", mL, mR, kL.size(),
              kR.size(), (int)u.size(), (int)ok.size(), ni, np1, np2, np3, np4, nfm, nd, un.empty() ? 0.f : un[0].x, bounds[0], gray[5] + eq[7] + nfm_bow + np5 + np6 + np6r + np7 + np8);
  return 0;
}
