// Drives the C++ mirror classes the way the reference's Frame does (src/Frame.cc:149-232): two extractor
// instances called from two threads, then ComputeStereoMatches; optional SearchForInitialization between
// two mono frames.  Reads raw u8 images, writes results as flat binary files for the pytest to diff
// against the oracle.   usage: frame_like <w> <h> <nfeat> <left.raw> <right.raw> <outprefix>
// Built twice by tests/test_cpp_mirror.py: with -DORBX_NO_OPENCV (the cvlite stand-ins) and with
// -Itests/cpp/opencv_stub (the cv::InputArray / cv::OutputArray / cv::Mat branch = the reference's own signatures).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../../orb_slam3_fast_amd/csrc/ORBextractor.h"
#include "../../orb_slam3_fast_amd/csrc/ORBmatcher.h"
#include "../../orb_slam3_fast_amd/csrc/ORBVocabulary.h"
#include "../../orb_slam3_fast_amd/csrc/Preprocess.h"

using namespace ORB_SLAM3;

// a gray image header over caller-owned pixels, in whichever Mat the mirror was built with
static ocv::Mat wrap(int h, int w, uint8_t* p) {
#ifdef ORBX_HAVE_OPENCV
  return cv::Mat(h, w, CV_8UC1, p, (size_t)w);
#else
  return ocv::Mat(h, w, p, (size_t)w);
#endif
}
// The slice of ORB_SLAM3::Frame that ORBmatcher::SearchForInitialization(Frame&, Frame&, ...) reads
// (include/Frame.h:249-272: mvKeysUn, mDescriptors, N; :314-319: the static image bounds).
struct FrameLike {
  std::vector<ocv::KeyPoint> mvKeysUn;
  ocv::Mat mDescriptors;
  int N = 0;
  static float mnMinX, mnMinY, mnMaxX, mnMaxY;
};
float FrameLike::mnMinX = 0, FrameLike::mnMinY = 0, FrameLike::mnMaxX = 0, FrameLike::mnMaxY = 0;

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T>
static void dump(const std::string& path, const T* p, size_t n) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}

int main(int argc, char** argv) {
  if (argc < 2) {
    // no arguments: construction must fail loudly without a GPU, succeed with one
    try {
      ORBextractor ex(1000, 1.2f, 8, 20, 7, 640, 480);
      std::printf("constructed on a GPU\n");
      return 0;
    } catch (const std::exception& e) {
      std::printf("no-device error: %s\n", e.what());
      return 3;
    }
  }
  if (std::string(argv[1]) == "bow") {
    // frame_like bow <voc.txt> <kfDesc.raw> <kfAngle.raw> <kfValid.raw> <fDesc.raw> <fAngle.raw> <levelsup> <nLeftF> <outprefix>
    ORBVocabulary voc;
    if (!voc.loadFromTextFile(argv[2])) {
      std::printf("cannot load %s: %s\n", argv[2], orbx_last_error());
      return 4;
    }
    std::vector<uint8_t> kd = slurp(argv[3]), kab = slurp(argv[4]), kv = slurp(argv[5]), fd = slurp(argv[6]), fab = slurp(argv[7]);
    const int levelsup = std::atoi(argv[8]), nLeftF = std::atoi(argv[9]);
    const std::string out = argv[10];
    const int nk = (int)kd.size() / 32, nf = (int)fd.size() / 32;
    std::vector<ocv::KeyPoint> kk(nk), fk(nf);
    for (int i = 0; i < nk; i++) std::memcpy(&kk[i].angle, kab.data() + 4 * i, 4);
    for (int i = 0; i < nf; i++) std::memcpy(&fk[i].angle, fab.data() + 4 * i, 4);
    DBoW2::BowVector bk, bf;
    DBoW2::FeatureVector vk, vf;
    voc.transform(kd.data(), nk, bk, vk, levelsup);  // KeyFrame::ComputeBoW
    voc.transform(fd.data(), nf, bf, vf, levelsup);  // Frame::ComputeBoW
    std::vector<int> matches;
    const int n = SearchByBoW(vk, kk, kd.data(), kv, vf, fk, fd.data(), nLeftF, 0.7f, true, matches);
    std::vector<uint32_t> w;
    std::vector<double> val;
    for (const auto& e : bf) { w.push_back(e.first); val.push_back(e.second); }
    dump(out + ".words", w.data(), w.size());
    dump(out + ".values", val.data(), val.size());
    dump(out + ".match", matches.data(), matches.size());
    std::printf("bow %u words, %zu / %zu nodes, %d matches\n", voc.size(), vk.size(), vf.size(), n);
    return 0;
  }
  if (std::string(argv[1]) == "reloc_tri") {
    // frame_like reloc_tri <prefix> <w> <h> <ORBdist>: the relocalisation SearchByProjection(Frame&, KeyFrame*, ...) overload and
    // SearchForTriangulation on <prefix>.{k1,d1,k2,d2,pts,occ,n1,s1,f1,n2,s2,f2,mp1,mp2,sf,sg,epF}; writes <prefix>.{rmatch,rocc,m12}
    const std::string pre = argv[2];
    auto rd = [&](const char* ext) { return slurp((pre + "." + ext).c_str()); };
    std::vector<uint8_t> k1b = rd("k1"), d1 = rd("d1"), k2b = rd("k2"), d2 = rd("d2"), ptsb = rd("pts"), occ = rd("occ");
    std::vector<uint8_t> mp1 = rd("mp1"), mp2 = rd("mp2"), sfb = rd("sf"), sgb = rd("sg"), epFb = rd("epF");
    std::vector<ocv::KeyPoint> k1(k1b.size() / 28), k2(k2b.size() / 28);
    std::memcpy(static_cast<void*>(k1.data()), k1b.data(), k1b.size());
    std::memcpy(static_cast<void*>(k2.data()), k2b.data(), k2b.size());
    std::vector<orbx_projected_point> pts(ptsb.size() / sizeof(orbx_projected_point));
    std::memcpy(static_cast<void*>(pts.data()), ptsb.data(), ptsb.size());
    FrameView cur;
    cur.mvKeysUn = k2.data(); cur.mDescriptors = d2.data(); cur.N = (int)k2.size();
    cur.mnMinX = 0; cur.mnMinY = 0; cur.mnMaxX = (float)std::atoi(argv[3]); cur.mnMaxY = (float)std::atoi(argv[4]);
    ORBmatcher matcher(0.75f, true);
    std::vector<int> rmatch;
    const int nr = matcher.SearchByProjection(cur, pts, std::atoi(argv[5]), occ, rmatch);
    dump(pre + ".rmatch", rmatch.data(), rmatch.size());
    dump(pre + ".rocc", occ.data(), occ.size());
    auto fv = [&](const char* n, const char* st, const char* f) {
      std::vector<uint8_t> nb = rd(n), sb = rd(st), fb = rd(f);
      const uint32_t* nodes = reinterpret_cast<const uint32_t*>(nb.data());
      const int32_t* start = reinterpret_cast<const int32_t*>(sb.data());
      const uint32_t* feats = reinterpret_cast<const uint32_t*>(fb.data());
      DBoW2::FeatureVector v;
      for (size_t j = 0; j < nb.size() / 4; j++) v[nodes[j]].assign(feats + start[j], feats + start[j + 1]);
      return v;
    };
    const DBoW2::FeatureVector fv1 = fv("n1", "s1", "f1"), fv2 = fv("n2", "s2", "f2");
    std::vector<float> sf(sfb.size() / 4), sg(sgb.size() / 4), epF(11);
    std::memcpy(sf.data(), sfb.data(), sfb.size());
    std::memcpy(sg.data(), sgb.data(), sgb.size());
    std::memcpy(epF.data(), epFb.data(), 44);
    KeyFrameView a, b;
    a.mFeatVec = &fv1; a.mvKeysUn = &k1; a.mDescriptors = d1.data(); a.hasMapPoint = &mp1; a.mvScaleFactors = &sf; a.mvLevelSigma2 = &sg;
    b.mFeatVec = &fv2; b.mvKeysUn = &k2; b.mDescriptors = d2.data(); b.hasMapPoint = &mp2; b.mvScaleFactors = &sf; b.mvLevelSigma2 = &sg;
    std::vector<std::pair<size_t, size_t>> pairs;
    const int nt = SearchForTriangulation(a, b, epF.data(), epF.data() + 2, pairs, false, false);
    std::vector<int> m12(k1.size(), -1);
    for (const auto& pr : pairs) m12[pr.first] = (int)pr.second;
    dump(pre + ".m12", m12.data(), m12.size());
    // Fuse (search), SearchBySim3 and SearchByBoW(KeyFrame*, KeyFrame*) on <prefix>.{fp,isg,p12,p21,good1,good2}
    std::vector<uint8_t> fpb = rd("fp"), isgb = rd("isg"), p12b = rd("p12"), p21b = rd("p21"), good1 = rd("good1"), good2 = rd("good2");
    auto fpts = [](const std::vector<uint8_t>& b) {
      std::vector<orbx_fuse_point> v(b.size() / sizeof(orbx_fuse_point));
      std::memcpy(static_cast<void*>(v.data()), b.data(), v.size() * sizeof(orbx_fuse_point));
      return v;
    };
    std::vector<float> isg(isgb.size() / 4);
    std::memcpy(isg.data(), isgb.data(), isgb.size());
    std::vector<int> fuseIdx, sim12, bow12;
    const int nf = matcher.Fuse(cur, isg, fpts(fpb), fuseIdx);
    FrameView kf1 = cur;
    kf1.mvKeysUn = k1.data(); kf1.mDescriptors = d1.data(); kf1.N = (int)k1.size();
    const int ns = matcher.SearchBySim3(kf1, cur, fpts(p12b), fpts(p21b), sim12);
    const int nb = SearchByBoW(fv1, k1, d1.data(), good1, fv2, k2, d2.data(), good2, 0.75f, true, bow12);
    dump(pre + ".fuse", fuseIdx.data(), fuseIdx.size());
    dump(pre + ".sim3", sim12.data(), sim12.size());
    dump(pre + ".bowkf", bow12.data(), bow12.size());
    // two-camera key frames: mFeatVec / the map-point flags cover NLeft + NRight features, mvKeysUn only the first part;
    // the reference skips idx >= mvKeysUn.size() (src/ORBmatcher.cc:799,816)
    std::vector<ocv::KeyPoint> k1r(k1.begin(), k1.begin() + (std::ptrdiff_t)(k1.size() * 3 / 4)),
        k2r(k2.begin(), k2.begin() + (std::ptrdiff_t)(k2.size() * 3 / 4));
    std::vector<int> bowRig;
    const int nbr = SearchByBoW(fv1, k1r, d1.data(), good1, fv2, k2r, d2.data(), good2, 0.75f, true, bowRig);
    dump(pre + ".bowkf_rig", bowRig.data(), bowRig.size());
    std::printf("%d %d %zu %d %d %d %d\n", nr, nt, pairs.size(), nf, ns, nb, nbr);
    return 0;
  }
  if (std::string(argv[1]) == "rectify") {
    // frame_like rectify <sw> <sh> <dw> <dh> <L.raw> <R.raw> <maps.raw (M1l M2l M1r M2r, dw*dh floats each)> <outprefix>
    const int sw = std::atoi(argv[2]), sh = std::atoi(argv[3]), dw = std::atoi(argv[4]), dh = std::atoi(argv[5]);
    std::vector<uint8_t> Lb = slurp(argv[6]), Rb = slurp(argv[7]), mb = slurp(argv[8]);
    const std::string out = argv[9];
    const float* m = reinterpret_cast<const float*>(mb.data());
    const size_t per = (size_t)dw * dh;
    ocv::Mat L = wrap(sh, sw, Lb.data()), R = wrap(sh, sw, Rb.data()), eqL, eqR, a, b, c, d;
    auto clahe = createCLAHE(3.0, 8, 8);  // Examples/Stereo/stereo_tum_vi.cc:100,142-143
    clahe->apply(L, eqL);
    clahe->apply(R, eqR);
    remap(eqL, a, m, m + per, dw, dh);  // src/System.cc:294
    remap(eqR, b, m + 2 * per, m + 3 * per, dw, dh);
    StereoRectifier rect(sw, sh, dw, dh, m, m + per, m + 2 * per, m + 3 * per, 3.0, 8);
    rect(L, R, c, d);
    dump(out + ".eqL", eqL.data, (size_t)sw * sh);
    dump(out + ".a", a.data, per);
    dump(out + ".b", b.data, per);
    dump(out + ".c", c.data, per);
    dump(out + ".d", d.data, per);
#ifdef ORBX_HAVE_OPENCV
    {  // the reference's call shape: cv::remap(imLeft, imLeftToFeed, M1l, M2l, cv::INTER_LINEAR)  (src/System.cc:294)
      cv::Mat M1(dh, dw, CV_32FC1, const_cast<float*>(m)), M2(dh, dw, CV_32FC1, const_cast<float*>(m + per)), e;
      remap(eqL, e, M1, M2, cv::INTER_LINEAR);
      if (e.rows != dh || e.cols != dw || std::memcmp(e.data, a.data, per) != 0) {
        std::printf("cv::remap overload differs from the pointer form\n");
        return 5;
      }
    }
#endif
    return 0;
  }
  if (std::string(argv[1]) == "fisheye") {
    // frame_like fisheye <kL.raw> <dL.raw> <monoL> <kR.raw> <dR.raw> <monoR> <rig.raw (29 floats)> <sigma2.raw> <outprefix>
    std::vector<uint8_t> kLb = slurp(argv[2]), dLb = slurp(argv[3]), kRb = slurp(argv[5]), dRb = slurp(argv[6]);
    std::vector<uint8_t> rigb = slurp(argv[8]), sgb = slurp(argv[9]);
    const std::string out = argv[10];
    std::vector<ocv::KeyPoint> kL(kLb.size() / 28), kR(kRb.size() / 28);
    std::memcpy(static_cast<void*>(kL.data()), kLb.data(), kLb.size());
    std::memcpy(static_cast<void*>(kR.data()), kRb.data(), kRb.size());
    orbx_kb8_rig rig;
    static_assert(sizeof(orbx_kb8_rig) == 29 * sizeof(float), "rig is 29 packed floats");
    std::memcpy(&rig, rigb.data(), sizeof(rig));
    std::vector<float> sigma2(sgb.size() / 4);
    std::memcpy(sigma2.data(), sgb.data(), sgb.size());
    std::vector<int> l2r, r2l;
    std::vector<float> depth, uR, p3d;
    const int n = ComputeStereoFishEyeMatches(kL, dLb.data(), std::atoi(argv[4]), kR, dRb.data(), std::atoi(argv[7]), rig, sigma2,
                                              l2r, r2l, depth, uR, p3d);
    dump(out + ".l2r", l2r.data(), l2r.size());
    dump(out + ".r2l", r2l.data(), r2l.size());
    dump(out + ".depth", depth.data(), depth.size());
    dump(out + ".p3d", p3d.data(), p3d.size());
    dump(out + ".uR", uR.data(), uR.size());
    std::printf("%d\n", n);
    return 0;
  }
  if (std::string(argv[1]) == "latency") {
    // frame_like latency <w> <h> <nfeatures> <frames.raw: n x (left, right) gray images> <n> <calls>: the stereo Frame constructor's
    // extraction + ComputeStereoMatches (src/Frame.cc:196-232) through the drop-in class, one frame at a time, timed in C++ --
    // ExtractStereo fills std::vector<cv::KeyPoint> / cv::Mat / mvuRight / mvDepth like the reference's members.  Prints the mean
    // and median per frame with mbKeepHostPyramid = false and with the class's default (true: mvImagePyramid kept current).
    const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), nf = std::atoi(argv[4]), n = std::atoi(argv[6]), calls = std::atoi(argv[7]);
    std::vector<uint8_t> frames = slurp(argv[5]);
    if (frames.size() < (size_t)n * 2 * w * h) return 4;
    ORBextractor ex(nf, 1.2f, 8, 20, 7, w, h);
    std::vector<int> lap = {0, 0};
    for (int keep = 0; keep < 2; keep++) {
      ex.mbKeepHostPyramid = keep != 0;
      std::vector<double> ms;
      size_t nk = 0;
      for (int i = -20; i < calls; i++) {
        uint8_t* pl = frames.data() + (size_t)((i + 20) % n) * 2 * w * h;
        ocv::Mat imL = wrap(h, w, pl), imR = wrap(h, w, pl + (size_t)w * h);
        std::vector<ocv::KeyPoint> kL, kR;
        ocv::Mat dL, dR;
        std::vector<float> uR, depth;
        int mL = 0, mR = 0;
        const auto t0 = std::chrono::steady_clock::now();
        ex.ExtractStereo(imL, imR, kL, dL, kR, dR, lap, lap, mL, mR, 0.12f * 532.03f, 0.12f, &uR, &depth);
        const auto t1 = std::chrono::steady_clock::now();
        if (i >= 0) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        nk += kL.size();
        if (keep && (ex.mvImagePyramid.size() != 8 || ex.mvImagePyramid[0].rows != h)) return 5;
      }
      if (keep == 0 && std::getenv("ORBX_LAT_TWO_THREADS")) {
        // the UNMODIFIED reference flow (src/Frame.cc:200-232): two threads, one operator() each, then ComputeStereoMatches
        ORBextractor exL(nf, 1.2f, 8, 20, 7, w, h), exR(nf, 1.2f, 8, 20, 7, w, h);
        exL.mbKeepHostPyramid = exR.mbKeepHostPyramid = false;
        std::vector<double> m2;
        double csm = 0;
        ocv::Mat mask;
        for (int i = -20; i < calls; i++) {
          uint8_t* pl = frames.data() + (size_t)((i + 20) % n) * 2 * w * h;
          ocv::Mat imL = wrap(h, w, pl), imR = wrap(h, w, pl + (size_t)w * h);
          std::vector<ocv::KeyPoint> kL, kR;
          ocv::Mat dL, dR;
          std::vector<float> uR, depth;
          const auto t0 = std::chrono::steady_clock::now();
          std::thread tl([&] { exL(imL, mask, kL, dL, lap); });
          std::thread tr([&] { exR(imR, mask, kR, dR, lap); });
          tl.join();
          tr.join();
          const auto tj = std::chrono::steady_clock::now();
          ComputeStereoMatches(exL, exR, (int)kL.size(), 0.12f * 532.03f, 0.12f, uR, depth);
          const auto t1 = std::chrono::steady_clock::now();
          if (i >= 0) m2.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
          if (i >= 0) csm += std::chrono::duration<double, std::milli>(t1 - tj).count();
        }
        double mean2 = 0, var2 = 0;
        for (double v : m2) mean2 += v;
        mean2 /= (double)m2.size();
        for (double v : m2) var2 += (v - mean2) * (v - mean2);
        std::sort(m2.begin(), m2.end());
        std::printf("two threads x operator() + ComputeStereoMatches (the unmodified Frame constructor): mean %.4f ms  p50 %.4f  p90 %.4f  std %.4f  p99 %.4f  (ComputeStereoMatches alone %.4f)\n",
                    mean2, m2[m2.size() / 2], m2[m2.size() * 9 / 10], std::sqrt(var2 / (double)m2.size()),
                    m2[std::min(m2.size() - 1, m2.size() * 99 / 100)], csm / (double)m2.size());
      }
      double mean = 0, var = 0;
      for (double v : ms) mean += v;
      mean /= (double)ms.size();
      for (double v : ms) var += (v - mean) * (v - mean);
      std::sort(ms.begin(), ms.end());
      std::printf("ExtractStereo mbKeepHostPyramid=%d: mean %.4f ms  p50 %.4f  p90 %.4f  std %.4f  p99 %.4f  (%d calls, %d distinct frames, %.0f keypoints per left image)\n",
                  keep, mean, ms[ms.size() / 2], ms[ms.size() * 9 / 10], std::sqrt(var / (double)ms.size()), ms[std::min(ms.size() - 1, ms.size() * 99 / 100)],
                  calls, n, (double)nk / (calls + 20));
    }
    return 0;
  }
  const int w = std::atoi(argv[1]), h = std::atoi(argv[2]), nf = std::atoi(argv[3]);
  std::vector<uint8_t> L = slurp(argv[4]), R = slurp(argv[5]);
  const std::string out = argv[6];
  ORBextractor exL(nf, 1.2f, 8, 20, 7, w, h), exR(nf, 1.2f, 8, 20, 7, w, h);
  ocv::Mat imL = wrap(h, w, L.data()), imR = wrap(h, w, R.data()), mask;
  std::vector<ocv::KeyPoint> kL, kR;
  ocv::Mat dL, dR;
  std::vector<int> lap = {0, 0};
  int monoL = 0, monoR = 0;
  std::thread tl([&] { monoL = exL(imL, mask, kL, dL, lap); });  // src/Frame.cc:200-203
  std::thread tr([&] { monoR = exR(imR, mask, kR, dR, lap); });
  tl.join();
  tr.join();
  std::vector<float> uR, depth;
  ComputeStereoMatches(exL, exR, (int)kL.size(), 0.12f * 532.03f, 0.12f, uR, depth);
  {  // mbKeepHostPyramid (the default): mvImagePyramid[l] is a header over the library's page-locked copy, rows `step` apart
    const ocv::Mat& v = exL.mvImagePyramid[3];
    std::vector<uint8_t> rows((size_t)v.rows * v.cols);
    for (int y = 0; y < v.rows; y++) std::memcpy(rows.data() + (size_t)y * v.cols, v.ptr(y), (size_t)v.cols);
    dump(out + ".vpyr3", rows.data(), rows.size());
  }
  exL.SyncImagePyramid();
  dump(out + ".kL", kL.data(), kL.size());
  dump(out + ".dL", dL.data, (size_t)dL.rows * 32);
  dump(out + ".kR", kR.data(), kR.size());
  dump(out + ".dR", dR.data, (size_t)dR.rows * 32);
  dump(out + ".uR", uR.data(), uR.size());
  dump(out + ".depth", depth.data(), depth.size());
  dump(out + ".pyr3", exL.mvImagePyramid[3].data, (size_t)exL.mvImagePyramid[3].rows * exL.mvImagePyramid[3].cols);
  // monocular initialisation match between the two views (stand-in for two consecutive frames)
  FrameView F1, F2;
  F1.mvKeysUn = kL.data(); F1.mDescriptors = dL.data; F1.N = (int)kL.size();
  F2.mvKeysUn = kR.data(); F2.mDescriptors = dR.data; F2.N = (int)kR.size();
  F1.mnMaxX = F2.mnMaxX = (float)w; F1.mnMaxY = F2.mnMaxY = (float)h;
  std::vector<ocv::Point2f> prev(kL.size());
  for (size_t i = 0; i < kL.size(); i++) prev[i] = kL[i].pt;
  std::vector<int> m12;
  ORBmatcher matcher(0.9f, true);
  const int nm = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
  dump(out + ".m12", m12.data(), m12.size());
  {  // the same match through the reference's own signature: SearchForInitialization(Frame& F1, Frame& F2, ...)
    FrameLike G1, G2;
    G1.mvKeysUn = kL; G1.mDescriptors = dL; G1.N = (int)kL.size();
    G2.mvKeysUn = kR; G2.mDescriptors = dR; G2.N = (int)kR.size();
    FrameLike::mnMaxX = (float)w; FrameLike::mnMaxY = (float)h;
    std::vector<ocv::Point2f> prev2(kL.size());
    for (size_t i = 0; i < kL.size(); i++) prev2[i] = kL[i].pt;
    std::vector<int> m12b;
    const int nm2 = matcher.SearchForInitialization(G1, G2, prev2, m12b, 100);
    if (nm2 != nm || m12b != m12 || std::memcmp(prev2.data(), prev.data(), prev.size() * sizeof(prev[0])) != 0) {
      std::printf("SearchForInitialization(Frame&, Frame&) differs from the FrameView form\n");
      return 6;
    }
    if (ORBmatcher::DescriptorDistance(dL, dL) != 0) return 7;   // static int DescriptorDistance(const cv::Mat&, const cv::Mat&)
  }
  {  // the single-call stereo path must give the same keypoints, descriptors and depths
    ORBextractor exP(nf, 1.2f, 8, 20, 7, w, h);
    std::vector<ocv::KeyPoint> pL, pR;
    ocv::Mat pdL, pdR;
    std::vector<float> puR, pdepth;
    int pmL = 0, pmR = 0;
    exP.ExtractStereo(imL, imR, pL, pdL, pR, pdR, lap, lap, pmL, pmR, 0.12f * 532.03f, 0.12f, &puR, &pdepth);
    {  // the right eye's pyramid of the single-call path (mvImagePyramidRight), level 5, and the left eye's level 0
      const ocv::Mat& v = exP.mvImagePyramidRight[5];
      std::vector<uint8_t> rows((size_t)v.rows * v.cols);
      for (int y = 0; y < v.rows; y++) std::memcpy(rows.data() + (size_t)y * v.cols, v.ptr(y), (size_t)v.cols);
      dump(out + ".vpyrR5", rows.data(), rows.size());
      const ocv::Mat& z = exP.mvImagePyramid[0];
      if (z.rows != h || z.cols != w) return 8;
      for (int y = 0; y < h; y++)
        if (std::memcmp(z.ptr(y), imL.ptr(y), (size_t)w) != 0) return 9;
    }
    dump(out + ".pkL", pL.data(), pL.size());
    dump(out + ".pdR", pdR.data, (size_t)pdR.rows * 32);
    dump(out + ".puR", puR.data(), puR.size());
    dump(out + ".pdepth", pdepth.data(), pdepth.size());
  }
  std::printf("%d %d %d %d %d\n", monoL, monoR, (int)kL.size(), (int)kR.size(), nm);
  return 0;
}
