// orbx_api_match.hip — C ABI of the matchers (include/orbx.h): brute-force 2-NN, fisheye stereo association, SearchForInitialization,
// the SearchByProjection family, SearchByBoW.  One-shot calls: host arrays in, host results out, inputs through one packed upload.
#include "orbx_host.h"

extern "C" {

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
  // a FeatureVector is a std::map: ascending node ids, and the CSR offsets must be monotone (the kernels trust them)
  for (int j = 0; j < n_kf_nodes; j++)
    if (kf_node_start[j] < 0 || kf_node_start[j] > kf_node_start[j + 1] || (j && kf_node_ids[j] <= kf_node_ids[j - 1]))
      return fail(ORBX_E_BADARG, "keyframe feature vector: node ids must ascend and offsets must be monotone");
  for (int j = 0; j < n_f_nodes; j++)
    if (f_node_start[j] < 0 || f_node_start[j] > f_node_start[j + 1] || (j && f_node_ids[j] <= f_node_ids[j - 1]))
      return fail(ORBX_E_BADARG, "frame feature vector: node ids must ascend and offsets must be monotone");
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

// SearchByBoW(KeyFrame, Frame) for the frames of an extraction batch (round 5): frame f = image first_image + f of the handle's
// last batch, whose keypoints / descriptors AND feature vector (orbx_bow_transform_batch) stay in HBM; the key frame of pair f
// comes from the host as strided arrays.  One upload, one launch per kernel for all pairs (blockIdx.y = pair), one download.
// The search has no iteration: results are those of n_frames separate orbx_search_by_bow calls.
int orbx_search_by_bow_batch(orbx_extractor* ex, int first_image, int n_frames, const uint32_t* kf_node_ids,
                             const int32_t* kf_node_start, const int32_t* n_kf_nodes, int nodes_stride,
                             const uint32_t* kf_feature_idx, const orbx_keypoint* kf_kps, const uint8_t* kf_desc,
                             const uint8_t* kf_valid, const int32_t* n_kf, int kf_stride, int n_left_f, float nnratio,
                             int check_orientation, int32_t* matches, int32_t* n_matches) {
  if (!ex || n_frames < 0 || first_image < 0 || !n_kf_nodes || !n_kf || nodes_stride < 0 || kf_stride < 0 ||
      (n_frames && (!matches || !n_matches)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n_frames == 0) return 0;
  if (ex->lastN <= 0 || first_image + n_frames > ex->lastN) return fail(ORBX_E_BADARG, "frames outside the handle's last batch");
  if (!ex->d_bowWord.p || ex->bowImages < first_image + n_frames)
    return fail(ORBX_E_BADARG, "orbx_bow_transform_batch has not run on the handle's last extraction");
  const int F = n_frames, ns = std::max(nodes_stride, 1), ks = std::max(kf_stride, 1);
  int maxNodes = 0, maxKf = 0;
  for (int f = 0; f < F; f++) {
    if (n_kf_nodes[f] < 0 || n_kf_nodes[f] > nodes_stride || n_kf[f] < 0 || n_kf[f] > kf_stride)
      return fail(ORBX_E_BADARG, "n_kf_nodes[f] / n_kf[f] outside their strides");
    maxNodes = std::max(maxNodes, n_kf_nodes[f]);
    maxKf = std::max(maxKf, n_kf[f]);
  }
  if (maxNodes && (!kf_node_ids || !kf_node_start || !kf_feature_idx || !kf_kps || !kf_desc || !kf_valid))
    return fail(ORBX_E_BADARG, "null key-frame array");
  for (int f = 0; f < F; f++) {   // the same trust boundary as the one-shot call
    const uint32_t* ids = kf_node_ids + (size_t)f * ns;
    const int32_t* stt = kf_node_start + (size_t)f * (ns + 1);
    const int nn = n_kf_nodes[f], nkl = nn ? stt[nn] : 0;
    if (nkl < 0 || nkl > n_kf[f]) return fail(ORBX_E_BADARG, "feature vector larger than the key frame");
    for (int j = 0; j < nn; j++)
      if (stt[j] < 0 || stt[j] > stt[j + 1] || (j && ids[j] <= ids[j - 1]))
        return fail(ORBX_E_BADARG, "keyframe feature vector: node ids must ascend and offsets must be monotone");
    for (int i = 0; i < nkl; i++)
      if (kf_feature_idx[(size_t)f * ks + i] >= (uint32_t)n_kf[f]) return fail(ORBX_E_BADARG, "keyframe feature index out of range");
  }
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const int cap = ex->gmax.outCap;
  std::vector<int> nF(F), cnt(3 * (size_t)F);
  HIPC(hipStreamSynchronize(ex->stream));
  HIPC(hipMemcpy(nF.data(), ex->d_nOut.p + first_image, (size_t)F * sizeof(int), hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(cnt.data(), ex->d_bowCounts.p + 3 * (size_t)first_image, 3 * (size_t)F * sizeof(int), hipMemcpyDeviceToHost));
  int maxNF = 0;
  for (int f = 0; f < F; f++) {
    nF[f] = std::min(std::max(nF[f], 0), cap);
    maxNF = std::max(maxNF, nF[f]);
  }
  Pack pk;
  std::vector<BowMatchArgs> frames(F);
  auto rows = [&](const int32_t* n, int stride) { return (size_t)(F - 1) * stride + (size_t)n[F - 1]; };  // (last pair's padding unread)
  const size_t oKn = pk.add(maxNodes ? kf_node_ids : nullptr, (size_t)F * ns * 4, rows(n_kf_nodes, ns) * 4);
  const size_t oKs = pk.add(maxNodes ? kf_node_start : nullptr, (size_t)F * (ns + 1) * 4, ((size_t)(F - 1) * (ns + 1) + n_kf_nodes[F - 1] + 1) * 4);
  const size_t oKf = pk.add(maxNodes ? kf_feature_idx : nullptr, (size_t)F * ks * 4, rows(n_kf, ks) * 4);
  const size_t oKd = pk.add(maxNodes ? kf_desc : nullptr, (size_t)F * ks * 32, rows(n_kf, ks) * 32);
  const size_t oKv = pk.add(maxNodes ? kf_valid : nullptr, (size_t)F * ks, rows(n_kf, ks));
  const size_t oKk = pk.add(maxNodes ? kf_kps : nullptr, (size_t)F * ks * sizeof(orbx_keypoint), rows(n_kf, ks) * sizeof(orbx_keypoint));
  const size_t oFr = pk.add(frames.data(), (size_t)F * sizeof(BowMatchArgs));
  const size_t oOut = pk.add(nullptr, (size_t)F * ((size_t)cap + 1) * 4);   // per pair: result, then cap matches
  const size_t oBin = pk.add(nullptr, (size_t)F * cap * 4), oFlags = pk.add(nullptr, (size_t)F * 36 * 4);
  hipError_t e = pk.reserve();
  if (e != hipSuccess) { pk.release(); return fail(ORBX_E_HIP, hipGetErrorString(e)); }
  for (int f = 0; f < F; f++) {
    BowMatchArgs a{};
    const int img = first_image + f;
    a.kfNodes = pk.ptr<uint32_t>(oKn) + (size_t)f * ns; a.kfStart = pk.ptr<int>(oKs) + (size_t)f * (ns + 1);
    a.kfFeat = pk.ptr<uint32_t>(oKf) + (size_t)f * ks; a.nKfNodes = n_kf_nodes[f];
    a.kfDesc = pk.ptr<uint32_t>(oKd) + (size_t)f * ks * 8; a.kfKps = pk.ptr<orbx_keypoint>(oKk) + (size_t)f * ks;
    a.kfValid = pk.ptr<uint8_t>(oKv) + (size_t)f * ks;
    a.fNodes = ex->d_bowNodes.p + (size_t)img * cap; a.fStart = ex->d_bowStart.p + (size_t)img * (cap + 1);
    a.fFeat = ex->d_bowFeats.p + (size_t)img * cap; a.nFNodes = cnt[3 * f + 1];
    a.fDesc = reinterpret_cast<const uint32_t*>(ex->d_desc.p + (size_t)img * cap * 32); a.fKps = ex->d_kps.p + (size_t)img * cap;
    a.nF = nF[f]; a.nLeftF = n_left_f;
    a.nnratio = nnratio; a.checkOri = check_orientation ? 1 : 0;
    a.result = pk.ptr<int>(oOut) + (size_t)f * (cap + 1); a.match = a.result + 1;
    a.bin = pk.ptr<int>(oBin) + (size_t)f * cap; a.flags = pk.ptr<int>(oFlags) + (size_t)f * 36;
    frames[f] = a;
  }
  e = pk.commit();
  if (e == hipSuccess) e = launch_bow_match_batch(pk.ptr<BowMatchArgs>(oFr), F, maxNodes, maxNF, check_orientation ? 1 : 0, nullptr);
  int total = 0;
  bool tooLarge = false;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOut, (size_t)F * ((size_t)cap + 1) * 4, &e);
    if (e == hipSuccess)
      for (int f = 0; f < F; f++) {
        const int* r = reinterpret_cast<const int*>(h) + (size_t)f * (cap + 1);
        if (r[0] < 0) tooLarge = true;
        n_matches[f] = r[0];
        std::memcpy(matches + (size_t)f * cap, r + 1, (size_t)nF[f] * 4);
        for (int i = nF[f]; i < cap; i++) matches[(size_t)f * cap + i] = -1;
        total += std::max(r[0], 0);
      }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  if (tooLarge) return fail(ORBX_E_UNSUPPORTED, "a vocabulary node holds more than 4096 frame features");
  return total;
}

namespace {
// One attempt with candidate arrays of cand_cap entries; see search_by_projection_try.
int search_for_initialization_try(const orbx_keypoint* kps1, const uint8_t* desc1, int n1, const orbx_keypoint* kps2,
                                  const uint8_t* desc2, int n2, float min_x, float min_y, float max_x, float max_y,
                                  float* prev_matched, int32_t* matches12, int window_size, float nnratio,
                                  int check_orientation, int cand_cap, int* needed) {
  *needed = 0;
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
  chk(candIdx.alloc((size_t)cand_cap));
  chk(candDist.alloc((size_t)cand_cap));
  a.candIdx = candIdx.p;
  a.candDist = candDist.p;
  a.candCap = cand_cap;
  int res[2] = {0, 0};
  if (e == hipSuccess) chk(launch_search_init(a, nullptr));
  // resolve: parallel fixed-point rounds (k_init_round), serial walk as fallback / ORBX_PROJ_SERIAL=1 cross-check
  ScratchBuf<int2> cl0, cl1, cr0, cr1, cr2;
  ScratchBuf<int> nc0, nc1, nc2, fl;
  static const bool forceSerial = getenv("ORBX_PROJ_SERIAL") && atoi(getenv("ORBX_PROJ_SERIAL")) != 0;
  bool done = false;
  int lastRound = 0;
  if (e == hipSuccess) chk(launch_search_init_cands_fill(a, nullptr));
  if (!forceSerial && n2 > 0) {
    chk(cl0.alloc(n1)); chk(cl1.alloc(n1)); chk(cr0.alloc((size_t)n2 * kFeWriters)); chk(cr1.alloc((size_t)n2 * kFeWriters));
    chk(cr2.alloc((size_t)n2 * kFeWriters)); chk(nc0.alloc(n2)); chk(nc1.alloc(n2)); chk(nc2.alloc(n2)); chk(fl.alloc(40 + 48));
    a.claim[0] = cl0.p; a.claim[1] = cl1.p; a.claimers[0] = cr0.p; a.claimers[1] = cr1.p; a.claimers[2] = cr2.p;
    a.nclaimers[0] = nc0.p; a.nclaimers[1] = nc1.p; a.nclaimers[2] = nc2.p; a.flags = fl.p;
    for (int r = 0; r < 48 && e == hipSuccess && !done; r += 4) {
      chk(launch_search_init_rounds(a, r, 4, nullptr));
      int st[40 + 48];
      st[1] = 0;
      st[40 + r + 3] = 1;
      if (e == hipSuccess) chk(hipMemcpy(st, fl.p, sizeof(st), hipMemcpyDeviceToHost));  // synchronises
      if (st[1]) break;  // an i2 collected more than kFeWriters claimers in one round
      done = st[40 + r + 3] == 0;  // the last round of the group changed nothing
      lastRound = r + 3;
    }
  }
  if (e == hipSuccess) chk(done ? launch_search_init_finish(a, lastRound, nullptr) : launch_search_init_resolve_serial(a, nullptr));
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oPrev, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      std::memcpy(res, h + (oRes - oPrev), sizeof(res));
      if (res[1] > cand_cap) {
        *needed = res[1];
      } else {
        std::memcpy(prev_matched, h, (size_t)n1 * 2 * sizeof(float));
        std::memcpy(matches12, h + (oM12 - oPrev), (size_t)n1 * sizeof(int));
      }
    }
  }
  cl0.free(); cl1.free(); cr0.free(); cr1.free(); cr2.free(); nc0.free(); nc1.free(); nc2.free(); fl.free();
  pk.release(); cellStart.free(); cellItems.free();
  candOff.free(); candIdx.free(); candDist.free(); mdist.free(); m21.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return res[0];
}
}  // namespace

int orbx_search_for_initialization(int device, const orbx_keypoint* kps1, const uint8_t* desc1, int n1,
                                   const orbx_keypoint* kps2, const uint8_t* desc2, int n2, float min_x,
                                   float min_y, float max_x, float max_y, float* prev_matched,
                                   int32_t* matches12, int window_size, float nnratio, int check_orientation) {
  if (n1 < 0 || n2 < 0 || (n1 && (!kps1 || !desc1 || !prev_matched || !matches12)) || (n2 && (!kps2 || !desc2)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n1 == 0) return 0;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  // first guess: 128 candidates per keypoint (a 100 px window over 1500 level-0 keypoints holds ~60); a denser frame repeats
  // the call once with the exact size (ORBX_PROJ_CAND_CAP forces that path in the tests)
  static const int capEnv = getenv("ORBX_PROJ_CAND_CAP") ? atoi(getenv("ORBX_PROJ_CAND_CAP")) : 0;
  int cap = capEnv > 0 ? capEnv : n1 * 128, needed = 0;
  rc = search_for_initialization_try(kps1, desc1, n1, kps2, desc2, n2, min_x, min_y, max_x, max_y, prev_matched, matches12,
                                     window_size, nnratio, check_orientation, cap, &needed);
  if (rc >= 0 && needed > cap)
    rc = search_for_initialization_try(kps1, desc1, n1, kps2, desc2, n2, min_x, min_y, max_x, max_y, prev_matched, matches12,
                                       window_size, nnratio, check_orientation, needed, &needed);
  return rc;
}

// SearchForInitialization on the frames of an extraction batch (round 5; the batched form of the call above, like
// proj_batch_impl for the projection matchers): F2 of pair f = image first_image + f of the handle's last batch -- keypoints
// and descriptors stay in HBM --, F1 of pair f = kps1 / desc1 [f * stride .. + n1[f]) from the host (the initial frame of that
// camera).  One upload, every kernel launched ONCE for all pairs (launch_search_init_batch), kInitBlind fixed-point rounds
// enqueued without reading a convergence flag, one download.  A pair whose candidate lists overflowed the first guess, whose
// claimer lists overflowed or whose claims did not settle within the blind rounds is redone through the one-shot call.
int orbx_search_for_initialization_batch(orbx_extractor* ex, int first_image, int n_frames, const orbx_keypoint* kps1,
                                         const uint8_t* desc1, const int32_t* n1, int stride, float min_x, float min_y,
                                         float max_x, float max_y, float* prev_matched, int32_t* matches12, int window_size,
                                         float nnratio, int check_orientation, int32_t* n_matches) {
  if (!ex || n_frames < 0 || first_image < 0 || !n1 || stride < 0 || (n_frames && (!n_matches)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n_frames == 0) return 0;
  if (ex->lastN <= 0 || first_image + n_frames > ex->lastN) return fail(ORBX_E_BADARG, "frames outside the handle's last batch");
  int maxN1 = 0;
  for (int f = 0; f < n_frames; f++) {
    if (n1[f] < 0 || n1[f] > stride) return fail(ORBX_E_BADARG, "n1[f] outside [0, stride]");
    maxN1 = std::max(maxN1, n1[f]);
  }
  if (maxN1 && (!kps1 || !desc1 || !prev_matched || !matches12)) return fail(ORBX_E_BADARG, "null argument");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const int cap = ex->gmax.outCap, F = n_frames, st = std::max(stride, 1);
  std::vector<int> n2(F);
  HIPC(hipStreamSynchronize(ex->stream));
  HIPC(hipMemcpy(n2.data(), ex->d_nOut.p + first_image, (size_t)F * sizeof(int), hipMemcpyDeviceToHost));
  int maxN2 = 0;
  for (int f = 0; f < F; f++) {
    n2[f] = std::min(std::max(n2[f], 0), cap);
    maxN2 = std::max(maxN2, n2[f]);
  }
  static const int capEnv = getenv("ORBX_PROJ_CAND_CAP") ? atoi(getenv("ORBX_PROJ_CAND_CAP")) : 0;
  static const int kBlind = getenv("ORBX_PROJ_BLIND") ? std::min(48, std::max(1, atoi(getenv("ORBX_PROJ_BLIND")))) : 16;
  const int nm = std::max(maxN1, 1), candCap = capEnv > 0 ? capEnv : nm * 128;
  constexpr int kFlags = 40 + 48;
  Pack pk;
  std::vector<InitArgs> frames(F);
  const size_t rows = (size_t)(F - 1) * st + (size_t)n1[F - 1];   // (the last pair's padding is not read: see proj_batch_impl)
  const size_t oK1 = pk.add(maxN1 ? kps1 : nullptr, (size_t)F * st * sizeof(orbx_keypoint), rows * sizeof(orbx_keypoint));
  const size_t oD1 = pk.add(maxN1 ? desc1 : nullptr, (size_t)F * st * 32, rows * 32);
  const size_t oFr = pk.add(frames.data(), (size_t)F * sizeof(InitArgs));
  const size_t oPrev = pk.add(maxN1 ? prev_matched : nullptr, (size_t)F * st * 2 * sizeof(float), rows * 2 * sizeof(float));
  const size_t oM12 = pk.add(nullptr, (size_t)F * st * sizeof(int)), oRes = pk.add(nullptr, (size_t)F * 2 * sizeof(int));
  const size_t oFlags = pk.add(nullptr, (size_t)F * kFlags * sizeof(int));
  const size_t outBytes = oFlags + (size_t)F * kFlags * sizeof(int) - oPrev;
  auto per = [&](size_t ints) { return pk.add(nullptr, (size_t)F * ints * sizeof(int)); };
  const size_t cs = 64 * 48 + 4, oCs = per(cs), oCi = per(cap), oMd = per(cap), oM21 = per(cap);
  const size_t oCo = per((size_t)nm + 4), oCx = per(candCap), oCd = per(candCap);
  const size_t oCl0 = per(2 * (size_t)nm), oCl1 = per(2 * (size_t)nm);
  const size_t oCr0 = per(2 * (size_t)cap * kFeWriters), oCr1 = per(2 * (size_t)cap * kFeWriters), oCr2 = per(2 * (size_t)cap * kFeWriters);
  const size_t oNc0 = per(cap), oNc1 = per(cap), oNc2 = per(cap);
  hipError_t e = pk.reserve();
  if (e != hipSuccess) { pk.release(); return fail(ORBX_E_HIP, hipGetErrorString(e)); }
  for (int f = 0; f < F; f++) {
    InitArgs a{};
    const int img = first_image + f;
    a.k1 = pk.ptr<orbx_keypoint>(oK1) + (size_t)f * st;
    a.d1 = pk.ptr<uint8_t>(oD1) + (size_t)f * st * 32;
    a.k2 = ex->d_kps.p + (size_t)img * cap;
    a.d2 = ex->d_desc.p + (size_t)img * cap * 32;
    a.n1 = n1[f]; a.n2 = n2[f];
    a.minX = min_x; a.minY = min_y;
    a.invW = 64.f / (max_x - min_x);
    a.invH = 48.f / (max_y - min_y);
    a.prev = pk.ptr<float>(oPrev) + (size_t)f * st * 2;
    a.matches12 = pk.ptr<int>(oM12) + (size_t)f * st;
    a.window = window_size; a.nnratio = nnratio; a.checkOri = check_orientation;
    a.cellStart = pk.ptr<int>(oCs) + (size_t)f * cs;
    a.cellItems = pk.ptr<int>(oCi) + (size_t)f * cap;
    a.candOff = pk.ptr<int>(oCo) + (size_t)f * (nm + 4);
    a.candIdx = pk.ptr<int>(oCx) + (size_t)f * candCap;
    a.candDist = pk.ptr<int>(oCd) + (size_t)f * candCap;
    a.candCap = candCap;
    a.matchedDist = pk.ptr<int>(oMd) + (size_t)f * cap;
    a.matches21 = pk.ptr<int>(oM21) + (size_t)f * cap;
    a.result = pk.ptr<int>(oRes) + (size_t)f * 2;
    a.claim[0] = pk.ptr<int2>(oCl0) + (size_t)f * nm; a.claim[1] = pk.ptr<int2>(oCl1) + (size_t)f * nm;
    a.claimers[0] = pk.ptr<int2>(oCr0) + (size_t)f * cap * kFeWriters; a.claimers[1] = pk.ptr<int2>(oCr1) + (size_t)f * cap * kFeWriters;
    a.claimers[2] = pk.ptr<int2>(oCr2) + (size_t)f * cap * kFeWriters;
    a.nclaimers[0] = pk.ptr<int>(oNc0) + (size_t)f * cap; a.nclaimers[1] = pk.ptr<int>(oNc1) + (size_t)f * cap;
    a.nclaimers[2] = pk.ptr<int>(oNc2) + (size_t)f * cap;
    a.flags = pk.ptr<int>(oFlags) + (size_t)f * kFlags;
    frames[f] = a;
  }
  e = pk.commit();
  if (e == hipSuccess) e = launch_search_init_batch(pk.ptr<InitArgs>(oFr), F, maxN1, maxN2, kBlind, nullptr);
  std::vector<int> redo;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oPrev, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      for (int f = 0; f < F; f++) {
        int res[2];
        std::memcpy(res, h + (oRes - oPrev) + (size_t)f * 2 * sizeof(int), sizeof res);
        const int* fl = reinterpret_cast<const int*>(h + (oFlags - oPrev)) + (size_t)f * kFlags;
        if (n1[f] > 0 && (res[1] > candCap || fl[1] || fl[40 + kBlind - 1] || n2[f] == 0)) {
          redo.push_back(f);
          continue;
        }
        if (n1[f] > 0) {
          std::memcpy(prev_matched + (size_t)f * st * 2, h + (size_t)f * st * 2 * sizeof(float), (size_t)n1[f] * 2 * sizeof(float));
          std::memcpy(matches12 + (size_t)f * st, h + (oM12 - oPrev) + (size_t)f * st * sizeof(int), (size_t)n1[f] * sizeof(int));
        }
        n_matches[f] = n1[f] > 0 ? res[0] : 0;
      }
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  for (int f : redo) {   // the one-shot call on a host copy of F2 (prev_matched / matches12 of this pair are still the caller's input)
    const int img = first_image + f, n = n2[f];
    std::vector<orbx_keypoint> k(std::max(n, 1));
    std::vector<uint8_t> d((size_t)std::max(n, 1) * 32);
    HIPC(hipMemcpy(k.data(), ex->d_kps.p + (size_t)img * cap, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(d.data(), ex->d_desc.p + (size_t)img * cap * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    rc = orbx_search_for_initialization(ex->device, kps1 + (size_t)f * st, desc1 + (size_t)f * st * 32, n1[f], k.data(), d.data(), n,
                                        min_x, min_y, max_x, max_y, prev_matched + (size_t)f * st * 2, matches12 + (size_t)f * st,
                                        window_size, nnratio, check_orientation);
    if (rc < 0) return rc;
    n_matches[f] = rc;
  }
  int total = 0;
  for (int f = 0; f < F; f++) total += n_matches[f];
  return total;
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
// One attempt with candidate arrays of cand_cap entries (all candidate lists together).  Nothing synchronises between the
// upload and the convergence check of the rounds: the lists are counted, scanned and filled back to back; when they did not
// fit, the kernels worked on truncated lists, *needed reports the size and the outputs are left untouched.
int search_by_projection_try(const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right, int n,
                             float min_x, float min_y, float max_x, float max_y, const float* scale_factors, int nlevels,
                             const orbx_map_point_view* map_points, const orbx_projected_point* points, int n_points,
                             float th, int far_points, float th_far_points, float nnratio, int check_ori,
                             uint8_t* occupied, int32_t* match, int cand_cap, int* needed, bool serial, bool* converged,
                             int max_dist, int claim_all) {
  const int mode = points ? 1 : 0;
  *needed = 0;
  *converged = true;
  ScratchBuf<int> cellStart, cellItems, candOff, candIdx, candDist, mdist, m21, m12, taker0, taker1, taker2, choice;
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
  const size_t oOcc = pk.add(occupied, n), oRes = pk.add(nullptr, 2 * sizeof(int));
  const size_t oFlags = pk.add(nullptr, (40 + 48) * sizeof(int)), oMt = pk.add(nullptr, (size_t)n * sizeof(int));
  const size_t outBytes = oMt + (size_t)n * sizeof(int) - oOcc;
  chk(pk.commit());
  struct { orbx_keypoint* p; } k{pk.ptr<orbx_keypoint>(oK)};
  struct { uint8_t* p; } d{pk.ptr<uint8_t>(oD)}, occ{pk.ptr<uint8_t>(oOcc)};
  struct { float* p; } ur{pk.ptr<float>(oUr)}, sf{pk.ptr<float>(oSf)};
  struct { orbx_map_point_view* p; } mp{pk.ptr<orbx_map_point_view>(oMp)};
  struct { orbx_projected_point* p; } pp{pk.ptr<orbx_projected_point>(oPp)};
  struct { int* p; } mt{pk.ptr<int>(oMt)}, result{pk.ptr<int>(oRes)}, flags{pk.ptr<int>(oFlags)};
  chk(taker0.alloc(n)); chk(taker1.alloc(n)); chk(taker2.alloc(n)); chk(choice.alloc(nm));
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
  a.mode = mode; a.checkOri = check_ori; a.maxDist = max_dist; a.claimAll = claim_all;
  a.th = th; a.thFar = th_far_points; a.nnratio = nnratio; a.far = far_points;
  a.occupied = occ.p; a.match = mt.p; a.candOff = candOff.p; a.result = result.p;
  chk(candIdx.alloc((size_t)cand_cap));
  chk(candDist.alloc((size_t)cand_cap));
  a.candIdx = candIdx.p; a.candDist = candDist.p; a.candCap = cand_cap;
  int res[2] = {0, 0};
  if (e == hipSuccess) chk(launch_proj_count(a, nullptr));
  a.taker[0] = taker0.p; a.taker[1] = taker1.p; a.taker[2] = taker2.p; a.choice = choice.p; a.flags = flags.p;
  // Resolve: kProjBlindRounds (16; ORBX_PROJ_BLIND) rounds of the parallel fixed-point iteration (k_proj_round) are enqueued without looking --
  // rounds after the fixed point return at once -- followed by the finish kernels; whether the last round still changed
  // something comes back with the results.  Then (pathological claim chains) the caller repeats the attempt with the
  // one-wave serial walk, which ORBX_PROJ_SERIAL=1 also forces (tests).
  static const int kProjBlindRounds = getenv("ORBX_PROJ_BLIND") ? std::min(48, std::max(1, atoi(getenv("ORBX_PROJ_BLIND")))) : 16;
  if (e == hipSuccess) chk(launch_proj_cands_fill(a, nullptr));
  const bool rounds = !serial && n_points > 0;
  if (e == hipSuccess && rounds) chk(launch_proj_rounds(a, 0, kProjBlindRounds, nullptr));
  if (e == hipSuccess) chk(rounds ? launch_proj_finish(a, 0, nullptr) : launch_proj_resolve_serial(a, nullptr));
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOcc, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      std::memcpy(res, h + (oRes - oOcc), sizeof(res));
      int lastChanged = 0;
      if (rounds) std::memcpy(&lastChanged, h + (oFlags - oOcc) + (40 + kProjBlindRounds - 1) * sizeof(int), sizeof(int));
      if (res[1] > cand_cap) {
        *needed = res[1];
      } else if (lastChanged) {
        *converged = false;
      } else {
        std::memcpy(occupied, h, n);
        std::memcpy(match, h + (oMt - oOcc), (size_t)n * sizeof(int));
      }
    }
  }
  pk.release(); cellStart.free(); cellItems.free();
  candOff.free(); candIdx.free(); candDist.free(); mdist.free(); m21.free(); m12.free();
  taker0.free(); taker1.free(); taker2.free(); choice.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return res[0];
}

int search_by_projection_impl(int device, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right, int n,
                              float min_x, float min_y, float max_x, float max_y, const float* scale_factors, int nlevels,
                              const orbx_map_point_view* map_points, const orbx_projected_point* points, int n_points,
                              float th, int far_points, float th_far_points, float nnratio, int check_ori,
                              uint8_t* occupied, int32_t* match, int max_dist = 100 /* TH_HIGH */, int claim_all = 0) {
  if (n == 0) return 0;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  // room for 96 candidates per point on average (a search window holds 10-40); denser inputs repeat the call once with the
  // exact size.  ORBX_PROJ_CAND_CAP overrides the first guess (tests force the second attempt with it).
  static const int capEnv = getenv("ORBX_PROJ_CAND_CAP") ? atoi(getenv("ORBX_PROJ_CAND_CAP")) : 0;
  static const bool forceSerial = getenv("ORBX_PROJ_SERIAL") && atoi(getenv("ORBX_PROJ_SERIAL")) != 0;
  int cap = capEnv > 0 ? capEnv : std::max(n_points, 1) * 96, needed = 0;
  bool serial = forceSerial, converged = true;
  for (int attempt = 0; attempt < 3; attempt++) {  // at most: capacity retry, then serial retry
    rc = search_by_projection_try(kps_un, desc, u_right, n, min_x, min_y, max_x, max_y, scale_factors, nlevels, map_points, points,
                                  n_points, th, far_points, th_far_points, nnratio, check_ori, occupied, match, cap, &needed,
                                  serial, &converged, max_dist, claim_all);
    if (rc < 0) break;
    if (needed > cap) { cap = needed; continue; }
    if (!converged) {
      if (trace_slow_ms() > 0) std::fprintf(stderr, "ORBX_TRACE_SLOW SearchByProjection: no fixed point within the blind rounds, serial walk\n");
      serial = true;
      continue;
    }
    break;
  }
  return rc;
}
}  // namespace

int orbx_search_by_projection(int device, const orbx_keypoint* kps_un, const uint8_t* desc, const float* u_right,
                              int n, float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                              int nlevels, const orbx_map_point_view* map_points, int n_map_points, float th,
                              int far_points, float th_far_points, float nnratio, uint8_t* occupied, int32_t* match) {
  if (n < 0 || n_map_points < 0 || nlevels < 1 || !scale_factors || (n && (!kps_un || !desc || !occupied || !match)) ||
      (n_map_points && !map_points))
    return fail(ORBX_E_BADARG, "bad argument");
  // mnTrackScaleLevel is only read for points with mbTrackInView set (src/ORBmatcher.cc:58-61) and is uninitialised in
  // MapPoint's constructors otherwise: validate it only there (the kernels never read it for the other points)
  for (int i = 0; i < n_map_points; i++)
    if (map_points[i].in_view && !map_points[i].bad &&
        (map_points[i].predicted_level < 0 || map_points[i].predicted_level >= nlevels))
      return fail(ORBX_E_BADARG, "in-view map point with a predicted level outside [0, nlevels)");
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
// The pinhole SearchByProjection flavours on the frames of an extraction batch: keypoints and
// descriptors of frame f are image first_image + f of the handle's last batch and never leave HBM; only the points (and the
// occupancy flags) are uploaded, in one copy; every kernel of the chain is launched ONCE for all frames (blockIdx.y = frame,
// launch_proj_batch) with kProjBlindRounds fixed-point rounds enqueued without looking; match / occupied / counts come back in
// one copy.  A frame whose candidate lists overflowed the first guess or whose claims did not settle within the blind rounds
// (pathological chains) is redone through the one-shot path -- same result by construction, checked by the tests.
int proj_batch_impl(orbx_extractor* ex, int first_image, int n_frames, float min_x, float min_y, float max_x, float max_y, int mode,
                    const orbx_map_point_view* mps, const orbx_projected_point* pts, const int32_t* n_points, int stride, float th,
                    int far_points, float th_far, float nnratio, int check_ori, int stereo_pair0, const uint8_t* occupied_in,
                    uint8_t* occupied, int32_t* match, int32_t* n_matches) {
  if (!ex || n_frames < 0 || first_image < 0 || !n_points || stride < 0 || (n_frames && (!occupied || !match || !n_matches)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n_frames == 0) return 0;
  if (ex->lastN <= 0 || first_image + n_frames > ex->lastN) return fail(ORBX_E_BADARG, "frames outside the handle's last batch");
  if (stereo_pair0 >= 0 && stereo_pair0 + n_frames > ex->lastStereoPairs)
    return fail(ORBX_E_BADARG, "u_right requested but the handle's last stereo association does not cover these frames");
  int maxPts = 0;
  for (int f = 0; f < n_frames; f++) {
    if (n_points[f] < 0 || n_points[f] > stride) return fail(ORBX_E_BADARG, "n_points[f] outside [0, points_stride]");
    if (n_points[f] > 15000) return fail(ORBX_E_CAPACITY, "more than 15000 points in a frame");
    maxPts = std::max(maxPts, n_points[f]);
  }
  const bool devViewsP = mode == 1 && !pts && maxPts > 0;  // views made on the device by orbx_project_last_frames_batch
  if (devViewsP) {
    if (ex->pviewsFrames < n_frames || ex->lfStride != stride)
      return fail(ORBX_E_BADARG, "points == NULL needs a preceding orbx_project_last_frames_batch with the same frames and points_stride");
    for (int f = 0; f < n_frames; f++)
      if (n_points[f] != ex->lfN[f]) return fail(ORBX_E_BADARG, "points == NULL: n_points[f] must equal the uploaded LastFrame's");
  }
  const bool devViews = mode == 0 && !mps && maxPts > 0;   // views made on the device by orbx_project_map_points_batch
  if (devViews) {
    if (ex->viewsFrames < n_frames || ex->viewsStride != stride) return fail(ORBX_E_BADARG, "map_points == NULL needs a preceding orbx_project_map_points_batch with the same frames and points_stride == its n");
    for (int f = 0; f < n_frames; f++)
      if (n_points[f] != ex->viewsStride) return fail(ORBX_E_BADARG, "map_points == NULL: n_map_points[f] must equal the uploaded map's n");
  }
  if (maxPts && !devViews && !devViewsP && (mode == 0 ? !mps : !pts)) return fail(ORBX_E_BADARG, "null points");
  const int nlevels = ex->prm.nlevels;
  if (mode == 0 && !devViews)
    for (int f = 0; f < n_frames; f++)
      for (int i = 0; i < n_points[f]; i++) {
        const orbx_map_point_view& m = mps[(size_t)f * stride + i];
        if (m.in_view && !m.bad && (m.predicted_level < 0 || m.predicted_level >= nlevels))
          return fail(ORBX_E_BADARG, "in-view map point with a predicted level outside [0, nlevels)");
      }
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const int cap = ex->gmax.outCap, F = n_frames;
  // the frames' keypoint counts live on the device: one small copy behind the extraction
  std::vector<int> n2(F);
  HIPC(hipStreamSynchronize(ex->stream));
  HIPC(hipMemcpy(n2.data(), ex->d_nOut.p + first_image, (size_t)F * sizeof(int), hipMemcpyDeviceToHost));
  int maxN2 = 0;
  for (int f = 0; f < F; f++) {
    n2[f] = std::min(std::max(n2[f], 0), cap);
    maxN2 = std::max(maxN2, n2[f]);
  }
  static const int capEnv = getenv("ORBX_PROJ_CAND_CAP") ? atoi(getenv("ORBX_PROJ_CAND_CAP")) : 0;
  static const int kBlind = getenv("ORBX_PROJ_BLIND") ? std::min(48, std::max(1, atoi(getenv("ORBX_PROJ_BLIND")))) : 16;
  const int nm = std::max(maxPts, 1), candCap = capEnv > 0 ? capEnv : nm * 96;
  const size_t ptBytes = mode == 0 ? sizeof(orbx_map_point_view) : sizeof(orbx_projected_point);
  constexpr int kFlags = 40 + 48;
  // ---- one device block: inputs | outputs (occupied, match, result, flags: one copy back) | scratch
  Pack pk;
  std::vector<ProjArgs> frames(F);
  std::vector<uint8_t> occ0;
  if (!occupied_in) occ0.assign((size_t)F * cap, 0);
  // (the header only promises points[f * stride .. f * stride + n_points[f]) of every frame: the last frame's padding is not read)
  const size_t ptsRead = ((size_t)(F - 1) * std::max(stride, 1) + (size_t)std::max(n_points[F - 1], 0)) * ptBytes;
  const size_t oPts = (devViews || devViewsP) ? pk.add(nullptr, 16)
                               : pk.add(maxPts ? (mode == 0 ? (const void*)mps : (const void*)pts) : nullptr,
                                        (size_t)F * std::max(stride, 1) * ptBytes, ptsRead);
  const size_t oSf = pk.add(ex->scale.data(), (size_t)nlevels * sizeof(float));
  const size_t oFr = pk.add(frames.data(), (size_t)F * sizeof(ProjArgs));
  const size_t oOcc = pk.add(occupied_in ? occupied_in : occ0.data(), (size_t)F * cap);
  const size_t oMt = pk.add(nullptr, (size_t)F * cap * sizeof(int)), oRes = pk.add(nullptr, (size_t)F * 2 * sizeof(int));
  const size_t oFlags = pk.add(nullptr, (size_t)F * kFlags * sizeof(int));
  const size_t outBytes = oFlags + (size_t)F * kFlags * sizeof(int) - oOcc;
  auto per = [&](size_t ints) { return pk.add(nullptr, (size_t)F * ints * sizeof(int)); };
  const size_t cs = 64 * 48 + 4, oCs = per(cs), oCi = per(cap), oMd = per(cap), oM21 = per(cap), oM12 = per(4);
  const size_t oCo = per((size_t)nm + 4), oCx = per(candCap), oCd = per(candCap);
  const size_t oT0 = per(cap), oT1 = per(cap), oT2 = per(cap), oCh = per(nm);
  hipError_t e = pk.reserve();
  if (e != hipSuccess) { pk.release(); return fail(ORBX_E_HIP, hipGetErrorString(e)); }
  for (int f = 0; f < F; f++) {
    ProjArgs a{};
    const int img = first_image + f;
    a.grid.k2 = ex->d_kps.p + (size_t)img * cap;
    a.grid.n2 = n2[f];
    a.grid.n1 = 0;
    a.grid.minX = min_x; a.grid.minY = min_y;
    a.grid.invW = 64.f / (max_x - min_x);
    a.grid.invH = 48.f / (max_y - min_y);
    a.grid.cellStart = pk.ptr<int>(oCs) + (size_t)f * cs;
    a.grid.cellItems = pk.ptr<int>(oCi) + (size_t)f * cap;
    a.grid.matchedDist = pk.ptr<int>(oMd) + (size_t)f * cap;
    a.grid.matches21 = pk.ptr<int>(oM21) + (size_t)f * cap;
    a.grid.matches12 = pk.ptr<int>(oM12) + (size_t)f * 4;
    a.grid.result = pk.ptr<int>(oRes) + (size_t)f * 2;
    a.grid.candOff = pk.ptr<int>(oCo) + (size_t)f * (nm + 4);
    a.grid.candCap = 1 << 30;
    a.desc = ex->d_desc.p + (size_t)img * cap * 32;
    a.uRight = stereo_pair0 >= 0 ? ex->d_uR.p + (size_t)(stereo_pair0 + f) * cap : nullptr;
    a.scale = pk.ptr<float>(oSf);
    a.mps = mode == 0 ? (devViews ? ex->d_views.p : pk.ptr<orbx_map_point_view>(oPts)) + (size_t)f * stride : nullptr;
    a.pts = mode == 1 ? (devViewsP ? ex->d_pviews.p : pk.ptr<orbx_projected_point>(oPts)) + (size_t)f * stride : nullptr;
    a.nmp = n_points[f];
    a.mode = mode; a.checkOri = check_ori; a.maxDist = 100 /* TH_HIGH */; a.claimAll = 0;
    a.th = th; a.thFar = th_far; a.nnratio = nnratio; a.far = far_points;
    a.occupied = pk.ptr<uint8_t>(oOcc) + (size_t)f * cap;
    a.match = pk.ptr<int>(oMt) + (size_t)f * cap;
    a.candOff = a.grid.candOff;
    a.candIdx = pk.ptr<int>(oCx) + (size_t)f * candCap;
    a.candDist = pk.ptr<int>(oCd) + (size_t)f * candCap;
    a.candCap = candCap;
    a.result = a.grid.result;
    a.taker[0] = pk.ptr<int>(oT0) + (size_t)f * cap; a.taker[1] = pk.ptr<int>(oT1) + (size_t)f * cap;
    a.taker[2] = pk.ptr<int>(oT2) + (size_t)f * cap;
    a.choice = pk.ptr<int>(oCh) + (size_t)f * nm;
    a.flags = pk.ptr<int>(oFlags) + (size_t)f * kFlags;
    frames[f] = a;
  }
  e = pk.commit();  // points, scale factors, the argument blocks (now complete) and the occupancy flags: one upload
  if (e == hipSuccess) e = launch_proj_batch(pk.ptr<ProjArgs>(oFr), F, maxPts, maxN2, mode, check_ori, kBlind, nullptr);
  std::vector<int> redo;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOcc, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      for (int f = 0; f < F; f++) {
        int res[2], lastChanged = 0;
        std::memcpy(res, h + (oRes - oOcc) + (size_t)f * 2 * sizeof(int), sizeof res);
        std::memcpy(&lastChanged, h + (oFlags - oOcc) + ((size_t)f * kFlags + 40 + kBlind - 1) * sizeof(int), sizeof(int));
        if (n_points[f] > 0 && (res[1] > candCap || lastChanged)) {
          redo.push_back(f);
          continue;
        }
        std::memcpy(occupied + (size_t)f * cap, h + (size_t)f * cap, cap);
        std::memcpy(match + (size_t)f * cap, h + (oMt - oOcc) + (size_t)f * cap * sizeof(int), (size_t)cap * sizeof(int));
        for (int i = n2[f]; i < cap; i++) match[(size_t)f * cap + i] = -1;   // rows past the frame's keypoints
        n_matches[f] = res[0];
      }
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  // ---- the rare frames the blind batch could not finish: the one-shot path on a host copy of the frame
  for (int f : redo) {
    const int img = first_image + f, n = n2[f];
    std::vector<orbx_keypoint> k(std::max(n, 1));
    std::vector<uint8_t> d((size_t)std::max(n, 1) * 32);
    std::vector<float> ur(std::max(n, 1));
    HIPC(hipMemcpy(k.data(), ex->d_kps.p + (size_t)img * cap, (size_t)n * sizeof(orbx_keypoint), hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(d.data(), ex->d_desc.p + (size_t)img * cap * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    if (stereo_pair0 >= 0)
      HIPC(hipMemcpy(ur.data(), ex->d_uR.p + (size_t)(stereo_pair0 + f) * cap, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    uint8_t* occ = occupied + (size_t)f * cap;
    int32_t* mt = match + (size_t)f * cap;
    if (occupied_in) std::memcpy(occ, occupied_in + (size_t)f * cap, cap); else std::memset(occ, 0, cap);
    for (int i = 0; i < cap; i++) mt[i] = -1;
    std::vector<orbx_map_point_view> hv;
    std::vector<orbx_projected_point> hp;
    if (devViews) {   // (the one-shot path takes host arrays)
      hv.resize((size_t)stride);
      HIPC(hipMemcpy(hv.data(), ex->d_views.p + (size_t)f * stride, (size_t)stride * sizeof(orbx_map_point_view), hipMemcpyDeviceToHost));
    }
    if (devViewsP) {
      hp.resize((size_t)stride);
      HIPC(hipMemcpy(hp.data(), ex->d_pviews.p + (size_t)f * stride, (size_t)stride * sizeof(orbx_projected_point), hipMemcpyDeviceToHost));
    }
    rc = search_by_projection_impl(ex->device, k.data(), d.data(), stereo_pair0 >= 0 ? ur.data() : nullptr, n, min_x, min_y, max_x,
                                   max_y, ex->scale.data(), nlevels, mode == 0 ? (devViews ? hv.data() : mps + (size_t)f * stride) : nullptr,
                                   mode == 1 ? (devViewsP ? hp.data() : pts + (size_t)f * stride) : nullptr, n_points[f], th, far_points, th_far, nnratio,
                                   check_ori, occ, mt);
    if (rc < 0) return rc;
    n_matches[f] = rc;
  }
  int total = 0;
  for (int f = 0; f < F; f++) total += n_matches[f];
  return total;
}
}  // namespace

int orbx_map_upload(orbx_extractor* ex, int n, const float* world_pos, const float* normal, const float* min_distance,
                    const float* max_distance, const uint8_t* desc, const uint8_t* flags) {
  if (!ex || n < 0 || n > 15000 || (n && (!world_pos || !normal || !min_distance || !max_distance || !desc || !flags)))
    return fail(ORBX_E_BADARG, "bad argument (at most 15000 points)");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  const size_t m = (size_t)std::max(n, 1);
  HIPC(hipStreamSynchronize(ex->stream));   // (a projection of the previous map may still read the arrays)
  hipError_t e = ex->d_mapPos.grow(3 * m);
  if (e == hipSuccess) e = ex->d_mapNormal.grow(3 * m);
  if (e == hipSuccess) e = ex->d_mapMinD.grow(m);
  if (e == hipSuccess) e = ex->d_mapMaxD.grow(m);
  if (e == hipSuccess) e = ex->d_mapDesc.grow(32 * m);
  if (e == hipSuccess) e = ex->d_mapFlags.grow(m);
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  ex->mapN = 0;
  ex->viewsFrames = 0;
  ex->fviewsFrames = 0;
  if (n) {
    HIPC(hipMemcpy(ex->d_mapPos.p, world_pos, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_mapNormal.p, normal, (size_t)n * 12, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_mapMinD.p, min_distance, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_mapMaxD.p, max_distance, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_mapDesc.p, desc, (size_t)n * 32, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_mapFlags.p, flags, (size_t)n, hipMemcpyHostToDevice));
  }
  ex->mapN = n;
  return ORBX_OK;
}

int orbx_project_map_points_batch(orbx_extractor* ex, int n_frames, const orbx_frame_pose* poses, float min_x, float min_y,
                                  float max_x, float max_y, float viewing_cos_limit, const uint8_t* skip,
                                  orbx_map_point_view* views_out) {
  if (!ex || n_frames < 0 || n_frames > 4096 || (n_frames && !poses)) return fail(ORBX_E_BADARG, "bad argument");
  if (ex->mapN <= 0) return fail(ORBX_E_BADARG, "no map uploaded (orbx_map_upload)");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  ex->viewsFrames = 0;
  if (n_frames == 0) return ORBX_OK;
  const int n = ex->mapN;
  hipError_t e = ex->d_poses.grow((size_t)n_frames);
  if (e == hipSuccess) e = ex->d_views.grow((size_t)n_frames * n);
  if (e == hipSuccess && skip) e = ex->d_mapSkip.grow((size_t)n_frames * n);
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  HIPC(hipMemcpyAsync(ex->d_poses.p, poses, (size_t)n_frames * sizeof(orbx_frame_pose), hipMemcpyHostToDevice, ex->stream));
  if (skip) HIPC(hipMemcpyAsync(ex->d_mapSkip.p, skip, (size_t)n_frames * n, hipMemcpyHostToDevice, ex->stream));
  MapProjArgs a{};
  a.pos = ex->d_mapPos.p; a.normal = ex->d_mapNormal.p; a.minDist = ex->d_mapMinD.p; a.maxDist = ex->d_mapMaxD.p;
  a.desc = ex->d_mapDesc.p; a.flags = ex->d_mapFlags.p; a.skip = skip ? ex->d_mapSkip.p : nullptr;
  a.poses = ex->d_poses.p; a.views = ex->d_views.p;
  a.n = n; a.nlevels = ex->prm.nlevels;
  a.minX = min_x; a.minY = min_y; a.maxX = max_x; a.maxY = max_y; a.viewCosLimit = viewing_cos_limit;
  a.logScaleFactor = logf(ex->prm.scale_factor);   // Frame::mfLogScaleFactor = log(mfScaleFactor)
  HIPC(launch_project_map(a, n_frames, ex->stream));
  if (views_out) {
    HIPC(hipMemcpyAsync(views_out, ex->d_views.p, (size_t)n_frames * n * sizeof(orbx_map_point_view), hipMemcpyDeviceToHost, ex->stream));
    HIPC(hipStreamSynchronize(ex->stream));   // (the poses / skip flags of a pageable caller array have been read as well)
  } else {
    HIPC(hipStreamSynchronize(ex->stream));   // pageable host sources: the call returns when they have been consumed
  }
  ex->viewsFrames = n_frames;
  ex->viewsStride = n;
  return ORBX_OK;
}

int orbx_project_map_points_fisheye_batch(orbx_extractor* ex, int n_frames, const orbx_frame_pose_kb8* left_poses,
                                          const orbx_frame_pose_kb8* right_poses, float min_x, float min_y, float max_x, float max_y,
                                          float viewing_cos_limit, const uint8_t* skip, orbx_map_point_view* views_out,
                                          orbx_map_point_right* views_right_out) {
  if (!ex || n_frames < 0 || n_frames > 4096 || (n_frames && (!left_poses || !right_poses))) return fail(ORBX_E_BADARG, "bad argument");
  if (ex->mapN <= 0) return fail(ORBX_E_BADARG, "no map uploaded (orbx_map_upload)");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  ex->fviewsFrames = 0;
  if (n_frames == 0) return ORBX_OK;
  const int n = ex->mapN;
  hipError_t e = ex->d_posesK.grow(2 * (size_t)n_frames);
  if (e == hipSuccess) e = ex->d_fviewsL.grow((size_t)n_frames * n);
  if (e == hipSuccess) e = ex->d_fviewsR.grow((size_t)n_frames * n);
  if (e == hipSuccess && skip) e = ex->d_mapSkip.grow((size_t)n_frames * n);
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  HIPC(hipMemcpyAsync(ex->d_posesK.p, left_poses, (size_t)n_frames * sizeof(orbx_frame_pose_kb8), hipMemcpyHostToDevice, ex->stream));
  HIPC(hipMemcpyAsync(ex->d_posesK.p + n_frames, right_poses, (size_t)n_frames * sizeof(orbx_frame_pose_kb8), hipMemcpyHostToDevice, ex->stream));
  if (skip) HIPC(hipMemcpyAsync(ex->d_mapSkip.p, skip, (size_t)n_frames * n, hipMemcpyHostToDevice, ex->stream));
  MapProjKb8Args a{};
  a.pos = ex->d_mapPos.p; a.normal = ex->d_mapNormal.p; a.minDist = ex->d_mapMinD.p; a.maxDist = ex->d_mapMaxD.p;
  a.desc = ex->d_mapDesc.p; a.flags = ex->d_mapFlags.p; a.skip = skip ? ex->d_mapSkip.p : nullptr;
  a.posesL = ex->d_posesK.p; a.posesR = ex->d_posesK.p + n_frames; a.viewsL = ex->d_fviewsL.p; a.viewsR = ex->d_fviewsR.p;
  a.n = n; a.nlevels = ex->prm.nlevels;
  a.minX = min_x; a.minY = min_y; a.maxX = max_x; a.maxY = max_y; a.viewCosLimit = viewing_cos_limit;
  a.logScaleFactor = logf(ex->prm.scale_factor);
  HIPC(launch_project_map_kb8(a, n_frames, ex->stream));
  HIPC(hipStreamSynchronize(ex->stream));   // pageable host sources: the call returns when they have been consumed
  if (views_out || views_right_out) {       // the matcher's input form: left views (proj_xr = mTrackProjXR) + the right camera's members
    std::vector<orbx_map_point_view> hl((size_t)n_frames * n), hr((size_t)n_frames * n);
    HIPC(hipMemcpy(hl.data(), ex->d_fviewsL.p, hl.size() * sizeof(orbx_map_point_view), hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(hr.data(), ex->d_fviewsR.p, hr.size() * sizeof(orbx_map_point_view), hipMemcpyDeviceToHost));
    for (size_t o = 0; o < hl.size(); o++) {
      if (views_out) views_out[o] = hl[o];
      if (views_right_out) {
        orbx_map_point_right r{};
        r.proj_yr = hr[o].proj_y; r.view_cos_r = hr[o].view_cos;
        r.predicted_level_r = hr[o].in_view ? hr[o].predicted_level : -1;
        r.in_view_r = hr[o].in_view;
        views_right_out[o] = r;
      }
    }
  }
  ex->fviewsFrames = n_frames;
  ex->fviewsStride = n;
  return ORBX_OK;
}

int orbx_last_frames_upload(orbx_extractor* ex, int n_frames, int points_stride, const int32_t* n_points, const float* world_pos,
                            const int32_t* octave, const float* angle, const uint8_t* desc, const uint8_t* flags) {
  if (!ex || n_frames < 0 || n_frames > 4096 || points_stride < 0 || points_stride > 15000 ||
      (n_frames && (!n_points || (points_stride && (!world_pos || !octave || !angle || !desc || !flags)))))
    return fail(ORBX_E_BADARG, "bad argument (at most 15000 points per frame)");
  for (int f = 0; f < n_frames; f++)
    if (n_points[f] < 0 || n_points[f] > points_stride) return fail(ORBX_E_BADARG, "n_points[f] outside [0, points_stride]");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  HIPC(hipStreamSynchronize(ex->stream));   // (a projection of the previous upload may still read the arrays)
  ex->lfFrames = 0;
  ex->pviewsFrames = 0;
  const size_t m = std::max<size_t>((size_t)n_frames * points_stride, 1);
  hipError_t e = ex->d_lfPos.grow(3 * m);
  if (e == hipSuccess) e = ex->d_lfAngle.grow(m);
  if (e == hipSuccess) e = ex->d_lfOct.grow(m);
  if (e == hipSuccess) e = ex->d_lfDesc.grow(32 * m);
  if (e == hipSuccess) e = ex->d_lfFlags.grow(m);
  if (e == hipSuccess) e = ex->d_lfN.grow((size_t)std::max(n_frames, 1));
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  const size_t n = (size_t)n_frames * points_stride;
  if (n) {
    HIPC(hipMemcpy(ex->d_lfPos.p, world_pos, n * 12, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_lfOct.p, octave, n * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_lfAngle.p, angle, n * 4, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_lfDesc.p, desc, n * 32, hipMemcpyHostToDevice));
    HIPC(hipMemcpy(ex->d_lfFlags.p, flags, n, hipMemcpyHostToDevice));
  }
  if (n_frames) HIPC(hipMemcpy(ex->d_lfN.p, n_points, (size_t)n_frames * sizeof(int), hipMemcpyHostToDevice));
  ex->lfN.assign(n_points, n_points + n_frames);
  ex->lfFrames = n_frames;
  ex->lfStride = points_stride;
  return ORBX_OK;
}

int orbx_project_last_frames_batch(orbx_extractor* ex, int n_frames, const orbx_frame_pose_q* poses, float min_x, float min_y,
                                   float max_x, float max_y, float th, orbx_projected_point* views_out) {
  if (!ex || n_frames < 0 || (n_frames && !poses)) return fail(ORBX_E_BADARG, "bad argument");
  if (n_frames > ex->lfFrames) return fail(ORBX_E_BADARG, "more frames than LastFrames uploaded (orbx_last_frames_upload)");
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  ex->pviewsFrames = 0;
  if (n_frames == 0) return ORBX_OK;
  const int stride = ex->lfStride;
  hipError_t e = ex->d_posesQ.grow((size_t)n_frames);
  if (e == hipSuccess) e = ex->d_pviews.grow((size_t)n_frames * std::max(stride, 1));
  if (e == hipSuccess) e = ex->d_scaleF.grow((size_t)ex->prm.nlevels);
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  HIPC(hipMemcpyAsync(ex->d_posesQ.p, poses, (size_t)n_frames * sizeof(orbx_frame_pose_q), hipMemcpyHostToDevice, ex->stream));
  HIPC(hipMemcpyAsync(ex->d_scaleF.p, ex->scale.data(), (size_t)ex->prm.nlevels * sizeof(float), hipMemcpyHostToDevice, ex->stream));
  LastProjArgs a{};
  a.pos = ex->d_lfPos.p; a.octave = ex->d_lfOct.p; a.angle = ex->d_lfAngle.p; a.desc = ex->d_lfDesc.p; a.flags = ex->d_lfFlags.p;
  a.npts = ex->d_lfN.p; a.poses = ex->d_posesQ.p; a.scale = ex->d_scaleF.p; a.views = ex->d_pviews.p;
  a.stride = stride; a.nlevels = ex->prm.nlevels;
  a.minX = min_x; a.minY = min_y; a.maxX = max_x; a.maxY = max_y; a.th = th;
  HIPC(launch_project_last(a, n_frames, ex->stream));
  if (views_out && stride)
    HIPC(hipMemcpyAsync(views_out, ex->d_pviews.p, (size_t)n_frames * stride * sizeof(orbx_projected_point), hipMemcpyDeviceToHost, ex->stream));
  HIPC(hipStreamSynchronize(ex->stream));   // pageable host sources: the call returns when they have been consumed
  ex->pviewsFrames = n_frames;
  return ORBX_OK;
}

int orbx_search_by_projection_batch(orbx_extractor* ex, int first_image, int n_frames, float min_x, float min_y, float max_x,
                                    float max_y, const orbx_map_point_view* map_points, const int32_t* n_map_points,
                                    int points_stride, float th, int far_points, float th_far_points, float nnratio,
                                    int stereo_pair0, const uint8_t* occupied_in, uint8_t* occupied, int32_t* match,
                                    int32_t* n_matches) {
  return proj_batch_impl(ex, first_image, n_frames, min_x, min_y, max_x, max_y, 0, map_points, nullptr, n_map_points, points_stride,
                         th, far_points, th_far_points, nnratio, 0, stereo_pair0, occupied_in, occupied, match, n_matches);
}

int orbx_search_by_projection_frame_batch(orbx_extractor* ex, int first_image, int n_frames, float min_x, float min_y, float max_x,
                                          float max_y, const orbx_projected_point* points, const int32_t* n_points,
                                          int points_stride, int check_orientation, int stereo_pair0, const uint8_t* occupied_in,
                                          uint8_t* occupied, int32_t* match, int32_t* n_matches) {
  return proj_batch_impl(ex, first_image, n_frames, min_x, min_y, max_x, max_y, 1, nullptr, points, n_points, points_stride, 1.0f, 0,
                         0.f, 0.f, check_orientation, stereo_pair0, occupied_in, occupied, match, n_matches);
}

int orbx_search_for_triangulation_rig(int device, const uint32_t* node_ids1, const int32_t* node_start1, const uint32_t* feature_idx1,
                                      int n_nodes1, const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1,
                                      int n_left1, int n1, const uint32_t* node_ids2, const int32_t* node_start2,
                                      const uint32_t* feature_idx2, int n_nodes2, const orbx_keypoint* kps2, const uint8_t* desc2,
                                      const uint8_t* has_map_point2, int n_left2, int n2, const float* level_sigma2_1,
                                      const float* level_sigma2_2, int nlevels, const orbx_tri_rig* rig, int only_stereo, int coarse,
                                      int check_orientation, int32_t* matches12) {
  if (n1 < 0 || n2 < 0 || n_nodes1 < 0 || n_nodes2 < 0 || nlevels < 1 || !level_sigma2_1 || !level_sigma2_2 || !rig ||
      n_left1 < 0 || n_left1 > n1 || n_left2 < 0 || n_left2 > n2 || (n1 && (!matches12 || !kps1 || !desc1 || !has_map_point1)) ||
      (n2 && (!kps2 || !desc2 || !has_map_point2)) || (n_nodes1 && (!node_ids1 || !node_start1 || !feature_idx1)) ||
      (n_nodes2 && (!node_ids2 || !node_start2 || !feature_idx2)))
    return fail(ORBX_E_BADARG, "bad argument");
  const int nl1 = n_nodes1 ? node_start1[n_nodes1] : 0, nl2 = n_nodes2 ? node_start2[n_nodes2] : 0;
  if (nl1 < 0 || nl1 > n1 || nl2 < 0 || nl2 > n2) return fail(ORBX_E_BADARG, "feature vector larger than the key frame");
  if (nl2 >= (1 << 24)) return fail(ORBX_E_CAPACITY, "more than 2^24 features");
  for (int j = 0; j < n_nodes1; j++)
    if (node_start1[j] < 0 || node_start1[j] > node_start1[j + 1] || (j && node_ids1[j] <= node_ids1[j - 1]))
      return fail(ORBX_E_BADARG, "feature vector 1: node ids must ascend and offsets must be monotone");
  for (int j = 0; j < n_nodes2; j++)
    if (node_start2[j] < 0 || node_start2[j] > node_start2[j + 1] || (j && node_ids2[j] <= node_ids2[j - 1]))
      return fail(ORBX_E_BADARG, "feature vector 2: node ids must ascend and offsets must be monotone");
  for (int i = 0; i < nl1; i++)
    if (feature_idx1[i] >= (uint32_t)n1) return fail(ORBX_E_BADARG, "feature index 1 out of range");
  for (int i = 0; i < nl2; i++)
    if (feature_idx2[i] >= (uint32_t)n2) return fail(ORBX_E_BADARG, "feature index 2 out of range");
  for (int i = 0; i < n1; i++)
    if (kps1[i].octave < 0 || kps1[i].octave >= nlevels) return fail(ORBX_E_BADARG, "keypoint octave outside [0, nlevels)");
  for (int i = 0; i < n2; i++)
    if (kps2[i].octave < 0 || kps2[i].octave >= nlevels) return fail(ORBX_E_BADARG, "keypoint octave outside [0, nlevels)");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (nl1 == 0 || nl2 == 0) return 0;
  Pack pk;
  const size_t oN1 = pk.add(node_ids1, (size_t)n_nodes1 * 4), oS1 = pk.add(node_start1, ((size_t)n_nodes1 + 1) * 4);
  const size_t oF1 = pk.add(feature_idx1, (size_t)nl1 * 4), oD1 = pk.add(desc1, (size_t)n1 * 32);
  const size_t oM1 = pk.add(has_map_point1, n1), oK1 = pk.add(kps1, (size_t)n1 * sizeof(orbx_keypoint));
  const size_t oN2 = pk.add(node_ids2, (size_t)n_nodes2 * 4), oS2 = pk.add(node_start2, ((size_t)n_nodes2 + 1) * 4);
  const size_t oF2 = pk.add(feature_idx2, (size_t)nl2 * 4), oD2 = pk.add(desc2, (size_t)n2 * 32);
  const size_t oM2 = pk.add(has_map_point2, n2), oK2 = pk.add(kps2, (size_t)n2 * sizeof(orbx_keypoint));
  const size_t oG1 = pk.add(level_sigma2_1, (size_t)nlevels * 4), oG2 = pk.add(level_sigma2_2, (size_t)nlevels * 4);
  const size_t oRig = pk.add(rig, sizeof(orbx_tri_rig));
  const size_t oFlags = pk.add(nullptr, 33 * 4);
  const size_t oOut = pk.add(nullptr, ((size_t)n1 + 1) * 4);
  hipError_t e = pk.commit();
  TriArgs a{};
  a.nodes1 = pk.ptr<uint32_t>(oN1); a.start1 = pk.ptr<int>(oS1); a.feat1 = pk.ptr<uint32_t>(oF1); a.nNodes1 = n_nodes1; a.nList1 = nl1;
  a.nodes2 = pk.ptr<uint32_t>(oN2); a.start2 = pk.ptr<int>(oS2); a.feat2 = pk.ptr<uint32_t>(oF2); a.nNodes2 = n_nodes2;
  a.k1 = pk.ptr<orbx_keypoint>(oK1); a.k2 = pk.ptr<orbx_keypoint>(oK2);
  a.d1 = pk.ptr<uint32_t>(oD1); a.d2 = pk.ptr<uint32_t>(oD2); a.mp1 = pk.ptr<uint8_t>(oM1); a.mp2 = pk.ptr<uint8_t>(oM2);
  a.n1 = n1; a.n2 = n2; a.nLeft1 = n_left1; a.nLeft2 = n_left2;
  a.sigma1 = pk.ptr<float>(oG1); a.sigma2 = pk.ptr<float>(oG2); a.rig = pk.ptr<orbx_tri_rig>(oRig);
  a.onlyStereo = only_stereo ? 1 : 0; a.coarse = coarse ? 1 : 0; a.checkOri = check_orientation ? 1 : 0;
  a.flags = pk.ptr<int>(oFlags); a.result = pk.ptr<int>(oOut); a.match = pk.ptr<int>(oOut) + 1;
  if (e == hipSuccess) e = launch_search_for_triangulation(a, nullptr);
  int n = 0;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOut, ((size_t)n1 + 1) * 4, &e);
    if (e == hipSuccess) {
      std::memcpy(&n, h, 4);
      std::memcpy(matches12, h + 4, (size_t)n1 * 4);
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return n;
}

int orbx_search_by_bow_keyframes(int device, const uint32_t* node_ids1, const int32_t* node_start1, const uint32_t* feature_idx1,
                                 int n_nodes1, const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* valid1, int n1,
                                 const uint32_t* node_ids2, const int32_t* node_start2, const uint32_t* feature_idx2, int n_nodes2,
                                 const orbx_keypoint* kps2, const uint8_t* desc2, const uint8_t* valid2, int n2, float nnratio,
                                 int check_orientation, int32_t* matches12) {
  if (n1 < 0 || n2 < 0 || n_nodes1 < 0 || n_nodes2 < 0 || (n1 && (!matches12 || !kps1 || !desc1 || !valid1)) ||
      (n2 && (!kps2 || !desc2 || !valid2)) || (n_nodes1 && (!node_ids1 || !node_start1 || !feature_idx1)) ||
      (n_nodes2 && (!node_ids2 || !node_start2 || !feature_idx2)))
    return fail(ORBX_E_BADARG, "bad argument");
  const int nl1 = n_nodes1 ? node_start1[n_nodes1] : 0, nl2 = n_nodes2 ? node_start2[n_nodes2] : 0;
  if (nl1 < 0 || nl1 > n1 || nl2 < 0 || nl2 > n2) return fail(ORBX_E_BADARG, "feature vector larger than the key frame");
  for (int j = 0; j < n_nodes1; j++)
    if (node_start1[j] < 0 || node_start1[j] > node_start1[j + 1] || (j && node_ids1[j] <= node_ids1[j - 1]))
      return fail(ORBX_E_BADARG, "feature vector 1: node ids must ascend and offsets must be monotone");
  for (int j = 0; j < n_nodes2; j++)
    if (node_start2[j] < 0 || node_start2[j] > node_start2[j + 1] || (j && node_ids2[j] <= node_ids2[j - 1]))
      return fail(ORBX_E_BADARG, "feature vector 2: node ids must ascend and offsets must be monotone");
  for (int i = 0; i < nl1; i++)
    if (feature_idx1[i] >= (uint32_t)n1) return fail(ORBX_E_BADARG, "feature index 1 out of range");
  for (int i = 0; i < nl2; i++)
    if (feature_idx2[i] >= (uint32_t)n2) return fail(ORBX_E_BADARG, "feature index 2 out of range");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (nl1 == 0 || nl2 == 0) return 0;
  Pack pk;
  const size_t oN1 = pk.add(node_ids1, (size_t)n_nodes1 * 4), oS1 = pk.add(node_start1, ((size_t)n_nodes1 + 1) * 4);
  const size_t oF1 = pk.add(feature_idx1, (size_t)nl1 * 4), oD1 = pk.add(desc1, (size_t)n1 * 32);
  const size_t oV1 = pk.add(valid1, n1), oK1 = pk.add(kps1, (size_t)n1 * sizeof(orbx_keypoint));
  const size_t oN2 = pk.add(node_ids2, (size_t)n_nodes2 * 4), oS2 = pk.add(node_start2, ((size_t)n_nodes2 + 1) * 4);
  const size_t oF2 = pk.add(feature_idx2, (size_t)nl2 * 4), oD2 = pk.add(desc2, (size_t)n2 * 32);
  const size_t oV2 = pk.add(valid2, n2), oK2 = pk.add(kps2, (size_t)n2 * sizeof(orbx_keypoint));
  const size_t oFlags = pk.add(nullptr, 33 * 4);
  const size_t oOut = pk.add(nullptr, ((size_t)n1 + 1) * 4);  // result, then the matches: one copy back
  hipError_t e = pk.commit();
  TriArgs a{};
  a.nodes1 = pk.ptr<uint32_t>(oN1); a.start1 = pk.ptr<int>(oS1); a.feat1 = pk.ptr<uint32_t>(oF1); a.nNodes1 = n_nodes1; a.nList1 = nl1;
  a.nodes2 = pk.ptr<uint32_t>(oN2); a.start2 = pk.ptr<int>(oS2); a.feat2 = pk.ptr<uint32_t>(oF2); a.nNodes2 = n_nodes2;
  a.k1 = pk.ptr<orbx_keypoint>(oK1); a.k2 = pk.ptr<orbx_keypoint>(oK2);
  a.d1 = pk.ptr<uint32_t>(oD1); a.d2 = pk.ptr<uint32_t>(oD2); a.mp1 = pk.ptr<uint8_t>(oV1); a.mp2 = pk.ptr<uint8_t>(oV2);
  a.n1 = n1; a.n2 = n2; a.nnratio = nnratio; a.checkOri = check_orientation ? 1 : 0;
  a.flags = pk.ptr<int>(oFlags); a.result = pk.ptr<int>(oOut); a.match = pk.ptr<int>(oOut) + 1;
  if (e == hipSuccess) e = launch_search_by_bow_keyframes(a, nullptr);
  int n = 0;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOut, ((size_t)n1 + 1) * 4, &e);
    if (e == hipSuccess) {
      std::memcpy(&n, h, 4);
      std::memcpy(matches12, h + 4, (size_t)n1 * 4);
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  if (n < 0) {
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    return fail(ORBX_E_UNSUPPORTED, "a vocabulary node holds more than 4096 features of pKF2");
  }
  return n;
}

int orbx_fuse_search(int device, const orbx_keypoint* kps, const uint8_t* desc, const float* u_right, int n, float min_x,
                     float min_y, float max_x, float max_y, const float* inv_level_sigma2, int nlevels,
                     const orbx_fuse_point* points, int n_points, int max_dist, int32_t* best_idx, int32_t* best_dist) {
  if (max_dist < 0 || max_dist > 255) return fail(ORBX_E_BADARG, "max_dist outside [0, 255]");
  if (n < 0 || n_points < 0 || nlevels < 1 || !inv_level_sigma2 || (n && (!kps || !desc)) || (n_points && (!points || !best_idx)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n >= (1 << 20)) return fail(ORBX_E_CAPACITY, "more than 2^20 keypoints");
  for (int i = 0; i < n; i++)  // kp.octave indexes mvInvLevelSigma2 (:1227,1236)
    if (kps[i].octave < 0 || kps[i].octave >= nlevels) return fail(ORBX_E_BADARG, "keypoint octave outside [0, nlevels)");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  for (int i = 0; i < n_points; i++) {
    best_idx[i] = -1;
    if (best_dist) best_dist[i] = 256;
  }
  if (n == 0 || n_points == 0) return 0;
  ScratchBuf<int> cellStart, cellItems, mdist, m21, m12;
  hipError_t e = hipSuccess;
  auto chk = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  Pack pk;
  const size_t oK = pk.add(kps, (size_t)n * sizeof(orbx_keypoint)), oD = pk.add(desc, (size_t)n * 32);
  const size_t oU = pk.add(u_right, (size_t)n * 4), oS = pk.add(inv_level_sigma2, (size_t)nlevels * 4);
  const size_t oP = pk.add(points, (size_t)n_points * sizeof(orbx_fuse_point));
  static const int zero4[4] = {0, 0, 0, 0};
  const size_t oRes = pk.add(zero4, sizeof(zero4));  // [0] nFused, [2..3] the grid kernel's own result words; then best_idx | best_dist: one copy back
  const size_t oBi = pk.add(nullptr, (size_t)n_points * 4), oBd = pk.add(nullptr, (size_t)n_points * 4);
  const size_t outBytes = oBd + (size_t)n_points * 4 - oRes;
  chk(pk.commit());
  chk(cellStart.alloc(64 * 48 + 1)); chk(cellItems.alloc(n)); chk(mdist.alloc(n)); chk(m21.alloc(n)); chk(m12.alloc(1));
  FuseArgs a{};
  a.grid.k2 = pk.ptr<orbx_keypoint>(oK); a.grid.n2 = n; a.grid.n1 = 0;
  a.grid.minX = min_x; a.grid.minY = min_y;
  a.grid.invW = 64.f / (max_x - min_x);
  a.grid.invH = 48.f / (max_y - min_y);
  a.grid.cellStart = cellStart.p; a.grid.cellItems = cellItems.p; a.grid.matchedDist = mdist.p; a.grid.matches21 = m21.p;
  a.grid.matches12 = m12.p; a.grid.result = pk.ptr<int>(oRes) + 2; a.grid.candCap = 1 << 30;
  a.desc = pk.ptr<uint32_t>(oD); a.uRight = u_right ? pk.ptr<float>(oU) : nullptr; a.invSigma2 = pk.ptr<float>(oS);
  a.pts = pk.ptr<orbx_fuse_point>(oP); a.npts = n_points; a.maxDist = max_dist;
  a.bestIdx = pk.ptr<int>(oBi); a.bestDist = pk.ptr<int>(oBd); a.result = pk.ptr<int>(oRes);
  if (e == hipSuccess) chk(launch_fuse_search(a, nullptr));
  int nf = 0;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oRes, outBytes, &e);
    if (e == hipSuccess) {
      std::memcpy(&nf, h, 4);
      std::memcpy(best_idx, h + (oBi - oRes), (size_t)n_points * 4);
      if (best_dist) std::memcpy(best_dist, h + (oBd - oRes), (size_t)n_points * 4);
    }
  }
  pk.release(); cellStart.free(); cellItems.free(); mdist.free(); m21.free(); m12.free();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return nf;
}

int orbx_search_for_triangulation(int device, const uint32_t* node_ids1, const int32_t* node_start1, const uint32_t* feature_idx1,
                                  int n_nodes1, const orbx_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_map_point1,
                                  const float* u_right1, int n1, const uint32_t* node_ids2, const int32_t* node_start2,
                                  const uint32_t* feature_idx2, int n_nodes2, const orbx_keypoint* kps2, const uint8_t* desc2,
                                  const uint8_t* has_map_point2, const float* u_right2, int n2, const float* scale_factors2,
                                  const float* level_sigma2_2, int nlevels2, const float ep[2], const float F12[9], int only_stereo,
                                  int coarse, int check_orientation, int32_t* matches12) {
  if (n1 < 0 || n2 < 0 || n_nodes1 < 0 || n_nodes2 < 0 || nlevels2 < 1 || !scale_factors2 || !level_sigma2_2 || !ep ||
      (!coarse && !F12) || (n1 && (!matches12 || !kps1 || !desc1 || !has_map_point1)) || (n2 && (!kps2 || !desc2 || !has_map_point2)) ||
      (n_nodes1 && (!node_ids1 || !node_start1 || !feature_idx1)) || (n_nodes2 && (!node_ids2 || !node_start2 || !feature_idx2)))
    return fail(ORBX_E_BADARG, "bad argument");
  const int nl1 = n_nodes1 ? node_start1[n_nodes1] : 0, nl2 = n_nodes2 ? node_start2[n_nodes2] : 0;
  if (nl1 < 0 || nl1 > n1 || nl2 < 0 || nl2 > n2) return fail(ORBX_E_BADARG, "feature vector larger than the key frame");
  if (nl2 >= (1 << 24)) return fail(ORBX_E_CAPACITY, "more than 2^24 features");
  for (int j = 0; j < n_nodes1; j++)
    if (node_start1[j] < 0 || node_start1[j] > node_start1[j + 1] || (j && node_ids1[j] <= node_ids1[j - 1]))
      return fail(ORBX_E_BADARG, "feature vector 1: node ids must ascend and offsets must be monotone");
  for (int j = 0; j < n_nodes2; j++)
    if (node_start2[j] < 0 || node_start2[j] > node_start2[j + 1] || (j && node_ids2[j] <= node_ids2[j - 1]))
      return fail(ORBX_E_BADARG, "feature vector 2: node ids must ascend and offsets must be monotone");
  for (int i = 0; i < nl1; i++)
    if (feature_idx1[i] >= (uint32_t)n1) return fail(ORBX_E_BADARG, "feature index 1 out of range");
  for (int i = 0; i < nl2; i++)
    if (feature_idx2[i] >= (uint32_t)n2) return fail(ORBX_E_BADARG, "feature index 2 out of range");
  for (int i = 0; i < n2; i++)  // kp2.octave indexes mvScaleFactors / mvLevelSigma2 (:1001,1054)
    if (kps2[i].octave < 0 || kps2[i].octave >= nlevels2) return fail(ORBX_E_BADARG, "keypoint octave outside [0, nlevels2)");
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (nl1 == 0 || nl2 == 0) return 0;
  Pack pk;
  const size_t oN1 = pk.add(node_ids1, (size_t)n_nodes1 * 4), oS1 = pk.add(node_start1, ((size_t)n_nodes1 + 1) * 4);
  const size_t oF1 = pk.add(feature_idx1, (size_t)nl1 * 4), oD1 = pk.add(desc1, (size_t)n1 * 32);
  const size_t oM1 = pk.add(has_map_point1, n1), oK1 = pk.add(kps1, (size_t)n1 * sizeof(orbx_keypoint));
  const size_t oU1 = pk.add(u_right1, (size_t)n1 * 4);
  const size_t oN2 = pk.add(node_ids2, (size_t)n_nodes2 * 4), oS2 = pk.add(node_start2, ((size_t)n_nodes2 + 1) * 4);
  const size_t oF2 = pk.add(feature_idx2, (size_t)nl2 * 4), oD2 = pk.add(desc2, (size_t)n2 * 32);
  const size_t oM2 = pk.add(has_map_point2, n2), oK2 = pk.add(kps2, (size_t)n2 * sizeof(orbx_keypoint));
  const size_t oU2 = pk.add(u_right2, (size_t)n2 * 4);
  const size_t oSf = pk.add(scale_factors2, (size_t)nlevels2 * 4), oSg = pk.add(level_sigma2_2, (size_t)nlevels2 * 4);
  const size_t oFlags = pk.add(nullptr, 33 * 4);
  const size_t oOut = pk.add(nullptr, ((size_t)n1 + 1) * 4);  // result, then vMatches12: one copy back
  hipError_t e = pk.commit();
  TriArgs a{};
  a.nodes1 = pk.ptr<uint32_t>(oN1); a.start1 = pk.ptr<int>(oS1); a.feat1 = pk.ptr<uint32_t>(oF1); a.nNodes1 = n_nodes1; a.nList1 = nl1;
  a.nodes2 = pk.ptr<uint32_t>(oN2); a.start2 = pk.ptr<int>(oS2); a.feat2 = pk.ptr<uint32_t>(oF2); a.nNodes2 = n_nodes2;
  a.k1 = pk.ptr<orbx_keypoint>(oK1); a.k2 = pk.ptr<orbx_keypoint>(oK2);
  a.d1 = pk.ptr<uint32_t>(oD1); a.d2 = pk.ptr<uint32_t>(oD2); a.mp1 = pk.ptr<uint8_t>(oM1); a.mp2 = pk.ptr<uint8_t>(oM2);
  a.ur1 = u_right1 ? pk.ptr<float>(oU1) : nullptr; a.ur2 = u_right2 ? pk.ptr<float>(oU2) : nullptr;
  a.n1 = n1; a.n2 = n2; a.scale2 = pk.ptr<float>(oSf); a.sigma2 = pk.ptr<float>(oSg);
  a.ep0 = ep[0]; a.ep1 = ep[1];
  for (int i = 0; i < 9; i++) a.F[i] = F12 ? F12[i] : 0.f;
  a.onlyStereo = only_stereo ? 1 : 0; a.coarse = coarse ? 1 : 0; a.checkOri = check_orientation ? 1 : 0;
  a.flags = pk.ptr<int>(oFlags); a.result = pk.ptr<int>(oOut); a.match = pk.ptr<int>(oOut) + 1;
  if (e == hipSuccess) e = launch_search_for_triangulation(a, nullptr);
  int n = 0;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOut, ((size_t)n1 + 1) * 4, &e);
    if (e == hipSuccess) {
      std::memcpy(&n, h, 4);
      std::memcpy(matches12, h + 4, (size_t)n1 * 4);
    }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return n;
}

int orbx_search_by_projection_keyframe(int device, const orbx_keypoint* kps_un, const uint8_t* desc, int n, float min_x,
                                       float min_y, float max_x, float max_y, const orbx_projected_point* points,
                                       int n_points, int orb_dist, int check_orientation, uint8_t* occupied,
                                       int32_t* match) {
  if (n < 0 || n_points < 0 || (n && (!kps_un || !desc || !occupied || !match)) || (n_points && !points))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n_points > 15000) return fail(ORBX_E_CAPACITY, "more than 15000 projected points");
  // bestDist starts at 256 and is only replaced by a strictly smaller distance (src/ORBmatcher.cc:1864-1882): no candidate
  // ever yields 256, and with no free candidate bestIdx2 stays -1 -- the reference would then index mvpMapPoints[-1] for
  // ORBdist >= 256; the thresholds it is called with are 100 and 64 (src/Tracking.cc:3631-3632,3645-3646)
  if (orb_dist < 0 || orb_dist > 255) return fail(ORBX_E_BADARG, "ORBdist outside [0, 255]");
  static const orbx_projected_point dummy{};
  return search_by_projection_impl(device, kps_un, desc, nullptr, n, min_x, min_y, max_x, max_y, nullptr, 0, nullptr,
                                   points ? points : &dummy, n_points, 1.0f, 0, 0.f, 0.f, check_orientation, occupied, match,
                                   orb_dist, 1);
}

namespace {
// One side (left or right camera) of a stereo-fisheye projection search: grid + candidate lists on the device.
struct ProjSide {
  ScratchBuf<int> cellStart, cellItems, candOff, candIdx, candDist, mdist, m21, m12;
  orbx_map_point_view* mpp = nullptr;  // views of this camera inside the call's packed upload
  orbx_projected_point* ppp = nullptr;
  ProjArgs a{};
  void release() {
    cellStart.free(); cellItems.free(); candOff.free(); candIdx.free(); candDist.free(); mdist.free(); m21.free();
    m12.free();
  }
};

// One attempt with candidate arrays of cand_cap entries per camera; see search_by_projection_try.
int search_by_projection_fisheye_try(const orbx_keypoint* kps, const uint8_t* desc, int n_left, int n_right,
                                     float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                                     int nlevels, const orbx_map_point_view* viewsL, const orbx_map_point_view* viewsR,
                                     const orbx_projected_point* ptsL, const orbx_projected_point* ptsR, int n_points,
                                     float th, int far_points, float th_far_points, float nnratio, int check_ori,
                                     const int32_t* l2r, const int32_t* r2l, uint8_t* occupied, int32_t* match, int cand_cap,
                                     int* needed) {
  const int mode = ptsL ? 1 : 0, n = n_left + n_right;
  *needed = 0;
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
  static const int zero4[4] = {0, 0, 0, 0};
  const size_t oOcc = pk.add(occupied, n), oSideRes = pk.add(zero4, 4 * sizeof(int));  // per camera: {count, lists needed}
  const size_t oRes = pk.add(nullptr, 2 * sizeof(int)), oMt = pk.add(nullptr, (size_t)n * sizeof(int));
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
    chk(P.mdist.alloc(std::max(ns, 1))); chk(P.m21.alloc(std::max(ns, 1))); chk(P.m12.alloc(1));
    chk(P.candIdx.alloc((size_t)cand_cap)); chk(P.candDist.alloc((size_t)cand_cap));
    int* sideRes = pk.ptr<int>(oSideRes) + 2 * side;
    P.mpp = pk.ptr<orbx_map_point_view>(oMp[side]);
    P.ppp = pk.ptr<orbx_projected_point>(oPp[side]);
    ProjArgs& a = P.a;
    a.grid.k2 = k.p + first; a.grid.n2 = ns; a.grid.n1 = 0;
    a.grid.minX = min_x; a.grid.minY = min_y;
    a.grid.invW = 64.f / (max_x - min_x);
    a.grid.invH = 48.f / (max_y - min_y);
    a.grid.cellStart = P.cellStart.p; a.grid.cellItems = P.cellItems.p; a.grid.matchedDist = P.mdist.p;
    a.grid.matches21 = P.m21.p; a.grid.matches12 = P.m12.p; a.grid.result = sideRes; a.grid.candOff = P.candOff.p;
    a.grid.candCap = cand_cap;
    a.desc = d.p + (size_t)first * 32; a.uRight = nullptr;  // no mvuRight gate when F.Nleft != -1 (:90, :1667)
    a.scale = sf.p; a.mps = P.mpp; a.pts = P.ppp; a.nmp = n_points; a.mode = mode; a.checkOri = check_ori;
    a.th = side ? 1.0f : th;  // the right-camera radius is not scaled by th (:144)
    a.thFar = th_far_points; a.nnratio = nnratio; a.far = far_points;
    a.occupied = occ.p + first; a.match = mt.p + first; a.candOff = P.candOff.p; a.result = sideRes; a.candCap = cand_cap;
    a.candIdx = P.candIdx.p; a.candDist = P.candDist.p;
    if (e != hipSuccess) break;
    if (ns > 0) {
      chk(launch_proj_count(a, nullptr));
      chk(launch_proj_cands_fill(a, nullptr));
    } else {
      chk(hipMemsetAsync(P.candOff.p, 0, (size_t)(nm + 1) * sizeof(int), nullptr));
    }
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
  ScratchBuf<int> wl0, wl1, wl2, wc0, wc1, wc2, fl;
  static const bool forceSerial = getenv("ORBX_PROJ_SERIAL") && atoi(getenv("ORBX_PROJ_SERIAL")) != 0;
  bool done = false;
  int lastRound = 0;
  if (!forceSerial && n_points > 0) {
    chk(wr0.alloc(nm)); chk(wr1.alloc(nm)); chk(wl0.alloc((size_t)n * kFeWriters)); chk(wl1.alloc((size_t)n * kFeWriters));
    chk(wl2.alloc((size_t)n * kFeWriters)); chk(wc0.alloc(n)); chk(wc1.alloc(n)); chk(wc2.alloc(n)); chk(fl.alloc(40 + 48));
    f.writes[0] = wr0.p; f.writes[1] = wr1.p; f.writers[0] = wl0.p; f.writers[1] = wl1.p; f.writers[2] = wl2.p;
    f.nwriters[0] = wc0.p; f.nwriters[1] = wc1.p; f.nwriters[2] = wc2.p; f.flags = fl.p;
    for (int r = 0; r < 48 && e == hipSuccess && !done; r += 4) {
      chk(launch_proj_rounds_fisheye(f, r, 4, nullptr));
      int st[40 + 48];
      st[1] = 0;
      st[40 + r + 3] = 1;
      if (e == hipSuccess) chk(hipMemcpy(st, fl.p, sizeof(st), hipMemcpyDeviceToHost));  // synchronises
      if (st[1]) break;  // a slot collected more than kFeWriters writers in one round
      done = st[40 + r + 3] == 0;  // the last round of the group changed nothing
      lastRound = r + 3;
    }
  }
  if (e == hipSuccess) chk(done ? launch_proj_finish_fisheye(f, lastRound, nullptr) : launch_proj_resolve_fisheye(f, nullptr));
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOcc, outBytes, &e);  // synchronises
    if (e == hipSuccess) {
      int side_res[4];
      std::memcpy(side_res, h + (oSideRes - oOcc), sizeof(side_res));
      std::memcpy(result, h + (oRes - oOcc), sizeof(int));
      if (std::max(side_res[1], side_res[3]) > cand_cap) {
        *needed = std::max(side_res[1], side_res[3]);
      } else {
        std::memcpy(occupied, h, n);
        std::memcpy(match, h + (oMt - oOcc), (size_t)n * sizeof(int));
      }
    }
  }
  wr0.free(); wr1.free(); wl0.free(); wl1.free(); wl2.free(); wc0.free(); wc1.free(); wc2.free(); fl.free();
  pk.release();
  S[0].release(); S[1].release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  return result[0];
}

int search_by_projection_fisheye_impl(int device, const orbx_keypoint* kps, const uint8_t* desc, int n_left, int n_right,
                                      float min_x, float min_y, float max_x, float max_y, const float* scale_factors,
                                      int nlevels, const orbx_map_point_view* viewsL, const orbx_map_point_view* viewsR,
                                      const orbx_projected_point* ptsL, const orbx_projected_point* ptsR, int n_points,
                                      float th, int far_points, float th_far_points, float nnratio, int check_ori,
                                      const int32_t* l2r, const int32_t* r2l, uint8_t* occupied, int32_t* match) {
  if (n_left + n_right == 0) return 0;
  int rc = set_device(device);
  if (rc != ORBX_OK) return rc;
  static const int capEnv = getenv("ORBX_PROJ_CAND_CAP") ? atoi(getenv("ORBX_PROJ_CAND_CAP")) : 0;
  int cap = capEnv > 0 ? capEnv : std::max(n_points, 1) * 96, needed = 0;
  rc = search_by_projection_fisheye_try(kps, desc, n_left, n_right, min_x, min_y, max_x, max_y, scale_factors, nlevels, viewsL,
                                        viewsR, ptsL, ptsR, n_points, th, far_points, th_far_points, nnratio, check_ori, l2r, r2l,
                                        occupied, match, cap, &needed);
  if (rc >= 0 && needed > cap)
    rc = search_by_projection_fisheye_try(kps, desc, n_left, n_right, min_x, min_y, max_x, max_y, scale_factors, nlevels, viewsL,
                                          viewsR, ptsL, ptsR, n_points, th, far_points, th_far_points, nnratio, check_ori, l2r,
                                          r2l, occupied, match, needed, &needed);
  return rc;
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

namespace {
// The two stereo-fisheye SearchByProjection flavours on the two-camera frames of an extraction batch (round 5; the last one-shot
// matchers of VERDICT round 4, item 5).  Frame f = left image first_left + f and right image first_right + f of the handle's last
// batch; their keypoints / descriptors are laid side by side on the device ([left | right], k_fe_concat) and never visit the host.
// One upload (views / points of both cameras, left-right matches, occupancy), one launch per kernel for all frames and cameras
// (launch_proj_fisheye_batch), kBlind fixed-point rounds without a convergence-flag read, one download.  A frame whose candidate
// lists or writer lists overflowed or whose writes did not settle is redone through the one-shot path.
int proj_fisheye_batch_impl(orbx_extractor* ex, int first_left, int first_right, int n_frames, float min_x, float min_y, float max_x,
                            float max_y, int mode, const orbx_map_point_view* viewsL, const orbx_map_point_view* viewsR,
                            const orbx_projected_point* ptsL, const orbx_projected_point* ptsR, const int32_t* n_points, int stride,
                            float th, int far_points, float th_far, float nnratio, int check_ori, const int32_t* l2r,
                            const int32_t* r2l, const uint8_t* occupied_in, uint8_t* occupied, int32_t* match, int32_t* n_matches) {
  const int F = n_frames, cap = ex->gmax.outCap, nlevels = ex->prm.nlevels, st = std::max(stride, 1);
  int maxPts = 0;
  for (int f = 0; f < F; f++) maxPts = std::max(maxPts, n_points[f]);
  const bool devViews = mode == 0 && !viewsL && !viewsR && maxPts > 0;   // both lists made by orbx_project_map_points_fisheye_batch
  std::vector<orbx_map_point_view> hvL, hvR;                             // (host copies for a frame that falls back to the one-shot path)
  int rc = set_device(ex->device);
  if (rc != ORBX_OK) return rc;
  std::vector<int> nL(F), nR(F);
  HIPC(hipStreamSynchronize(ex->stream));
  HIPC(hipMemcpy(nL.data(), ex->d_nOut.p + first_left, (size_t)F * sizeof(int), hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(nR.data(), ex->d_nOut.p + first_right, (size_t)F * sizeof(int), hipMemcpyDeviceToHost));
  int maxN = 0;
  for (int f = 0; f < F; f++) {
    nL[f] = std::min(std::max(nL[f], 0), cap);
    nR[f] = std::min(std::max(nR[f], 0), cap);
    maxN = std::max(maxN, nL[f] + nR[f]);
    if (mode == 0) {
      for (int i = 0; i < nL[f]; i++)
        if (l2r[(size_t)f * cap + i] < -1 || l2r[(size_t)f * cap + i] >= nR[f]) return fail(ORBX_E_BADARG, "left_to_right entry out of range");
      for (int i = 0; i < nR[f]; i++)
        if (r2l[(size_t)f * cap + i] < -1 || r2l[(size_t)f * cap + i] >= nL[f]) return fail(ORBX_E_BADARG, "right_to_left entry out of range");
    }
  }
  static const int capEnv = getenv("ORBX_PROJ_CAND_CAP") ? atoi(getenv("ORBX_PROJ_CAND_CAP")) : 0;
  static const int kBlind = getenv("ORBX_PROJ_BLIND") ? std::min(48, std::max(1, atoi(getenv("ORBX_PROJ_BLIND")))) : 16;
  const int nm = std::max(maxPts, 1), candCap = capEnv > 0 ? capEnv : nm * 96, n2c = 2 * cap;
  const size_t ptBytes = mode == 0 ? sizeof(orbx_map_point_view) : sizeof(orbx_projected_point);
  constexpr int kFlags = 40 + 48;
  Pack pk;
  std::vector<ProjArgs> sides(2 * (size_t)F);
  std::vector<ProjFeArgs> frames(F);
  std::vector<uint8_t> occ0;
  if (!occupied_in) occ0.assign((size_t)F * n2c, 0);
  const size_t ptsRead = ((size_t)(F - 1) * st + (size_t)n_points[F - 1]) * ptBytes;
  const void* srcL = mode == 0 ? (const void*)viewsL : (const void*)ptsL;
  const void* srcR = mode == 0 ? (const void*)viewsR : (const void*)ptsR;
  const size_t oPL = devViews ? pk.add(nullptr, 16) : pk.add(maxPts ? srcL : nullptr, (size_t)F * st * ptBytes, ptsRead);
  const size_t oPR = devViews ? pk.add(nullptr, 16) : pk.add(maxPts ? srcR : nullptr, (size_t)F * st * ptBytes, ptsRead);
  const size_t oSf = pk.add(ex->scale.data(), (size_t)nlevels * sizeof(float));
  const size_t oA12 = pk.add(mode == 0 ? l2r : nullptr, (size_t)F * cap * 4), oA21 = pk.add(mode == 0 ? r2l : nullptr, (size_t)F * cap * 4);
  const size_t oSd = pk.add(sides.data(), sides.size() * sizeof(ProjArgs)), oFr = pk.add(frames.data(), (size_t)F * sizeof(ProjFeArgs));
  const size_t oOcc = pk.add(occupied_in ? occupied_in : occ0.data(), (size_t)F * n2c);
  const size_t oMt = pk.add(nullptr, (size_t)F * n2c * sizeof(int)), oRes = pk.add(nullptr, (size_t)F * 2 * sizeof(int));
  const size_t oSideRes = pk.add(nullptr, (size_t)F * 4 * sizeof(int)), oFlags = pk.add(nullptr, (size_t)F * kFlags * sizeof(int));
  const size_t outBytes = oFlags + (size_t)F * kFlags * sizeof(int) - oOcc;
  auto per = [&](size_t ints) { return pk.add(nullptr, (size_t)F * ints * sizeof(int)); };
  const size_t oKc = pk.add(nullptr, (size_t)F * n2c * sizeof(orbx_keypoint)), oDc = pk.add(nullptr, (size_t)F * n2c * 32);
  const size_t cs = 64 * 48 + 4;
  size_t oCs[2], oCi[2], oMd[2], oM21[2], oM12[2], oCo[2], oCx[2], oCd[2];
  for (int sd = 0; sd < 2; sd++) {
    oCs[sd] = per(cs); oCi[sd] = per(cap); oMd[sd] = per(cap); oM21[sd] = per(cap); oM12[sd] = per(4);
    oCo[sd] = per((size_t)nm + 4); oCx[sd] = per(candCap); oCd[sd] = per(candCap);
  }
  const size_t oW0 = per(4 * (size_t)nm), oW1 = per(4 * (size_t)nm);
  const size_t oWl0 = per((size_t)n2c * kFeWriters), oWl1 = per((size_t)n2c * kFeWriters), oWl2 = per((size_t)n2c * kFeWriters);
  const size_t oWc0 = per(n2c), oWc1 = per(n2c), oWc2 = per(n2c);
  hipError_t e = pk.reserve();
  if (e != hipSuccess) { pk.release(); return fail(ORBX_E_HIP, hipGetErrorString(e)); }
  for (int f = 0; f < F; f++) {
    orbx_keypoint* kc = pk.ptr<orbx_keypoint>(oKc) + (size_t)f * n2c;
    uint8_t* dc = pk.ptr<uint8_t>(oDc) + (size_t)f * n2c * 32;
    uint8_t* occ = pk.ptr<uint8_t>(oOcc) + (size_t)f * n2c;
    int* mt = pk.ptr<int>(oMt) + (size_t)f * n2c;
    for (int sd = 0; sd < 2; sd++) {
      ProjArgs a{};
      const int ns = sd ? nR[f] : nL[f], first = sd ? nL[f] : 0;
      int* sideRes = pk.ptr<int>(oSideRes) + (size_t)f * 4 + 2 * sd;
      a.grid.k2 = kc + first; a.grid.n2 = ns; a.grid.n1 = 0;
      a.grid.minX = min_x; a.grid.minY = min_y;
      a.grid.invW = 64.f / (max_x - min_x);
      a.grid.invH = 48.f / (max_y - min_y);
      a.grid.cellStart = pk.ptr<int>(oCs[sd]) + (size_t)f * cs; a.grid.cellItems = pk.ptr<int>(oCi[sd]) + (size_t)f * cap;
      a.grid.matchedDist = pk.ptr<int>(oMd[sd]) + (size_t)f * cap; a.grid.matches21 = pk.ptr<int>(oM21[sd]) + (size_t)f * cap;
      a.grid.matches12 = pk.ptr<int>(oM12[sd]) + (size_t)f * 4; a.grid.result = sideRes;
      a.grid.candOff = pk.ptr<int>(oCo[sd]) + (size_t)f * (nm + 4); a.grid.candCap = 1 << 30;
      a.desc = dc + (size_t)first * 32; a.uRight = nullptr;   // no mvuRight gate when F.Nleft != -1 (:90, :1667)
      a.scale = pk.ptr<float>(oSf);
      const size_t oP = sd ? oPR : oPL;
      a.mps = mode == 0 ? (devViews ? (sd ? ex->d_fviewsR.p : ex->d_fviewsL.p) : pk.ptr<orbx_map_point_view>(oP)) + (size_t)f * st : nullptr;
      a.pts = mode == 1 ? pk.ptr<orbx_projected_point>(oP) + (size_t)f * st : nullptr;
      a.nmp = n_points[f]; a.mode = mode; a.checkOri = check_ori;
      a.th = sd ? 1.0f : th;   // the right-camera radius is not scaled by th (:144)
      a.thFar = th_far; a.nnratio = nnratio; a.far = far_points;
      a.occupied = occ + first; a.match = mt + first;
      a.candOff = a.grid.candOff; a.result = sideRes; a.candCap = candCap;
      a.candIdx = pk.ptr<int>(oCx[sd]) + (size_t)f * candCap; a.candDist = pk.ptr<int>(oCd[sd]) + (size_t)f * candCap;
      sides[2 * (size_t)f + sd] = a;
    }
    ProjFeArgs q{};
    const ProjArgs &sl = sides[2 * (size_t)f], &sr = sides[2 * (size_t)f + 1];
    q.offL = sl.candOff; q.idxL = sl.candIdx; q.distL = sl.candDist;
    q.offR = sr.candOff; q.idxR = sr.candIdx; q.distR = sr.candDist;
    q.nLeft = nL[f]; q.n = nL[f] + nR[f]; q.nmp = n_points[f]; q.mode = mode; q.checkOri = check_ori; q.nnratio = nnratio;
    q.mps = sl.mps; q.pts = sl.pts; q.kps = kc;
    q.l2r = pk.ptr<int>(oA12) + (size_t)f * cap; q.r2l = pk.ptr<int>(oA21) + (size_t)f * cap;
    q.occupied = occ; q.match = mt; q.result = pk.ptr<int>(oRes) + (size_t)f * 2;
    q.writes[0] = pk.ptr<int4>(oW0) + (size_t)f * nm; q.writes[1] = pk.ptr<int4>(oW1) + (size_t)f * nm;
    q.writers[0] = pk.ptr<int>(oWl0) + (size_t)f * n2c * kFeWriters; q.writers[1] = pk.ptr<int>(oWl1) + (size_t)f * n2c * kFeWriters;
    q.writers[2] = pk.ptr<int>(oWl2) + (size_t)f * n2c * kFeWriters;
    q.nwriters[0] = pk.ptr<int>(oWc0) + (size_t)f * n2c; q.nwriters[1] = pk.ptr<int>(oWc1) + (size_t)f * n2c;
    q.nwriters[2] = pk.ptr<int>(oWc2) + (size_t)f * n2c;
    q.flags = pk.ptr<int>(oFlags) + (size_t)f * kFlags;
    frames[f] = q;
  }
  e = pk.commit();
  if (e == hipSuccess) e = hipMemsetAsync(pk.ptr<int>(oSideRes), 0, (size_t)F * 4 * sizeof(int), nullptr);
  FeConcatArgs ca{ex->d_kps.p, ex->d_desc.p, ex->d_nOut.p, first_left, first_right, cap, pk.ptr<orbx_keypoint>(oKc), pk.ptr<uint8_t>(oDc)};
  if (e == hipSuccess) e = launch_fe_concat(ca, F, nullptr);
  if (e == hipSuccess)
    e = launch_proj_fisheye_batch(pk.ptr<ProjArgs>(oSd), pk.ptr<ProjFeArgs>(oFr), F, maxPts, maxN, mode, check_ori, kBlind, nullptr);
  std::vector<int> redo;
  if (e == hipSuccess) {
    const uint8_t* h = pk.fetch(oOcc, outBytes, &e);  // synchronises
    if (e == hipSuccess)
      for (int f = 0; f < F; f++) {
        const int n = nL[f] + nR[f];
        int res[2], sres[4];
        std::memcpy(res, h + (oRes - oOcc) + (size_t)f * 2 * sizeof(int), sizeof res);
        std::memcpy(sres, h + (oSideRes - oOcc) + (size_t)f * 4 * sizeof(int), sizeof sres);
        const int* fl = reinterpret_cast<const int*>(h + (oFlags - oOcc)) + (size_t)f * kFlags;
        if (n_points[f] > 0 && n > 0 && (std::max(sres[1], sres[3]) > candCap || fl[1] || fl[40 + kBlind - 1])) {
          redo.push_back(f);
          continue;
        }
        std::memcpy(occupied + (size_t)f * n2c, h + (size_t)f * n2c, n2c);
        std::memcpy(match + (size_t)f * n2c, h + (oMt - oOcc) + (size_t)f * n2c * sizeof(int), (size_t)n * sizeof(int));
        for (int i = n; i < n2c; i++) match[(size_t)f * n2c + i] = -1;
        n_matches[f] = (n_points[f] > 0 && n > 0) ? res[0] : 0;
      }
  }
  pk.release();
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  for (int f : redo) {   // the one-shot path on a host copy of the frame's two cameras
    const int n = nL[f] + nR[f];
    std::vector<orbx_keypoint> k(std::max(n, 1));
    std::vector<uint8_t> d((size_t)std::max(n, 1) * 32);
    HIPC(hipMemcpy(k.data(), ex->d_kps.p + (size_t)(first_left + f) * cap, (size_t)nL[f] * sizeof(orbx_keypoint), hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(k.data() + nL[f], ex->d_kps.p + (size_t)(first_right + f) * cap, (size_t)nR[f] * sizeof(orbx_keypoint), hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(d.data(), ex->d_desc.p + (size_t)(first_left + f) * cap * 32, (size_t)nL[f] * 32, hipMemcpyDeviceToHost));
    HIPC(hipMemcpy(d.data() + (size_t)nL[f] * 32, ex->d_desc.p + (size_t)(first_right + f) * cap * 32, (size_t)nR[f] * 32, hipMemcpyDeviceToHost));
    uint8_t* occ = occupied + (size_t)f * n2c;
    int32_t* mt = match + (size_t)f * n2c;
    if (occupied_in) std::memcpy(occ, occupied_in + (size_t)f * n2c, n2c); else std::memset(occ, 0, n2c);
    for (int i = 0; i < n2c; i++) mt[i] = -1;
    if (devViews) {
      hvL.resize((size_t)st); hvR.resize((size_t)st);
      HIPC(hipMemcpy(hvL.data(), ex->d_fviewsL.p + (size_t)f * st, (size_t)st * sizeof(orbx_map_point_view), hipMemcpyDeviceToHost));
      HIPC(hipMemcpy(hvR.data(), ex->d_fviewsR.p + (size_t)f * st, (size_t)st * sizeof(orbx_map_point_view), hipMemcpyDeviceToHost));
    }
    rc = search_by_projection_fisheye_impl(ex->device, k.data(), d.data(), nL[f], nR[f], min_x, min_y, max_x, max_y, ex->scale.data(),
                                           nlevels, mode == 0 ? (devViews ? hvL.data() : viewsL + (size_t)f * st) : nullptr,
                                           mode == 0 ? (devViews ? hvR.data() : viewsR + (size_t)f * st) : nullptr,
                                           mode == 1 ? ptsL + (size_t)f * st : nullptr, mode == 1 ? ptsR + (size_t)f * st : nullptr,
                                           n_points[f], th, far_points, th_far, nnratio, check_ori,
                                           mode == 0 ? l2r + (size_t)f * cap : nullptr, mode == 0 ? r2l + (size_t)f * cap : nullptr, occ, mt);
    if (rc < 0) return rc;
    n_matches[f] = rc;
  }
  int total = 0;
  for (int f = 0; f < F; f++) total += n_matches[f];
  return total;
}

int fisheye_batch_checks(orbx_extractor* ex, int first_left, int first_right, int n_frames, const int32_t* n_points, int stride,
                         const void* occupied, const void* match, const void* n_matches) {
  if (!ex || n_frames < 0 || first_left < 0 || first_right < 0 || !n_points || stride < 0 || (n_frames && (!occupied || !match || !n_matches)))
    return fail(ORBX_E_BADARG, "bad argument");
  if (n_frames && (ex->lastN <= 0 || first_left + n_frames > ex->lastN || first_right + n_frames > ex->lastN))
    return fail(ORBX_E_BADARG, "frames outside the handle's last batch");
  for (int f = 0; f < n_frames; f++) {
    if (n_points[f] < 0 || n_points[f] > stride) return fail(ORBX_E_BADARG, "n_points[f] outside [0, points_stride]");
    if (n_points[f] > 15000) return fail(ORBX_E_CAPACITY, "more than 15000 points in a frame");
  }
  return ORBX_OK;
}
}  // namespace

int orbx_search_by_projection_fisheye_batch(orbx_extractor* ex, int first_left, int first_right, int n_frames, float min_x, float min_y,
                                            float max_x, float max_y, const orbx_map_point_view* map_points,
                                            const orbx_map_point_right* map_points_right, const int32_t* n_map_points,
                                            int points_stride, float th, int far_points, float th_far_points, float nnratio,
                                            const int32_t* left_to_right, const int32_t* right_to_left, const uint8_t* occupied_in,
                                            uint8_t* occupied, int32_t* match, int32_t* n_matches) {
  int rc = fisheye_batch_checks(ex, first_left, first_right, n_frames, n_map_points, points_stride, occupied, match, n_matches);
  if (rc != ORBX_OK) return rc;
  if (n_frames == 0) return 0;
  if (!left_to_right || !right_to_left) return fail(ORBX_E_BADARG, "null left / right match arrays");
  const int nlevels = ex->prm.nlevels, st = std::max(points_stride, 1);
  int maxPts = 0;
  for (int f = 0; f < n_frames; f++) maxPts = std::max(maxPts, n_map_points[f]);
  if (maxPts && !map_points && !map_points_right) {   // views of both cameras made on the device
    if (ex->fviewsFrames < n_frames || ex->fviewsStride != points_stride)
      return fail(ORBX_E_BADARG, "map_points == NULL needs a preceding orbx_project_map_points_fisheye_batch with the same frames and points_stride == its n");
    for (int f = 0; f < n_frames; f++)
      if (n_map_points[f] != ex->fviewsStride) return fail(ORBX_E_BADARG, "map_points == NULL: n_map_points[f] must equal the uploaded map's n");
    return proj_fisheye_batch_impl(ex, first_left, first_right, n_frames, min_x, min_y, max_x, max_y, 0, nullptr, nullptr, nullptr, nullptr,
                                   n_map_points, points_stride, th, far_points, th_far_points, nnratio, 0, left_to_right, right_to_left,
                                   occupied_in, occupied, match, n_matches);
  }
  if (maxPts && (!map_points || !map_points_right)) return fail(ORBX_E_BADARG, "null map points");
  // the right camera as a second list of views, exactly as the one-shot call builds it (:141-144)
  std::vector<orbx_map_point_view> left((size_t)n_frames * st), right((size_t)n_frames * st);
  for (int f = 0; f < n_frames; f++)
    for (int i = 0; i < n_map_points[f]; i++) {
      const size_t o = (size_t)f * st + i;
      const orbx_map_point_right& r = map_points_right[o];
      left[o] = map_points[o];
      right[o] = map_points[o];
      if ((left[o].in_view && (left[o].predicted_level < 0 || left[o].predicted_level >= nlevels)) ||
          (r.in_view_r && (r.predicted_level_r < -1 || r.predicted_level_r >= nlevels)))
        return fail(ORBX_E_BADARG, "map point with a predicted level outside [0, nlevels)");
      if (!left[o].in_view) left[o].predicted_level = 0;
      right[o].proj_x = map_points[o].proj_xr;
      right[o].proj_y = r.proj_yr;
      right[o].view_cos = r.view_cos_r;
      right[o].predicted_level = r.predicted_level_r < 0 ? 0 : r.predicted_level_r;
      right[o].in_view = (r.in_view_r && r.predicted_level_r != -1) ? 1 : 0;
    }
  return proj_fisheye_batch_impl(ex, first_left, first_right, n_frames, min_x, min_y, max_x, max_y, 0, left.data(), right.data(), nullptr,
                                 nullptr, n_map_points, points_stride, th, far_points, th_far_points, nnratio, 0, left_to_right,
                                 right_to_left, occupied_in, occupied, match, n_matches);
}

int orbx_search_by_projection_frame_fisheye_batch(orbx_extractor* ex, int first_left, int first_right, int n_frames, float min_x,
                                                  float min_y, float max_x, float max_y, const orbx_projected_point* points,
                                                  const float* uv_right, const int32_t* n_points, int points_stride,
                                                  int check_orientation, const uint8_t* occupied_in, uint8_t* occupied,
                                                  int32_t* match, int32_t* n_matches) {
  int rc = fisheye_batch_checks(ex, first_left, first_right, n_frames, n_points, points_stride, occupied, match, n_matches);
  if (rc != ORBX_OK) return rc;
  if (n_frames == 0) return 0;
  const int st = std::max(points_stride, 1);
  int maxPts = 0;
  for (int f = 0; f < n_frames; f++) maxPts = std::max(maxPts, n_points[f]);
  if (maxPts && (!points || !uv_right)) return fail(ORBX_E_BADARG, "null points");
  std::vector<orbx_projected_point> right((size_t)n_frames * st);
  for (int f = 0; f < n_frames; f++)
    for (int i = 0; i < n_points[f]; i++) {
      const size_t o = (size_t)f * st + i;
      right[o] = points[o];
      right[o].u = uv_right[2 * o];
      right[o].v = uv_right[2 * o + 1];
    }
  return proj_fisheye_batch_impl(ex, first_left, first_right, n_frames, min_x, min_y, max_x, max_y, 1, nullptr, nullptr, points,
                                 right.data(), n_points, points_stride, 1.0f, 0, 0.f, 0.f, check_orientation, nullptr, nullptr,
                                 occupied_in, occupied, match, n_matches);
}

}  // extern "C"
