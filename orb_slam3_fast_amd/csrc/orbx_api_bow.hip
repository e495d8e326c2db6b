// orbx_api_bow.hip — C ABI of the bag of words: the vocabulary handle, its ORBvoc.txt loader, ComputeBoW (one frame and batched).
#include "orbx_host.h"

extern "C" {

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

}  // extern "C"
