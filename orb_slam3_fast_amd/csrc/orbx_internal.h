// orbx_internal.h — shared host/device structures of liborbx (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "../../include/orbx.h"

namespace orbx {

constexpr int kEdge = 19;        // EDGE_THRESHOLD, src/ORBextractor.cc:73
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE, :72
constexpr int kBorder = 16;      // EDGE_THRESHOLD - 3: origin of the FAST window, :893
constexpr int kMaxIni = 8;       // max initial quadtree roots supported (aspect <= 8.49)

// Per pyramid level geometry (derived on the host for the current (w, h), passed to kernels by value).
struct LevelDev {
  int w, h, pitch;               // pitch of the internal buffers (level 0 input uses Pyr::l0RowPitch)
  int nCols, nRows, wCell, hCell;  // FAST cell grid, src/ORBextractor.cc:904-907
  int cellStart;                 // number of cells in coarser-indexed levels before this one
  int quota;                     // mnFeaturesPerLevel[l]
  int candCap;                   // capacity of the dense candidate list of this level (per image)
  int cellCap;                   // slots per FAST cell in the sparse per-cell candidate store
  int selOff, selCap;            // slot range of this level in the per-image selected-keypoint block
  int xcoef, ycoef;              // offsets into the resize coefficient tables (level l from l-1)
  long long off;                 // byte offset of the level inside one image's pyramid block
  long long candOff;             // entry offset of the dense candidate list inside one image's block
  long long cellOff;             // entry offset of this level's per-cell slots inside one image's block
  float scale;                   // mvScaleFactor[l]
  float patch;                   // (float)(int)(31 * scale), :984
};

struct Geom {
  int nlevels, totalCells, iniTh, minTh;
  int tileP, tileH;              // LDS image-tile pitch / rows of the detect kernel
  int scoreP, scoreH;            // LDS score-tile pitch / rows
  int listCap;                   // max pixels in one cell's detectable window
  int selImg;                    // selected-keypoint slots per image (sum of selCap)
  int outCap;                    // output keypoint capacity per image
  int cv440;                     // 0: Gaussian taps of OpenCV >= 4.5.1; 1: of OpenCV 4.0 .. 4.5.0 (orbx_set_opencv_compat), every column rounded;
                                 // 16 / 32: the same taps with the flooring 16- / 32-column vector body of those releases' vertical pass
  long long pyrImg;              // bytes per image of the internal pyramid block
  long long candImg;             // dense candidate entries per image
  long long cellImg;             // per-cell slot entries per image
  int levelCell[ORBX_MAX_LEVELS];  // lv[l].cellStart again, contiguous (INT_MAX past the last level): k_detect's level lookup
  LevelDev lv[ORBX_MAX_LEVELS];
};

// Where the pyramid of the current batch lives.  Level 0 aliases the caller's images.
struct Pyr {
  const uint8_t* l0;
  long long l0Row, l0Img;
  uint8_t* pyr;                  // levels >= 1 (and an unused level-0 slot) : [B][pyrImg]
  uint8_t* blur;                 // blurred copies of all levels: [B][pyrImg]
};

__host__ __device__ inline const uint8_t* level_ptr(const Geom& g, const Pyr& p, int img, int l, int& pitch) {
  if (l == 0) {
    pitch = (int)p.l0Row;
    return p.l0 + (long long)img * p.l0Img;
  }
  pitch = g.lv[l].pitch;
  return p.pyr + (long long)img * g.pyrImg + g.lv[l].off;
}

// Candidate / selected keypoint packing: x (12 bits) | y (12 bits) << 12 | response << 24.
// Candidates: x, y relative to the (16,16) FAST window origin.  Selected: level coordinates.
__host__ __device__ inline uint32_t pack_key(int x, int y, int r) {
  return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)r << 24);
}
__host__ __device__ inline int key_x(uint32_t k) { return k & 0xFFF; }
__host__ __device__ inline int key_y(uint32_t k) { return (k >> 12) & 0xFFF; }
__host__ __device__ inline int key_r(uint32_t k) { return k >> 24; }

// Fused small levels of the resize chain (orbx_resize_tail.hip): levels lA .. lA+nT-1 in one launch, a workgroup per band of
// rows of the segment's last level.  bands[band * (nT + 1) + 0] = rows of level lA-1 the band stages; [.. + 1 + t] = rows of
// level lA+t it computes (first..last) and writes to the pyramid (first..ownEnd-1).
struct TailBand {
  int first, last, ownEnd, pad_;
};
struct TailPlan {
  int lA, nT, nBands, bandOff;         // bandOff: the segment's first entry in the handle's band table (host side)
  int tileBytes[2];                    // ping-pong LDS tiles (even / odd cascade position)
  int pitch[ORBX_MAX_LEVELS + 1];      // LDS row pitch of cascade position t (0 = level lA-1)
  int rtOff[ORBX_MAX_LEVELS];          // first row-table entry of fused level t
  int rtTotal;                         // row-table entries of all fused levels
  unsigned ldsBytes;
};
hipError_t launch_resize_tail(const Geom& g, const Pyr& p, const TailPlan& tp, const TailBand* bands, int img0, int nimg,
                              const uint4* xtab, const int* yofs, const short* yab, hipStream_t s);
hipError_t prepare_resize_tail(unsigned ldsBytes);
hipError_t raise_dynamic_lds(const void* fn, size_t bytes);   // per-device running maximum of a kernel's dynamic-LDS limit

// Launch wrappers (orbx_kernels.hip).  All enqueue on `s` and return the HIP status.
// whole single-channel frames through the pyramid's resize kernel (pre-processing plans; orbx_kernels.hip)
size_t resize_plain_lds(int sw, int sh, int dw, int dh);
hipError_t prepare_resize_plain(int sw, int sh, int dw, int dh);
hipError_t launch_resize_plain(const uint8_t* src, int sw, int sh, long long sp, long long sip, uint8_t* dst, int dw, int dh,
                               long long dp, long long dip, const uint4* xtab, const uint32_t* yrow, const short* yab, int nimg,
                               hipStream_t s);
hipError_t launch_resize(const Geom& g, const Pyr& p, int nimg, int level, const uint4* xtab, const uint32_t* yofs /* clamped row pairs */,
                         const short* yab, hipStream_t s);
hipError_t launch_detect(const Geom& g, const Pyr& p, int nimg, uint32_t* cellCand, int* cellCount, int level0,
                         int level1, uint8_t* dbgScore, hipStream_t s);
hipError_t launch_octree(const Geom& g, int nimg, const uint32_t* cellCand, const int* cellCount, int* cellPrefix,
                         uint32_t* cand, int* candCount, uint16_t* knode, uint32_t* sel, int* selCount, int level0,
                         int level1, hipStream_t s);
hipError_t launch_blur(const Geom& g, const Pyr& p, int nimg, int level0, int level1, hipStream_t s);
hipError_t launch_slots(const Geom& g, int nimg, const uint32_t* sel, const int* selCount, const int* lap,
                        int* slot, int* nOut, int* mono, hipStream_t s);
hipError_t launch_describe(const Geom& g, const Pyr& p, int nimg, const uint32_t* sel, const int* selCount,
                           const int* slot, orbx_keypoint* kps, uint8_t* desc, int* nOut, int* mono, hipStream_t s);
size_t octree_lds_bytes(const Geom& g);
size_t resize_lds_bytes(const Geom& g);

struct StereoArgs {
  const orbx_keypoint *kL, *kR;
  const uint8_t *dL, *dR;
  const int *nL, *nR;            // per image counts
  int capL, capR;                // per-image strides (entries)
  int firstL, firstR;
  float bf, b;
  float* uRight;                 // [pairs][capL]
  float* depth;
  int* sad;                      // [pairs][capL] best SAD or -1
  int* rowStart;                 // [pairs][2][imgH + 1] CSR of the left / right keypoints by integer row (k_stereo_sort)
  uint4* srec;                   // [pairs][2][cap] row-sorted records {x, y, octave | index << 8, minr | maxr << 16}
  uint4* sdesc;                  // [pairs][2][cap][2] the descriptors in the same order
  int cap;                       // max(capL, capR): stride of srec / sdesc
  int imgH;                      // level-0 height
  int band;                      // max rows a right keypoint's +-2*scale band can be away from its own row
};
hipError_t launch_stereo_sort(const Geom& g, const StereoArgs& a, int npairs, hipStream_t s);
// direct: k_stereo_band selects its keypoints from the unsorted arrays itself (no launch_stereo_sort in front; needs stereo_direct_ok)
bool stereo_direct_ok(const StereoArgs& a, int npairs);
void debug_set_stereo_direct(int max_pairs);
struct ResultPack;
// pack (direct form, one pair): extra workgroups of the launch gather both eyes' keypoints and descriptors into the host block
hipError_t launch_stereo_match(const Geom& g, const Pyr& pl, const Pyr& pr, const StereoArgs& a, int npairs,
                               hipStream_t s, bool direct = false, const ResultPack* pack = nullptr);
hipError_t launch_stereo_filter(const StereoArgs& a, int npairs, hipStream_t s);
hipError_t launch_bf_knn2(const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, int* idx2, int* dist2,
                          uint8_t* ok, hipStream_t s);

// Frame::ComputeStereoFishEyeMatches tail (src/Frame.cc:1298-1330) on the k_bf_knn2 results of the lapping rows.
struct FisheyeArgs {
  const orbx_keypoint* kL;  // all left keypoints (nL), lapping rows start at monoL
  const orbx_keypoint* kR;
  int nL, nR, monoL, monoR;
  const int* idx2;          // [nL - monoL][2] from k_bf_knn2
  const uint8_t* ratioOk;   // [nL - monoL]
  orbx_kb8_rig rig;
  const float* sigma2;
  int nLevels;
  int* leftToRight;         // [nL]  (pre-set to -1)
  int* rightToLeft;         // [nR]  (pre-set to -1; atomicMax = "last left index wins")
  float* depth;             // [nL]  (pre-set to -1)
  float* p3D;               // [nL][3] (pre-set to 0)
  int* counters;            // [0] = nMatches, [1] = descMatches (pre-set to 0)
};
hipError_t launch_fisheye_triangulate(const FisheyeArgs& a, hipStream_t s);

// Batched, device-resident Frame::ComputeStereoFishEyeMatches: pair p = image firstL + p of the left extraction against
// image firstR + p of the right one; lapping rows [mono, n) come from the extractors' device-side counts.
struct FisheyeBatchArgs {
  const orbx_keypoint* kL;  // [images][capL]
  const orbx_keypoint* kR;
  const uint8_t* dL;        // [images][capL][32]
  const uint8_t* dR;
  const int* nL;            // per image: keypoint count / monoIndex
  const int* nR;
  const int* monoL;
  const int* monoR;
  int capL, capR, firstL, firstR;
  orbx_kb8_rig rig;
  float sigma2[ORBX_MAX_LEVELS];
  int nLevels;
  int* leftToRight;         // [pairs][capL]  (pre-set to -1)
  int* rightToLeft;         // [pairs][capR]  (pre-set to -1)
  float* depth;             // [pairs][capL]  (pre-set to -1)
  float* p3D;               // [pairs][capL][3] (pre-set to 0)
  int* counters;            // [pairs][2] = nMatches, descMatches (pre-set to 0)
  uint2* cand;              // [pairs * capL] scratch: accepted (pair, iL | iR << 16) of the scan, read by the triangulation launch
  int* candCount;           // its length (pre-set to 0)
};
hipError_t launch_fisheye_batch(const FisheyeBatchArgs& a, int npairs, hipStream_t s);

// cv::undistortPoints (5 iterations, P = K) on n interleaved float pairs; k = 12 coefficients as doubles' source floats.
struct UndistortArgs {
  const float* in;
  float* out;
  int n, stride;  // stride in floats between consecutive points (2 = packed pairs, 7 = orbx_keypoint records)
  float K[4];
  float k[12];
  int hasDist;
};
hipError_t launch_undistort(const UndistortArgs& a, hipStream_t s);

// Pre-processing: gray conversion and a generic (any size, 1/3/4 channels) bilinear resize; pitches in bytes.
// ---- bag of words ---------------------------------------------------------------------------------------
// DBoW2 vocabulary tree on the device: children of node i = children[childStart[i] .. childStart[i + 1]) in file order.
struct BowVoc {
  const int* childStart; const int* children; const uint32_t* desc; const double* weight; const int* wordId;
  int L, nNodes, scoring, weighting;
};
constexpr int kBowMaxFeatures = 8192;  // per image: the assembly sorts (word, index) pairs in LDS
// Per image i of a batch: descriptors at desc + i * descImgPitch, count = counts ? counts[i] : n.  Per-feature results
// (word / weight / node, `cap` entries per image) and the assembled vectors (cap entries per image, nodeStart cap + 1,
// outCounts 3 per image = words, nodes, features).
struct BowArgs {
  BowVoc voc;
  const uint8_t* desc; long long descImgPitch; const int* counts; int n, cap, levelsup;
  int* word; double* weight; int* node;
  uint32_t* words; double* values; uint32_t* nodes; int* nodeStart; uint32_t* feats; int* outCounts;
};
hipError_t launch_bow_transform(const BowArgs& a, int nimg, hipStream_t s);
// ORBmatcher::SearchByBoW(KeyFrame*, Frame&): feature vectors as CSR, match[nF] = keyframe feature or -1,
// flags: [0] matches made, [1] matches culled, [2..32) rotation histogram; result[0] = nmatches.
struct BowMatchArgs {
  const uint32_t* kfNodes; const int* kfStart; const uint32_t* kfFeat; int nKfNodes;
  const uint32_t* kfDesc; const orbx_keypoint* kfKps; const uint8_t* kfValid;
  const uint32_t* fNodes; const int* fStart; const uint32_t* fFeat; int nFNodes;
  const uint32_t* fDesc; const orbx_keypoint* fKps; int nF, nLeftF;
  float nnratio; int checkOri;
  int* match; int* bin; int* flags; int* result;
};
hipError_t launch_bow_match(const BowMatchArgs& a, hipStream_t s);
hipError_t launch_bow_match_batch(const BowMatchArgs* d_frames, int nFrames, int maxKfNodes, int maxNF, int checkOri, hipStream_t s);

// Results of up to two images gathered into the handle's PINNED host block by one kernel (the single-frame host entries
// orbx_extract / orbx_extract_stereo): counts and mono indices, then the first n_i keypoint records / descriptor rows of every
// image and the left image's uRight / depth -- count-trimmed, written straight over PCIe, instead of six D2H copies.
struct ResultPack {
  const int* nOut; const int* mono;
  const uint32_t* kps; const uint32_t* desc; const uint32_t* uR; const uint32_t* depth;   // device arrays ([img][cap] rows)
  uint32_t* hCnt; uint32_t* hKps; uint32_t* hDesc; uint32_t* hUr; uint32_t* hDepth;        // host-block sections (device-visible)
  int nimg, cap, stereo;
  int mask;      // sections to gather: bit 0 / 1 keypoints of image 0 / 1, 2 / 3 descriptors, 4 uRight, 5 depth, 6 counts
  int fixedN;    // >= 0: copy this many uRight / depth entries instead of nOut[0] (orbx_stereo_download: the caller's capacity)
  // launch_stereo_match's gather workgroups: arrival counter (device, zero between frames), and the word of the host block that
  // receives `seq` once counts, keypoints and descriptors of both eyes are in the block (the host copies them out meanwhile)
  int* packCtr;
  uint32_t* hFlag;
  uint32_t seq;
};
hipError_t launch_result_pack(const ResultPack& a, hipStream_t s);
// (orbx_stereo.hip)  stereoOnly: keypoints / descriptors are already in the host block (launch_stereo_match's pack workgroups)
hipError_t launch_stereo_filter_pack(const StereoArgs& sa, const ResultPack& a, hipStream_t s, bool stereoOnly = false);

// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:886-1106), single-camera key frames (k_tri_*)
struct TriArgs {
  const uint32_t* nodes1; const int* start1; const uint32_t* feat1; int nNodes1, nList1;  // pKF1->mFeatVec as CSR
  const uint32_t* nodes2; const int* start2; const uint32_t* feat2; int nNodes2;          // pKF2->mFeatVec
  const orbx_keypoint* k1; const orbx_keypoint* k2;   // mvKeysUn
  const uint32_t* d1; const uint32_t* d2;             // mDescriptors
  const uint8_t* mp1; const uint8_t* mp2;             // GetMapPoint(idx) != NULL
  const float* ur1; const float* ur2;                 // mvuRight or nullptr
  int n1, n2;
  const float* scale2; const float* sigma2;           // pKF2->mvScaleFactors, mvLevelSigma2
  float ep0, ep1;                                     // epipole in image 2 (:901)
  float F[9];                                         // F12 row-major (Pinhole.cpp:133)
  int onlyStereo, coarse, checkOri;
  float nnratio; // SearchByBoW(KeyFrame*, KeyFrame*) only
  int nLeft1, nLeft2; const float* sigma1; const orbx_tri_rig* rig;  // two-camera rigs only (k_tri_match_rig): NLeft, pKF1's mvLevelSigma2
  int* match;    // n1: vMatches12
  int* flags;    // [0] accepted, [1] removed, [2..31] rotation histogram, [32] a node's list exceeded kBowNodeCap
  int* result;   // nmatches
};
hipError_t launch_search_for_triangulation(const TriArgs& a, hipStream_t s);
hipError_t launch_tri_match_rig(const TriArgs& a, hipStream_t s);  // orbx_stereo.hip (next to kb8_triangulate_matches)
// ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (src/ORBmatcher.cc:766-884) on the same argument block:
// mp1 / mp2 = "holds a good map point" flags, match = the feature of pKF2 whose map point vpMatches12[idx1] receives
hipError_t launch_search_by_bow_keyframes(const TriArgs& a, hipStream_t s);

// cv::remap INTER_LINEAR with float maps (k_remap): batch of nimg images, image i uses map i % nMaps.
struct RemapArgs {
  const uint8_t* src; int sw, sh, cn; long long srcPitch, srcImgPitch;
  const float* mapx; const float* mapy; long long mapPitch, mapImgPitch; int nMaps;  // pitches in floats
  uint8_t* dst; int dw, dh; long long dstPitch, dstImgPitch;
  int mapVec4, dstVec4;  // 16-byte aligned map rows / 4-byte aligned destination rows
  const int* tileTab;    // k_remap_lds: per (map, tile) footprint entries of remap_tile_table (nullptr: k_remap1 / k_remap)
  int tilesX, tilesY;
};
bool remap_tile_table(const float* mapx, const float* mapy, long long mapStride, int dw, int dh, int sw, int sh, int nMaps,
                      std::vector<int>& tab, int& tilesX, int& tilesY);
void debug_set_remap_lds(int on);
hipError_t launch_remap(const RemapArgs& a, int nimg, hipStream_t s);
// cv::CLAHE::apply (k_clahe_lut + k_clahe_apply)
struct ClaheArgs {
  const uint8_t* src; int w, h; long long srcPitch, srcImgPitch;
  uint8_t* dst; long long dstPitch, dstImgPitch;
  uint8_t* lut;  // [nimg][tilesY * tilesX][256]
  int tilesX, tilesY, tw, th, clip; float lutScale, invTw, invTh;
  int srcVec4, dstVec4;
};
// cells: clahe_cells_bytes() of scratch for the packed cell tables (nullptr: per-pixel lut gathers)
size_t clahe_cells_bytes(const ClaheArgs& a, int nimg);
hipError_t launch_clahe(const ClaheArgs& a, int nimg, uint32_t* cells, hipStream_t s);
hipError_t launch_cvt_gray(const uint8_t* src, int w, int h, long long sp, long long sip, int cn, int rgb, uint8_t* dst,
                           long long dp, long long dip, int nimg, hipStream_t s);
hipError_t launch_resize_generic(const uint8_t* src, int sw, int sh, long long sp, long long sip, int cn, uint8_t* dst, int dw,
                                 int dh, long long dp, long long dip, const int* xofs, const short* xab, const int* yofs,
                                 const short* yab, int nimg, hipStream_t s);
constexpr int kFeWriters = 4;  // writers / claimers remembered per slot and round by the fixed-point resolves
struct InitArgs {
  const orbx_keypoint *k1, *k2;
  const uint8_t *d1, *d2;
  int n1, n2;
  float minX, minY, invW, invH;
  float* prev;                   // 2*n1
  int* matches12;                // n1
  int window;
  float nnratio;
  int checkOri;
  // scratch
  int* cellStart;                // 64*48+1
  int* cellItems;                // n2
  int* candOff;                  // n1+1
  int* candIdx;                  // candidate i2 lists
  int* candDist;
  int candCap;
  int* matchedDist;              // n2
  int* matches21;                // n2
  int* result;                   // [0]=nmatches, [1]=overflow flag
  // parallel fixed-point rounds (k_init_round): claims of every F1 keypoint and per-F2-keypoint claimer lists
  int2* claim[2];                // n1 each: (i2, dist) or (-1, 0)
  int2* claimers[3];             // n2 * kFeWriters each: (i1, dist) of the points that claimed i2 in that round
  int* nclaimers[3];             // (rotating, see k_init_round); n2 each
  int* flags;                    // [0] changed, [1] claimer-list overflow, [2] claimed slots, [3] removed, [4..33] histogram
};
hipError_t launch_search_init_cands_fill(const InitArgs& a, hipStream_t s);
hipError_t launch_search_init_rounds(const InitArgs& a, int first_round, int rounds, hipStream_t s);
hipError_t launch_search_init_finish(const InitArgs& a, int last_round, hipStream_t s);
hipError_t launch_search_init_resolve_serial(const InitArgs& a, hipStream_t s);
hipError_t launch_search_init_batch(const InitArgs* d_frames, int nFrames, int maxN1, int maxN2, int rounds, hipStream_t s);
hipError_t launch_grid_build(const InitArgs& a, hipStream_t s);        // AssignFeaturesToGrid of (k2, n2) as CSR
hipError_t launch_area_query(const InitArgs& a, const float* q, int nq, int* qOff, int* out, int pass, hipStream_t s);
hipError_t launch_scan_offsets(const InitArgs& a, hipStream_t s);
hipError_t launch_search_init(const InitArgs& a, hipStream_t s);       // grid + candidate counts + scan
hipError_t launch_search_init_fill(const InitArgs& a, hipStream_t s);  // candidate fill + serial resolve
// ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, ...) — pinhole case.
struct ProjArgs {
  InitArgs grid;                  // k2/n2/minX/minY/invW/invH/cellStart/cellItems describe the frame F
  const uint8_t* desc;            // F.mDescriptors
  const float* uRight;            // F.mvuRight or nullptr
  const float* scale;             // F.mvScaleFactors
  const orbx_map_point_view* mps;        // mode 0: local map points
  const orbx_projected_point* pts;       // mode 1: projected LastFrame points
  int mode, checkOri;
  int maxDist;                    // acceptance threshold on the best Hamming distance: TH_HIGH (100), or ORBdist of the
                                  // relocalisation flavour SearchByProjection(Frame&, KeyFrame*, ...) (src/ORBmatcher.cc:1886)
  int claimAll;                   // 1: every assignment occupies its keypoint (the gate there is a plain non-null test,
                                  // src/ORBmatcher.cc:1871) and a culled slot is free again; 0: only points with observations do
  int nmp;
  float th, thFar, nnratio;
  int far;
  uint8_t* occupied;              // n2, in/out
  int* match;                     // n2
  int* candOff;                   // nmp + 1
  int* candIdx;                   // candidate keypoint index
  int* candDist;                  // (dist << 8) | octave
  int candCap;
  int* result;                    // [0] = nmatches
  // parallel fixed-point resolve (launch_proj_resolve_parallel): scratch
  int* taker[3];                  // n2 each: smallest point index that (with observations) claims the keypoint; round r
                                  // reads taker[r % 3], writes taker[(r + 1) % 3] and clears taker[(r + 2) % 3]
  int* choice;                    // nmp: chosen keypoint (-1 none)
  int* flags;                     // [0] changed in the last round, [1] accepted, [2] removed, [3..32] orientation histogram
};
hipError_t launch_proj_count(const ProjArgs& a, hipStream_t s);   // grid + candidate counts + scan
hipError_t launch_proj_fill(const ProjArgs& a, hipStream_t s);    // candidate fill + serial resolve
hipError_t launch_proj_cands_fill(const ProjArgs& a, hipStream_t s);                 // candidate fill only
hipError_t launch_proj_rounds(const ProjArgs& a, int first_round, int rounds, hipStream_t s);  // fixed-point rounds
hipError_t launch_proj_finish(const ProjArgs& a, int last_round, hipStream_t s);     // match / occupied / cull / count
hipError_t launch_proj_resolve_serial(const ProjArgs& a, hipStream_t s);             // the one-wave walk (fallback)
// on-device Frame::isInFrustum for (map point, frame) pairs -> orbx_map_point_view records [nFrames][n] (orbx_guided.hip)
struct MapProjArgs {
  const float *pos, *normal, *minDist, *maxDist;
  const uint8_t *desc, *flags, *skip;
  const orbx_frame_pose* poses;
  orbx_map_point_view* views;
  int n, nlevels;
  float minX, minY, maxX, maxY, viewCosLimit, logScaleFactor;
};
hipError_t launch_project_map(const MapProjArgs& a, int nFrames, hipStream_t s);
// k_project_map_kb8: Frame::isInFrustumChecks for both cameras of stereo-fisheye frames (orbx_stereo.hip, beside the KB8 model)
struct MapProjKb8Args {
  const float *pos, *normal, *minDist, *maxDist;
  const uint8_t *desc, *flags, *skip;
  const orbx_frame_pose_kb8 *posesL, *posesR;
  orbx_map_point_view *viewsL, *viewsR;   // the matcher's two lists (left camera; right camera with proj_x / proj_y = mTrackProjXR / YR)
  int n, nlevels;
  float minX, minY, maxX, maxY, viewCosLimit, logScaleFactor;
};
hipError_t launch_project_map_kb8(const MapProjKb8Args& a, int nFrames, hipStream_t s);
// k_project_last: the projection block of SearchByProjection(CurrentFrame, LastFrame) (src/ORBmatcher.cc:1606-1669) on the device
struct LastProjArgs {
  const float* pos;              // [frames][stride][3]
  const int* octave;             // [frames][stride]
  const float* angle;
  const uint8_t *desc, *flags;   // [frames][stride][32] / [frames][stride]
  const int* npts;               // [frames]
  const orbx_frame_pose_q* poses;
  const float* scale;            // mvScaleFactors
  orbx_projected_point* views;   // [frames][stride]
  int stride, nlevels;
  float minX, minY, maxX, maxY, th;
};
hipError_t launch_project_last(const LastProjArgs& a, int nFrames, hipStream_t s);
hipError_t launch_proj_batch(const ProjArgs* d_frames, int nFrames, int maxPts, int maxN2, int mode, int checkOri, int rounds,
                             hipStream_t s);                                        // all frames of a batch, one launch per kernel

// Search part of ORBmatcher::Fuse(pKF, vpMapPoints, th, bRight) (src/ORBmatcher.cc:1195-1256): k_fuse_search
struct FuseArgs {
  InitArgs grid;                 // k2/n2/minX/minY/invW/invH/cellStart/cellItems describe the key frame's camera
  const uint32_t* desc;          // mDescriptors rows of that camera
  const float* uRight;           // mvuRight or nullptr
  const float* invSigma2;        // mvInvLevelSigma2
  const orbx_fuse_point* pts;
  int npts;
  int maxDist;                   // TH_LOW (Fuse) or TH_HIGH (SearchBySim3)
  int* bestIdx;                  // npts
  int* bestDist;                 // npts
  int* result;                   // [0] = nFused
};
hipError_t launch_fuse_search(const FuseArgs& a, hipStream_t s);


// Stereo-fisheye resolve: one wave walks the points, left then right camera, with the partner-slot assignments.
struct ProjFeArgs {
  const int *offL, *idxL, *distL;   // candidate lists against the left / right grid (k_proj_cands)
  const int *offR, *idxR, *distR;
  int nLeft, n, nmp, mode, checkOri;
  float nnratio;
  const orbx_map_point_view* mps;   // mode 0
  const orbx_projected_point* pts;  // mode 1
  const orbx_keypoint* kps;         // n = nLeft + nRight
  const int *l2r, *r2l;             // mode 0
  uint8_t* occupied;                // n
  int* match;                       // n
  int* result;
  // parallel fixed-point rounds (k_proj_round_fe): tentative writes of every point and per-slot writer lists
  int4* writes[2];                  // nmp each: slots written by the point {left best, its partner, right best, its partner} or -1
  int* writers[3];                  // n * kFeWriters each: points that wrote the slot in that round (rotating, see
  int* nwriters[3];                 // k_proj_round_fe); n each
  int* flags;                       // [0] changed, [1] writer-list overflow, [2] writes, [3] removed, [4..33] histogram
};
hipError_t launch_proj_resolve_fisheye(const ProjFeArgs& a, hipStream_t s);
hipError_t launch_proj_rounds_fisheye(const ProjFeArgs& a, int first_round, int rounds, hipStream_t s);
hipError_t launch_proj_finish_fisheye(const ProjFeArgs& a, int last_round, hipStream_t s);
struct FeConcatArgs {   // k_fe_concat: [left | right] keypoints / descriptors of the two-camera frames of an extraction batch
  const orbx_keypoint* kps; const uint8_t* desc; const int* nOut;   // the handle's result arrays ([image][cap] rows) and counts
  int firstL, firstR, cap;
  orbx_keypoint* kcat; uint8_t* dcat;                               // [frame][2 cap] rows
};
hipError_t launch_fe_concat(const FeConcatArgs& a, int nFrames, hipStream_t s);
hipError_t launch_proj_fisheye_batch(const ProjArgs* d_sides, const ProjFeArgs* d_frames, int nFrames, int maxPts, int maxN, int mode,
                                     int checkOri, int rounds, hipStream_t s);
hipError_t prepare_kernels(const Geom& g);                             // raises the dynamic-LDS limits
hipError_t prepare_detect(const Geom& g);                              // (orbx_detect.hip)
void debug_introsort_host(uint64_t* v, int n);
void debug_set_detect_list_cap(int cap);
void debug_set_clahe_cell_kernel(int on);
void debug_set_octree_global(int on);
hipError_t launch_debug_sort(uint64_t* d_v, int n, hipStream_t s);
hipError_t launch_debug_sincos(const float* ang, int n, int fused, float* s, float* c, hipStream_t st);

}  // namespace orbx
