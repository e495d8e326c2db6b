// orbx_api_comm.hip — the one exchange step of the batched many-camera mode (BASELINE config C5): RCCL
// all-gather of every GPU's descriptor blocks, behind the C ABI and zero-copy.
//
// The extraction results already sit in HBM as two dense arrays per handle -- desc [n_images][cap][32] u8 (written by
// k_describe's epilogue) and counts [n_images] i32 -- so the "block {int32 n; u8 desc[cap][32]}" of SURVEY 8e is gathered
// as its two members: ONE grouped RCCL call (ncclGroupStart .. ncclGroupEnd) with two ncclAllGather straight from those
// arrays into the caller's destination arrays, enqueued on the handle's own stream behind the extraction.  No pack
// kernel, no staging copy, no host synchronisation.
//
// RCCL is bound at first use with dlopen("librccl.so.1") rather than at link time: a process that already holds an RCCL
// (PyTorch bundles one under the same SONAME) shares it, a plain C++ SLAM process gets /opt/rocm/lib's through
// liborbx's rpath, and processes that never gather (everything else in the ABI) do not map the 0.5 GB library.
// The reference has no counterpart (single rig, process-global Frame statics, include/Frame.h:230-235).
#include <dlfcn.h>

#include <chrono>
#include <mutex>
#include <thread>

#include "orbx_host.h"

using namespace orbx_host;

namespace {
// the slice of rccl.h this file needs (ABI-stable NCCL 2 entry points)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2 };
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) {
      r.err = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed");
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.so, n);
      if (!p) r.err += std::string(" missing symbol ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &r;
}
#define RCCLC(expr)                                                                                   \
  do {                                                                                                \
    ncclResult_t e_ = (expr);                                                                         \
    if (e_ != ncclSuccess)                                                                            \
      return fail(ORBX_E_HIP, std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(e_) : "RCCL error")); \
  } while (0)
}  // namespace

// ORDERING RULE.  One communicator per rank serves every handle of that rank.  RCCL matches the collectives of a
// communicator by ISSUE ORDER, so every rank must call orbx_allgather_descriptors in the same order (bench.py: step i uses
// handle i mod H on every rank), and two collectives of one communicator must not run concurrently from two streams: each
// call therefore waits (stream-side, hipStreamWaitEvent -- the host never blocks) for the event the previous call of this
// communicator recorded behind its collective.  The chain is what makes "three handles, three streams" legal by construction
// instead of by luck; `seq` counts the calls so that a rank-order mismatch can be reported (orbx_comm_wait).
struct orbx_comm {
  ncclComm_t comm = nullptr;
  int device = 0, ranks = 0, rank = 0;
  std::mutex mu;
  hipEvent_t last = nullptr;   // behind the most recent collective of this communicator
  bool lastValid = false;
  unsigned long long seq = 0;  // collectives enqueued so far
};

extern "C" {

int orbx_comm_unique_id(uint8_t id[ORBX_COMM_ID_BYTES]) {
  if (!id) return fail(ORBX_E_BADARG, "null id");
  Rccl* R = rccl();
  if (!R->err.empty()) return fail(ORBX_E_UNSUPPORTED, R->err);
  static_assert(sizeof(ncclUniqueId) == ORBX_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  RCCLC(R->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return ORBX_OK;
}

int orbx_comm_create(const uint8_t id[ORBX_COMM_ID_BYTES], int n_ranks, int rank, int device, orbx_comm** out) {
  if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(ORBX_E_BADARG, "bad communicator arguments");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return fail(ORBX_E_NODEVICE, "no HIP device: RCCL needs one GPU per rank");
  if (device < 0 || device >= nd) return fail(ORBX_E_BADARG, "device index out of range");
  Rccl* R = rccl();
  if (!R->err.empty()) return fail(ORBX_E_UNSUPPORTED, R->err);
  HIPC(hipSetDevice(device));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  std::unique_ptr<orbx_comm> c(new (std::nothrow) orbx_comm);
  if (!c) return fail(ORBX_E_HIP, "out of memory");
  RCCLC(R->CommInitRank(&c->comm, n_ranks, u, rank));
  HIPC(hipEventCreateWithFlags(&c->last, hipEventDisableTiming));
  c->device = device;
  c->ranks = n_ranks;
  c->rank = rank;
  *out = c.release();
  return ORBX_OK;
}

int orbx_comm_adopt(void* nccl_comm, int device, orbx_comm** out) {
  if (!nccl_comm || !out) return fail(ORBX_E_BADARG, "null argument");
  Rccl* R = rccl();
  if (!R->err.empty()) return fail(ORBX_E_UNSUPPORTED, R->err);
  std::unique_ptr<orbx_comm> c(new (std::nothrow) orbx_comm);
  if (!c) return fail(ORBX_E_HIP, "out of memory");
  c->comm = reinterpret_cast<ncclComm_t>(nccl_comm);
  RCCLC(R->CommCount(c->comm, &c->ranks));
  RCCLC(R->CommUserRank(c->comm, &c->rank));
  c->device = -1 - device;  // negative: not owned, orbx_comm_destroy leaves the ncclComm_t alone
  if (device >= 0) HIPC(hipSetDevice(device));
  HIPC(hipEventCreateWithFlags(&c->last, hipEventDisableTiming));
  *out = c.release();
  return ORBX_OK;
}

void orbx_comm_destroy(orbx_comm* c) {
  if (!c) return;
  Rccl* R = rccl();
  (void)hipSetDevice(c->device >= 0 ? c->device : -1 - c->device);
  if (c->last) (void)hipEventDestroy(c->last);
  if (c->device >= 0 && c->comm && R->CommDestroy) (void)R->CommDestroy(c->comm);
  delete c;
}

int orbx_comm_size(const orbx_comm* c, int* n_ranks, int* rank) {
  if (!c) return fail(ORBX_E_BADARG, "null communicator");
  if (n_ranks) *n_ranks = c->ranks;
  if (rank) *rank = c->rank;
  return ORBX_OK;
}

int orbx_allgather_descriptors(orbx_extractor* ex, orbx_comm* c, int n_images, uint8_t* d_all_desc, int32_t* d_all_counts) {
  if (!ex || !c || !d_all_desc || !d_all_counts) return fail(ORBX_E_BADARG, "null argument");
  if (ex->lastN <= 0) return fail(ORBX_E_BADARG, "no batch has been extracted on this handle");
  if (n_images <= 0 || n_images > ex->lastN) return fail(ORBX_E_BADARG, "n_images exceeds the last batch");
  Rccl* R = rccl();
  if (!R->err.empty()) return fail(ORBX_E_UNSUPPORTED, R->err);
  HIPC(hipSetDevice(ex->device));
  const size_t block = (size_t)ex->gmax.outCap * 32;
  std::lock_guard<std::mutex> lk(c->mu);  // (issue order = lock order; all ranks must use the same)
  // behind the previous collective of this communicator, whichever handle's stream it ran on (ordering rule above)
  if (c->lastValid) HIPC(hipStreamWaitEvent(ex->stream, c->last, 0));
  // both members of every block in ONE RCCL launch, on the handle's stream: ordered behind k_describe, nothing waits on the host
  RCCLC(R->GroupStart());
  ncclResult_t e1 = R->AllGather(ex->d_desc.p, d_all_desc, (size_t)n_images * block, ncclUint8, c->comm, ex->stream);
  ncclResult_t e2 = R->AllGather(ex->d_nOut.p, d_all_counts, (size_t)n_images, ncclInt32, c->comm, ex->stream);
  RCCLC(R->GroupEnd());
  RCCLC(e1);
  RCCLC(e2);
  HIPC(hipEventRecord(c->last, ex->stream));
  c->lastValid = true;
  c->seq++;
  return ORBX_OK;
}

int orbx_comm_wait(orbx_comm* c, int timeout_ms, unsigned long long* n_collectives) {
  if (!c || timeout_ms < 0) return fail(ORBX_E_BADARG, "bad arguments");
  hipEvent_t ev;
  unsigned long long seq;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    ev = c->last;
    seq = c->seq;
    if (n_collectives) *n_collectives = seq;
    if (!c->lastValid) return ORBX_OK;
  }
  HIPC(hipSetDevice(c->device >= 0 ? c->device : -1 - c->device));
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return ORBX_OK;
    if (e != hipErrorNotReady) return fail(ORBX_E_HIP, hipGetErrorString(e));
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (ms >= timeout_ms)
      return fail(ORBX_E_TIMEOUT, "collective #" + std::to_string(seq) + " of rank " + std::to_string(c->rank) + " / " +
                                      std::to_string(c->ranks) + " has not completed after " + std::to_string(timeout_ms) +
                                      " ms: a rank is missing, or the ranks issued their all-gathers in different orders / with "
                                      "different sizes (every rank must make the same orbx_allgather_descriptors calls in the same order)");
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

}  // extern "C"
