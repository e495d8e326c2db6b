// orbx_api_probe.hip — measurement aid: the shader clock the chip actually runs at WHILE the pipeline's kernels execute.
// One wave on a private non-blocking stream spins for a given time and brackets the spin with s_memtime (shader-clock
// cycles) and s_memrealtime (the constant 100 MHz reference counter); cycles / ns is the clock of the whole shader
// array (one clock domain), i.e. of whatever kernels ran beside the probe.  bench.py starts it right after the timed
// region, enqueues a few more steps, and divides by this clock instead of assuming the 2.4 GHz peak (DVFS: the dense
// integer kernels of this path hold 2.1 - 2.3 GHz, profiles/r3_valu_issue.txt).
#include <algorithm>

#include "orbx_host.h"

using namespace orbx_host;

namespace {
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* out, unsigned long long ticks_100mhz) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  while (r1 - r0 < ticks_100mhz) {
    __builtin_amdgcn_s_sleep(8);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
}
// Device-to-device copy with 16-byte accesses, four loads in flight per thread: the "measured copy peak" SURVEY 8d asks the
// roofline fractions to be quoted against as well (a uint8 torch copy_ reached 4.7 - 5.3 TB/s on this chip, the hardware
// guide's float4 copy kernel 6.29 TB/s: MI355X_MICROARCH.md:35).
typedef unsigned int probe_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_probe(const probe_u4* __restrict__ src, probe_u4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    probe_u4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i + 256 * k < n16) v[k] = __builtin_nontemporal_load(src + i + 256 * k);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (i + 256 * k < n16) __builtin_nontemporal_store(v[k], dst + i + 256 * k);
  }
}
struct Probe {
  int device = 0;
  hipStream_t stream = nullptr;
  unsigned long long* d = nullptr;
};
}  // namespace

extern "C" {

int orbx_clock_probe_start(int device, int spin_us, void** probe) {
  if (!probe || spin_us <= 0 || spin_us > 1000000) return fail(ORBX_E_BADARG, "bad probe arguments");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return fail(ORBX_E_NODEVICE, "no HIP device");
  if (device < 0 || device >= nd) return fail(ORBX_E_BADARG, "device index out of range");
  HIPC(hipSetDevice(device));
  std::unique_ptr<Probe> p(new (std::nothrow) Probe);
  if (!p) return fail(ORBX_E_HIP, "out of memory");
  p->device = device;
  HIPC(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  HIPC(hipMalloc((void**)&p->d, 2 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, p->stream, p->d, (unsigned long long)spin_us * 100ull);
  HIPC(hipGetLastError());
  *probe = p.release();
  return ORBX_OK;
}

int orbx_copy_probe(int device, size_t bytes, int iters, double* gbps) {
  if (!gbps || iters <= 0 || iters > 1000 || bytes < (1u << 20) || (bytes & 15)) return fail(ORBX_E_BADARG, "bad copy-probe arguments");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0) return fail(ORBX_E_NODEVICE, "no HIP device");
  if (device < 0 || device >= nd) return fail(ORBX_E_BADARG, "device index out of range");
  HIPC(hipSetDevice(device));
  probe_u4 *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc((void**)&src, bytes);
  if (e == hipSuccess) e = hipMalloc((void**)&dst, bytes);
  if (e == hipSuccess) e = hipMemset(src, 3, bytes);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float ms = 0.f;
  if (e == hipSuccess) {
    const size_t n16 = bytes / 16;
    const int blocks = (int)std::min<size_t>((n16 + 1023) / 1024, 256 * 16);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_copy_probe, dim3(blocks), dim3(256), 0, nullptr, src, dst, n16);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(k_copy_probe, dim3(blocks), dim3(256), 0, nullptr, src, dst, n16);
    (void)hipEventRecord(e1, nullptr);
    e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(src);
  (void)hipFree(dst);
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  *gbps = ms > 0.f ? 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9 : 0.0;  // bytes read + bytes written
  return ORBX_OK;
}

int orbx_clock_probe_finish(void* probe, double* ghz) {
  if (!probe || !ghz) return fail(ORBX_E_BADARG, "null argument");
  std::unique_ptr<Probe> p(static_cast<Probe*>(probe));
  HIPC(hipSetDevice(p->device));
  unsigned long long h[2] = {0, 0};
  hipError_t e = hipStreamSynchronize(p->stream);
  if (e == hipSuccess) e = hipMemcpy(h, p->d, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(p->d);
  (void)hipStreamDestroy(p->stream);
  if (e != hipSuccess) return fail(ORBX_E_HIP, hipGetErrorString(e));
  *ghz = h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0;  // cycles per ns (s_memrealtime ticks are 10 ns)
  return ORBX_OK;
}

}  // extern "C"
