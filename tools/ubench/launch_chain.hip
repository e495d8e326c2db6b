// Fixed cost of a dependent kernel launch inside one HIP stream: wall time of a chain of N launches (N = 1, 2, 4, 8, 16) of
//   (a) an empty kernel, 1 workgroup of 64;   (b) an empty kernel, 4800 workgroups of 64;   (c) a kernel that writes 2 MB (dirty L2)
// from "first launch call" to "stream idle", and the slope per added launch.  A single 1280x720 stereo frame is 7 dependent launches.
//   hipcc --offload-arch=gfx950 -O2 -o launch_chain launch_chain.hip && ./launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_nop(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_write(uint4* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = make_uint4(i, 1, 2, 3); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint4* buf;
  CK(hipMalloc((void**)&buf, 2 << 20));
  const int n16 = (2 << 20) / 16;
  for (int mode = 0; mode < 3; mode++) {
    double prev = 0;
    for (int N : {1, 2, 4, 8, 16}) {
      std::vector<double> t;
      for (int it = 0; it < 200; it++) {
        CK(hipStreamSynchronize(s));
        const double t0 = now_us();
        for (int k = 0; k < N; k++) {
          if (mode == 0) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, (int*)nullptr);
          else if (mode == 1) hipLaunchKernelGGL(k_nop, dim3(4800), dim3(64), 0, s, (int*)nullptr);
          else hipLaunchKernelGGL(k_write, dim3(n16 / 256), dim3(256), 0, s, buf, n16);
        }
        CK(hipStreamSynchronize(s));
        t.push_back(now_us() - t0);
      }
      std::sort(t.begin(), t.end());
      const double med = t[t.size() / 2];
      printf("%s  N = %2d: %7.1f us", mode == 0 ? "empty 1 x 64     " : mode == 1 ? "empty 4800 x 64  " : "write 2 MB       ", N, med);
      if (N > 1) printf("   (+%.2f us per added launch)", (med - prev) / (N / 2));
      printf("\n");
      prev = med;
    }
  }
  return 0;
}
