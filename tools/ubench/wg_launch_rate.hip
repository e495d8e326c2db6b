// Workgroup launch rate of gfx950: how long do N empty (or nearly empty) workgroups take?
// hipcc --offload-arch=gfx950 -O2 -o wg_launch_rate tools/ubench/wg_launch_rate.hip && ./wg_launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int* out, int work) {
  extern __shared__ int sm[];
  if (work) {
    int a = threadIdx.x;
    for (int i = 0; i < work; i++) a = a * 3 + 1;
    if (a == 0x7fffffff) out[0] = a;
  }
}
int main() {
  int* out;
  (void)hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int grids[][2] = {{127168, 64}, {63584, 128}, {31792, 256}, {127168 / 8, 512}, {114432 / 4, 256}, {29760 / 4, 256}};
  for (auto& g : grids)
    for (int lds : {0, 5376, 14784})
      for (int work : {0, 2000}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
          (void)hipEventRecord(e0);
          hipLaunchKernelGGL(k_empty, dim3(g[0]), dim3(g[1]), lds, 0, out, work);
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
        }
        printf("%7d workgroups x %3d threads, %5d B LDS, %4d dependent mads per thread: %8.1f us  (%.2f ns per workgroup)\n", g[0], g[1], lds,
               work, best * 1e3, best * 1e6 / g[0]);
      }
  return 0;
}
