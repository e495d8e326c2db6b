// How many 64-thread workgroups with N bytes of dynamic LDS does a gfx950 CU hold?  (LDS allocation granularity probe.)
// hipcc --offload-arch=gfx950 -O2 -o lds_granule tools/ubench/lds_granule.hip && ./lds_granule
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_hold(int* cur, int* peak, long long ticks) {
  extern __shared__ int sm[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));   // gfx9: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID separate
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int cu = (int)(((xcc & 0xf) << 8) | ((hw >> 8) & 0xff));      // (xcc, se, sh, cu) -> one slot of 4096
  if (threadIdx.x == 0) {
    sm[0] = cu;
    const int now = atomicAdd(&cur[cu], 1) + 1;
    atomicMax(&peak[cu], now);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    atomicSub(&cur[cu], 1);
  }
}
int main() {
  int *cur, *peak;
  hipMalloc(&cur, 4096 * 4);
  hipMalloc(&peak, 4096 * 4);
  const int sizes[] = {5120, 5121, 5376, 5632, 6400, 6401, 7680, 22592, 23040, 23041, 19456, 2560, 1280, 1281};
  for (int lds : sizes) {
    hipMemset(cur, 0, 4096 * 4);
    hipMemset(peak, 0, 4096 * 4);
    hipFuncSetAttribute((const void*)k_hold, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_hold, dim3(256 * 40), dim3(64), lds, 0, cur, peak, 2000LL);   // 20 us per workgroup
    hipDeviceSynchronize();
    std::vector<int> h(4096);
    hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
    int mx = 0, used = 0;
    for (int v : h) { if (v > mx) mx = v; if (v) used++; }
    printf("dynamic LDS %6d B per 64-thread workgroup: peak %2d workgroups on a CU (%d CUs seen)  -> %d B per slot\n", lds, mx, used, mx ? 163840 / mx : 0);
  }
  return 0;
}
