// Scalar-ALU issue rate of a gfx950 CU: is the scalar unit per SIMD or shared by the CU?
// hipcc --offload-arch=gfx950 -O2 -o salu_rate tools/ubench/salu_rate.hip && ./salu_rate
// One workgroup per CU with W waves, each wave runs N x 64 independent s_add_u32 / s_xor_b32 on 8 SGPR chains.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_salu(unsigned* out, int iters) {
  unsigned a0 = blockIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
      asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                   "s_xor_b32 %4, %4, %0\n s_xor_b32 %5, %5, %1\n s_xor_b32 %6, %6, %2\n s_xor_b32 %7, %7, %3\n"
                   : "+s"(a0), "+s"(a1), "+s"(a2), "+s"(a3), "+s"(a4), "+s"(a5), "+s"(a6), "+s"(a7) : : "scc");
  }
  if (threadIdx.x == 0) out[blockIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_valu(unsigned* out, int iters) {
  unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++)
      asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %4, %5\n"
                   "v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %5, %5, %6, %7\n v_perm_b32 %6, %6, %7, %0\n v_perm_b32 %7, %7, %0, %1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main() {
  unsigned* out;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 4000;
  for (int valu = 0; valu < 2; valu++)
    for (int waves : {1, 2, 4, 8, 16}) {
      for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        if (valu) hipLaunchKernelGGL(k_valu, dim3(256), dim3(64 * waves), 0, 0, out, iters);
        else hipLaunchKernelGGL(k_salu, dim3(256), dim3(64 * waves), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)iters * 64 * waves;   // per CU (one workgroup per CU)
      printf("%s  %2d waves per CU: %.3f ms, %.2f instructions per ns per CU = %.2f per cycle at 2.4 GHz\n", valu ? "v_perm_b32" : "s_add/s_xor",
             waves, ms, insts / (ms * 1e6), insts / (ms * 1e6) / 2.4);
    }
  return 0;
}
