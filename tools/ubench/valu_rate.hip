// Micro-benchmark: issue rate of the integer VALU ops k_detect is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; prints Gops/s (lane-ops) per op kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * (i + 1);
  uint32_t b = seed * 3 + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (KIND == 0) a[i] = a[i] + b;                                            // v_add_u32
        if (KIND == 1) a[i] = __builtin_amdgcn_alignbit(a[i], b, 31);              // v_alignbit_b32
        if (KIND == 2) a[i] = a[i] - ((b >> 8) & 0xFF);                            // v_sub_u32_sdwa (byte select)
        if (KIND == 3) a[i] = (uint32_t)min((int)a[i], (int)b);                    // v_min_i32
        if (KIND == 4) a[i] = (uint32_t)min(min((int)a[i], (int)b), (int)(b ^ i)); // v_min3_i32
        if (KIND == 5) a[i] = a[i] & (a[i] >> 1);                                  // shift + and (2 ops)
        if (KIND == 6) a[i] = __builtin_amdgcn_udot4(a[i], b, a[i], false);        // v_dot4_u32_u8
        if (KIND == 7) a[i] = a[i] * 18u + b;                                      // v_mad_u32_u24 / mul_lo
        b += 0x9E3779B9u;
      }
    }
  }
  uint32_t s = b;
#pragma unroll
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, double ops_per_inner) {
  uint32_t* d;
  const int blocks = 256 * 16, iters = 2000;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // each inner statement = ops_per_inner target ops + 1 v_add for b
  const double inner = (double)blocks * 256 * iters * 64;
  printf("%-28s %8.1f G inner-stmts/s  (%.2f ms)  => if b-add costs 1 op: %.1f Gops/s total\n", name, inner / ms / 1e6, ms,
         inner * (ops_per_inner + 1) / ms / 1e6);
  hipFree(d);
}

int main() {
  run<0>("v_add_u32", 1);
  run<1>("v_alignbit_b32", 1);
  run<2>("v_sub_u32_sdwa", 1);
  run<3>("v_min_i32", 1);
  run<4>("v_min3_i32 (+xor)", 2);
  run<5>("lshr+and", 2);
  run<6>("v_dot4_u32_u8", 1);
  run<7>("v_mad (a*18+b)", 1);
  return 0;
}
