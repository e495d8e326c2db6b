// What do ds_read_u8_d16 / ds_read_u8_d16_hi leave in the OTHER half of the destination register on gfx950?
// (With SRAM ECC the compiler assumes d16 loads do not preserve it and packs byte pairs with v_perm instead: k_detect's contrast
// pass spends 17 v_perm per pass on that.)   hipcc --offload-arch=gfx950 -O2 -o d16_probe d16_probe.hip && ./d16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(uint32_t* out) {
  __shared__ uint8_t s[256];
  for (int i = threadIdx.x; i < 256; i += 64) s[i] = (uint8_t)(i ^ 0x5A);
  __syncthreads();
  uint32_t a = 0xAAAAAAAAu, b = 0xBBBBBBBBu, c = 0xCCCCCCCCu;
  const uint32_t addr = (uint32_t)(uintptr_t)s + threadIdx.x;   // LDS byte address
  asm volatile("ds_read_u8_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr));
  asm volatile("ds_read_u8_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(b) : "v"(addr));
  asm volatile("ds_read_u8_d16 %0, %1\n\tds_read_u8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(c) : "v"(addr), "v"(addr + 64));
  out[threadIdx.x * 3] = a;
  out[threadIdx.x * 3 + 1] = b;
  out[threadIdx.x * 3 + 2] = c;
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 64 * 3 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint32_t h[192];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t : {0, 1, 7}) {
    printf("lane %d: byte %02x / byte+64 %02x | d16 into 0xAAAAAAAA -> %08x | d16_hi into 0xBBBBBBBB -> %08x | d16 then d16_hi into 0xCCCCCCCC -> %08x\n",
           t, (t ^ 0x5A) & 0xFF, ((t + 64) ^ 0x5A) & 0xFF, h[3 * t], h[3 * t + 1], h[3 * t + 2]);
  }
  return 0;
}
