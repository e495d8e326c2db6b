// occupancy of a 64-thread block vs dynamic LDS size: reveals the LDS allocation granule of the device
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int* o) { extern __shared__ int s[]; s[threadIdx.x] = 1; __syncthreads(); o[threadIdx.x] = s[63 - threadIdx.x]; }
int main() {
  int prev = -1;
  for (int lds = 2048; lds <= 9216; lds += 64) {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, lds);
    if (nb != prev) printf("lds %d -> %d blocks/CU\n", lds, nb);
    prev = nb;
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerBlock %zu regsPerMultiprocessor %d\n", p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerMultiprocessor);
}
