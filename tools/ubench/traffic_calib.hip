// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access shapes the ORB
// kernels use (MI355X_MICROARCH.md "HBM": only the 16 B/lane streaming read is calibrated there: x2).
// Every kernel touches each byte of its range exactly once; main() prints {"kernel": bytes} as JSON on stdout.
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and again with WRITE_SIZE; tools/pmc_calib.py joins both.
// Build: hipcc --offload-arch=gfx950 -O2 traffic_calib.hip -o traffic_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// ---- reads: one element per lane per trip, grid-stride, result folded into one dword per thread
__global__ __launch_bounds__(256) void k_rd_1B(const uint8_t* s, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= s[i];
  if (acc == 0x77u) sink[threadIdx.x] = acc;  // a byte XOR stays below 256: keep the test satisfiable or the loop is deleted
}
__global__ __launch_bounds__(256) void k_rd_4B(const uint32_t* s, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= s[i];
  if (acc == 0x12345677u) sink[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_rd_8B(const uint2* s, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint2 v = s[i];
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345677u) sink[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_rd_16B(const uint4* s, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = s[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345677u) sink[threadIdx.x] = acc;
}
// k_detect's loader shape: one wave per 48 x 44-byte-rows "cell" of an image with a 1280-byte pitch; lane = (row group,
// 8-byte column): 6 lanes cover a 48-byte row, 10 rows per trip.  Cells tile the image exactly once (no halo), so the
// known byte count is the image size.
__global__ __launch_bounds__(64) void k_rd_cell8B(const uint8_t* img, int pitch, int cellsX, int cellsY, uint32_t* sink) {
  const int cell = blockIdx.x, cx = cell % cellsX, cy = (cell / cellsX) % cellsY, im = cell / (cellsX * cellsY);
  const uint8_t* base = img + (size_t)im * pitch * (cellsY * 44) + (size_t)cy * 44 * pitch + cx * 48;
  const int col = threadIdx.x % 6, rg = threadIdx.x / 6;  // 60 lanes active
  uint32_t acc = 0;
  if (threadIdx.x < 60) {
    uint2 v[5];
#pragma unroll
    for (int t = 0; t < 5; t++) {
      const int row = rg + 10 * t;
      v[t] = row < 44 ? *reinterpret_cast<const uint2*>(base + (size_t)row * pitch + col * 8) : make_uint2(0, 0);
    }
#pragma unroll
    for (int t = 0; t < 5; t++) acc ^= v[t].x ^ v[t].y;
  }
  if (acc == 0x12345677u) sink[threadIdx.x] = acc;
}
// ---- writes
__global__ __launch_bounds__(256) void k_wr_1B(uint8_t* d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (uint8_t)i;
}
__global__ __launch_bounds__(256) void k_wr_4B(uint32_t* d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_wr_8B(uint2* d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = make_uint2((uint32_t)i, 1);
}
__global__ __launch_bounds__(256) void k_wr_16B(uint4* d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
// re-read of a buffer that fits the 256 MiB Infinity Cache (are MALL hits counted?): 64 MiB read right after it was read
__global__ __launch_bounds__(256) void k_rd_16B_again64M(const uint4* s, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = s[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345677u) sink[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_rd_16B_again8M(const uint4* s, size_t n, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = s[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345677u) sink[threadIdx.x] = acc;
}

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

int main() {
  const size_t N = (size_t)1 << 30;  // 1 GiB: four times the Infinity Cache
  uint8_t* buf;
  uint32_t* sink;
  CK(hipMalloc(&buf, N));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(buf, 1, N));
  CK(hipDeviceSynchronize());
  const int G = 256 * 16;
  hipLaunchKernelGGL(k_rd_1B, dim3(G), dim3(256), 0, 0, buf, N / 4, sink);  // 256 MiB of byte loads (slow shape)
  hipLaunchKernelGGL(k_rd_4B, dim3(G), dim3(256), 0, 0, (const uint32_t*)buf, N / 4, sink);
  hipLaunchKernelGGL(k_rd_8B, dim3(G), dim3(256), 0, 0, (const uint2*)buf, N / 8, sink);
  hipLaunchKernelGGL(k_rd_16B, dim3(G), dim3(256), 0, 0, (const uint4*)buf, N / 16, sink);
  // cell-shaped reads: 26 x 16 cells of 48 x 44 per 1280-pitch "image" (1248 of 1280 columns), 800 images = 0.9 GiB span
  const int cellsX = 26, cellsY = 16, imgs = 800, pitch = 1280;
  hipLaunchKernelGGL(k_rd_cell8B, dim3(cellsX * cellsY * imgs), dim3(64), 0, 0, buf, pitch, cellsX, cellsY, sink);
  hipLaunchKernelGGL(k_wr_1B, dim3(G), dim3(256), 0, 0, buf, N / 4);
  hipLaunchKernelGGL(k_wr_4B, dim3(G), dim3(256), 0, 0, (uint32_t*)buf, N / 4);
  hipLaunchKernelGGL(k_wr_8B, dim3(G), dim3(256), 0, 0, (uint2*)buf, N / 8);
  hipLaunchKernelGGL(k_wr_16B, dim3(G), dim3(256), 0, 0, (uint4*)buf, N / 16);
  // cache-resident re-reads
  const size_t M64 = (size_t)64 << 20, M8 = (size_t)8 << 20;
  hipLaunchKernelGGL(k_rd_16B, dim3(G), dim3(256), 0, 0, (const uint4*)buf, M64 / 16, sink);
  hipLaunchKernelGGL(k_rd_16B_again64M, dim3(G), dim3(256), 0, 0, (const uint4*)buf, M64 / 16, sink);
  hipLaunchKernelGGL(k_rd_16B, dim3(G), dim3(256), 0, 0, (const uint4*)buf, M8 / 16, sink);
  hipLaunchKernelGGL(k_rd_16B_again8M, dim3(G), dim3(256), 0, 0, (const uint4*)buf, M8 / 16, sink);
  CK(hipDeviceSynchronize());
  // launch order = dispatch order in the rocpd database; tools/pmc_calib.py joins by position
  printf("[[\"k_rd_1B\", %zu], [\"k_rd_4B\", %zu], [\"k_rd_8B\", %zu], [\"k_rd_16B\", %zu], [\"k_rd_cell8B\", %zu], "
         "[\"k_wr_1B\", %zu], [\"k_wr_4B\", %zu], [\"k_wr_8B\", %zu], [\"k_wr_16B\", %zu], [\"k_rd_16B(64M, cold)\", %zu], "
         "[\"k_rd_16B_again64M\", %zu], [\"k_rd_16B(8M, cold)\", %zu], [\"k_rd_16B_again8M\", %zu]]\n",
         N / 4, N, N, N, (size_t)cellsX * 48 * cellsY * 44 * imgs, N / 4, N, N, N, M64, M64, M8, M8);
  return 0;
}
