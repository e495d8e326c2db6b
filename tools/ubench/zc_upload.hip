// How should a single frame's two 1280 x 720 images reach HBM?  Wall time from "the frame is in host memory" to "both images are
// in device memory and the stream is idle", for
//   (a) two hipMemcpy2DAsync from pageable memory (what orbx_extract_stereo does with a caller's cv::Mat),
//   (b) two hipMemcpyAsync from page-locked memory,
//   (c) CPU memcpy into a page-locked staging block + ONE kernel that reads the block over PCIe and writes device memory,
//   (d) the kernel alone on already page-locked frames,
//   (e) as (c) but image by image: memcpy L, launch L, memcpy R, launch R.
// The source frames are rewritten (and so cache-warm, like a frame that has just been decoded / debayered) before every trial;
// `--cold` flushes them through a 256 MB sweep first (a frame DMA'd into memory by a capture card).
//   hipcc --offload-arch=gfx950 -O2 -o zc_upload zc_upload.hip && ./zc_upload [--cold]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_pull(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const bool cold = argc > 1 && !strcmp(argv[1], "--cold");
  const int W = 1280, H = 720, N = W * H;
  uint8_t* page[2];
  for (int i = 0; i < 2; i++) page[i] = (uint8_t*)aligned_alloc(4096, N);
  uint8_t *pin[2], *stage, *dev;
  for (int i = 0; i < 2; i++) CK(hipHostMalloc((void**)&pin[i], N, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&stage, 2 * N, hipHostMallocDefault));
  uint8_t* stageDev;
  CK(hipHostGetDevicePointer((void**)&stageDev, stage, 0));
  uint8_t* pinDev[2];
  for (int i = 0; i < 2; i++) CK(hipHostGetDevicePointer((void**)&pinDev[i], pin[i], 0));
  CK(hipMalloc((void**)&dev, 2 * N));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<uint8_t> sweep(cold ? (256u << 20) : 1);
  const int trials = 60;
  for (int blocks : {256, 512, 1024, 2048}) {
    std::vector<double> t[5];
    for (int it = 0; it < trials; it++) {
      for (int mode = 0; mode < 5; mode++) {
        for (int i = 0; i < 2; i++) { memset(page[i], it + i + mode, N); memset(pin[i], it + i + mode, N); }
        if (cold) { for (size_t j = 0; j < sweep.size(); j += 64) sweep[j]++; }
        CK(hipStreamSynchronize(s));
        const double t0 = now_us();
        if (mode == 0) {
          for (int i = 0; i < 2; i++) CK(hipMemcpy2DAsync(dev + (size_t)i * N, W, page[i], W, W, H, hipMemcpyHostToDevice, s));
        } else if (mode == 1) {
          for (int i = 0; i < 2; i++) CK(hipMemcpyAsync(dev + (size_t)i * N, pin[i], N, hipMemcpyHostToDevice, s));
        } else if (mode == 2) {
          for (int i = 0; i < 2; i++) memcpy(stage + (size_t)i * N, page[i], N);
          hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, s, (const uint4*)stageDev, (uint4*)dev, 2 * N / 16);
        } else if (mode == 3) {
          for (int i = 0; i < 2; i++)
            hipLaunchKernelGGL(k_pull, dim3(blocks / 2), dim3(256), 0, s, (const uint4*)pinDev[i], (uint4*)(dev + (size_t)i * N), N / 16);
        } else {
          for (int i = 0; i < 2; i++) {
            memcpy(stage + (size_t)i * N, page[i], N);
            hipLaunchKernelGGL(k_pull, dim3(blocks / 2), dim3(256), 0, s, (const uint4*)(stageDev + (size_t)i * N), (uint4*)(dev + (size_t)i * N), N / 16);
          }
        }
        CK(hipStreamSynchronize(s));
        t[mode].push_back(now_us() - t0);
      }
    }
    const char* names[5] = {"2 x hipMemcpy2DAsync pageable", "2 x hipMemcpyAsync page-locked", "CPU memcpy to staging + 1 pull kernel",
                            "2 pull kernels on page-locked frames", "memcpy L, pull L, memcpy R, pull R"};
    printf("pull grid %d blocks x 256%s\n", blocks, cold ? " (cold source)" : "");
    for (int m = 0; m < 5; m++) {
      std::sort(t[m].begin(), t[m].end());
      printf("  %-42s p10 %6.1f  p50 %6.1f  p90 %6.1f us\n", names[m], t[m][trials / 10], t[m][trials / 2], t[m][trials * 9 / 10]);
    }
  }
  // verify the last pull
  std::vector<uint8_t> back(2 * N);
  CK(hipMemcpy(back.data(), dev, 2 * N, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 2; i++) bad += memcmp(back.data() + (size_t)i * N, page[i], N) != 0;
  printf("last pull %s\n", bad ? "MISMATCH" : "verified");
  return 0;
}
