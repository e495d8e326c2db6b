// Prints the constant MFMA operands of k_describe's blur (csrc/orbx_blur_mfma.h) for tests/test_blur_mfma_plan.py.
#include <cstdio>
#include "../../orb_slam3_fast_amd/csrc/orbx_blur_mfma.h"
template <bool T>
static void dump() {
  constexpr orbx::BlurMfmaTab t = orbx::make_blur_mfma_tab<T>();
  std::printf("%d %d %d\n", orbx::BlurMfmaConst<T>::sumw, orbx::BlurMfmaConst<T>::bias, orbx::BlurMfmaConst<T>::kc);
  for (int a = 0; a < 3; a++)
    for (int l = 0; l < 64; l++) std::printf("%u %u %u %u\n", t.bh[a][l][0], t.bh[a][l][1], t.bh[a][l][2], t.bh[a][l][3]);
  for (int a = 0; a < 3; a++)
    for (int l = 0; l < 64; l++) std::printf("%u %u %u %u\n", t.av[a][l][0], t.av[a][l][1], t.av[a][l][2], t.av[a][l][3]);
}
int main() {
  std::printf("%d %d %d\n", orbx::BM_P, orbx::BM_ROWS, orbx::BM_WAVE_BYTES);
  dump<false>();
  dump<true>();
  return 0;
}
