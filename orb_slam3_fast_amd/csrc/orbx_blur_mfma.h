// orbx_blur_mfma.h — the 7x7 Gaussian of k_describe as two banded integer GEMMs on the matrix pipe (round 6).
//
// cv::GaussianBlur(7x7, sigma 2) of the reference's operator() (src/ORBextractor.cc:1074-1076) is separable and exact in
// integers: H[r][c] = sum_k w[k] In[r][c + k] (fits 16 bits), out = (sum_k w[k] H[r + k][c] + 32768) >> 16.  Both
// passes over a keypoint's 43x43 window are products with a banded Toeplitz matrix of the taps, so they run on
// v_mfma_i32_16x16x64_i8 (i32 accumulation: exact) instead of ~225 VALU + ~50 LDS instructions per keypoint:
//
//   horizontal   Hs = (In - 128) x Bh        In: window rows as the A operand (16 consecutive bytes of a row per lane: one
//                                            ds_read_b128 + one v_xor per dword), Bh[k][n] = w[k - n] constant
//   split        X = Hs + bias = 256 hi + lo_u,  lo_s = lo_u - 128 (both signed bytes; bias 128 only for the 257-sum taps)
//   vertical     Out^T = Hhi^T x Av^T * 256 + Hlo^T x Av^T,   Av[m][row] = w[row - m] constant
//
// The D layout of the horizontal product (lane = column, 4 consecutive rows per lane) IS the A layout the vertical product
// wants (lane = column, K = rows) once K is enumerated as k(g, j) = 16 (j / 4) + 4 g + j % 4 -- the order of K is free as long
// as both operands use the same one, and the constant operand is built for it here.  No cross-lane movement between the passes.
//
// Operand layouts assumed (gfx950, v_mfma_i32_16x16x64_i8; checked against a CPU model in tests/test_blur_mfma_plan.py and on the
// device by the descriptor parity tests): A lane (i = lane & 15, g = lane >> 4) byte j <-> A[i][16 g + j]; B lane (n, g) byte j
// <-> B[16 g + j][n]; C / D lane (n, g) register r <-> D[4 g + r][n].
#pragma once
#include <stdint.h>

namespace orbx {

constexpr int BM_P = 48;        // byte pitch of the raw window and of the blurred patch (rows are 16-byte aligned for ds_read_b128)
constexpr int BM_ROWS = 43;     // window rows / columns that carry data
// bytes of LDS one wave needs: window rows 0..47 are addressed (rows 43..47 only feed outputs nobody reads), lanes of K group 3
// read 16 bytes past their row, and the blurred patch (37 x 48, rows up to 47 written) overlays the dead window
constexpr int BM_WAVE_BYTES = 48 * BM_P + 32;

struct BlurMfmaTab {
  uint32_t bh[3][64][4];   // [column tile ct][lane][dword]: B operand of the horizontal product
  uint32_t av[3][64][4];   // [row tile mt][lane][dword]:    constant operand of the vertical product (dword 3 = 0)
};

template <bool T440>
constexpr BlurMfmaTab make_blur_mfma_tab() {
  BlurMfmaTab t{};
  const uint32_t tap[7] = {18u, 34u, T440 ? 49u : 48u, T440 ? 55u : 56u, T440 ? 49u : 48u, 34u, 18u};
  for (int tile = 0; tile < 3; tile++)
    for (int lane = 0; lane < 64; lane++) {
      const int n = lane & 15, g = lane >> 4;
      for (int d = 0; d < 4; d++) {
        uint32_t bh = 0, av = 0;
        for (int jj = 0; jj < 4; jj++) {
          const int j = 4 * d + jj;
          {  // horizontal: K = window byte k, N = tap-left column n' = 16 tile + n; H'[r][n'] = sum_e w[e] In[r][n' + e]
            const int k = 16 * g + j, e = k - (16 * tile + n);
            if (k < 48 && e >= 0 && e <= 6) bh |= tap[e] << (8 * jj);
          }
          if (d < 3) {  // vertical: K = H row(g, j), output row m = 16 tile + n; out[m][c] = sum_e w[e] H'[m + e][c]
            const int row = 16 * d + 4 * g + jj, e = row - (16 * tile + n);
            if (row < BM_ROWS && e >= 0 && e <= 6) av |= tap[e] << (8 * jj);
          }
        }
        t.bh[tile][lane][d] = bh;
        t.av[tile][lane][d] = av;
      }
    }
  return t;
}

// constants of the split / recombination: X = Hs + bias; out = (256 HI + LO + kc) >> 16
template <bool T440>
struct BlurMfmaConst {
  static constexpr int sumw = T440 ? 257 : 256;
  static constexpr int bias = T440 ? 128 : 0;   // keeps (Hs + bias) >> 8 inside a signed byte for the 257-sum taps
  static constexpr int kc = (128 - bias) * sumw + 128 * sumw * sumw + 32768;
};

}  // namespace orbx
