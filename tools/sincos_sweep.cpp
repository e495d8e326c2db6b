// tools/sincos_sweep.cpp — exhaustive check of the oracle's glibc sinf/cosf models against this host's libm over EVERY
// float in [0, 2 pi] (1.09e9 arguments: all values `angle * factorPI` can take in computeOrbDescriptor,
// src/ORBextractor.cc:106-107), and the list of arguments on which the FMA and SSE2 variants differ.
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -pthread -o /tmp/sincos_sweep tools/sincos_sweep.cpp oracle/orb_oracle.cpp
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include "../oracle/orb_oracle.h"
int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 8;
  const uint32_t hi = 0x40C91000u;  // just above 2 pi (0x40C90FDB)
  std::atomic<long long> badF{0}, badS{0}, diff{0};
  std::mutex mu;
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&, t]() {
      long long bF = 0, bS = 0, d = 0;
      for (uint64_t u = t; u <= hi; u += nt) {
        float y;
        const uint32_t uu = (uint32_t)u;
        std::memcpy(&y, &uu, 4);
        const float hs = sinf(y), hc = cosf(y);
        const float fs = orbo::glibc_sinf_model(y, true), fc = orbo::glibc_cosf_model(y, true);
        const float ss = orbo::glibc_sinf_model(y, false), sc = orbo::glibc_cosf_model(y, false);
        bF += (hs != fs) + (hc != fc);
        bS += (hs != ss) + (hc != sc);
        if (fs != ss || fc != sc) {
          d++;
          std::lock_guard<std::mutex> lk(mu);
          printf("differ: 0x%08x  sin fma %a sse2 %a  cos fma %a sse2 %a  host %a %a\n", uu, fs, ss, fc, sc, hs, hc);
        }
      }
      badF += bF; badS += bS; diff += d;
    });
  for (auto& x : th) x.join();
  printf("arguments 0..0x%08x: host-vs-FMA-model mismatches %lld, host-vs-SSE2-model mismatches %lld, FMA-vs-SSE2 differing arguments %lld\n",
         hi, (long long)badF, (long long)badS, (long long)diff);
  return 0;
}
