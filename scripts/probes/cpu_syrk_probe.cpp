// CPU probe behind cpu_baseline (round 5): why does the oracle built -march=native run several times slower on the GPU box's
// EPYC 9575F than built -march=x86-64-v3?  Prints, for the flags it was compiled with: the FMA rate of register-only loops
// with 256-bit and (when the ISA has them) 512-bit vectors, the rate of the oracle's J^T J micro-kernel and blocked Cholesky
// on the cfg2 shape (M = 192, n = 128).  scripts/probes/cpu_syrk_probe.sh builds it with each candidate flag set.
#include <chrono>
#include <cstdio>
#include <random>

#include "../../oracle/mmx_oracle.hpp"

using namespace mmx_oracle;
using Clock = std::chrono::steady_clock;

typedef float V32 __attribute__((vector_size(32)));
typedef float V64 __attribute__((vector_size(64)));
template <class V>
static double fmaRate() {
  constexpr int kBytes = int(sizeof(V));
  V acc[12], a, b;
  for (int l = 0; l < kBytes / 4; ++l) {
    a[l] = 1.0000001f, b[l] = 1e-9f;
    for (auto& v : acc) {
      v[l] = float(l);
    }
  }
  const long iters = 20000000;
  const auto t0 = Clock::now();
  for (long i = 0; i < iters; ++i) {
    for (auto& v : acc) {
      v = v * a + b;
    }
    asm volatile("" : "+x"(acc[0]), "+x"(acc[5]), "+x"(acc[11]));
  }
  const double s = std::chrono::duration<double>(Clock::now() - t0).count();
  float sink = 0.f;
  for (auto& v : acc) {
    sink += v[0];
  }
  if (sink == 12345.f) {
    std::printf("!");
  }
  return double(iters) * 12 * (kBytes / 4) * 2 / s / 1e9;
}

int main() {
  std::printf("vector bytes of the oracle's kernels in this build: %d\n", kOrcVecBytes);
  std::printf("register-only FMA loop, 256-bit: %.1f GFLOP/s\n", fmaRate<V32>());
#if defined(__AVX512F__)
  std::printf("register-only FMA loop, 512-bit: %.1f GFLOP/s\n", fmaRate<V64>());
#endif
  const int M = 192, n = 128;
  std::vector<float> J(size_t(M) * n), r(M), H(size_t(n) * n), g(n);
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> u(-1, 1);
  for (auto& v : J) {
    v = u(rng);
  }
  for (auto& v : r) {
    v = u(rng);
  }
  const int reps = 4000;
  auto t0 = Clock::now();
  for (int i = 0; i < reps; ++i) {
    accumulateNormalEquations<float>(J.data(), r.data(), M, M, n, H.data(), g.data());
  }
  double s = std::chrono::duration<double>(Clock::now() - t0).count() / reps;
  std::printf("J^T J (M n (n + 1) flops): %.4f ms  %.1f GFLOP/s\n", 1e3 * s, double(M) * n * (n + 1) / s / 1e9);
  std::vector<float> A(size_t(n) * n);
  t0 = Clock::now();
  for (int i = 0; i < reps; ++i) {
    for (int j = 0; j < n; ++j) {
      for (int k = 0; k < n; ++k) {
        A[j * n + k] = H[j * n + k] / reps + (j == k ? 1000.f : 0.f);
      }
    }
    choleskyLower<float>(A.data(), n);
  }
  s = std::chrono::duration<double>(Clock::now() - t0).count() / reps;
  std::printf("Cholesky (n^3 / 3 flops, incl. a copy of the matrix): %.4f ms  %.1f GFLOP/s\n", 1e3 * s, double(n) * n * n / 3 / s / 1e9);
  return 0;
}
