// store_k.hip -- experiment: write bandwidth against the number of stores a wave issues before it
// ends.  A 256-thread block writes K * 4 KB of contiguous memory, 16 B per thread and round.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/store_k scripts/store_k.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int K, bool NT, int TPB>
__global__ void __launch_bounds__(TPB) kfill(float* o) {
  float* p = o + (size_t(blockIdx.x) * K * TPB + threadIdx.x) * 4;
  v4 x = {1.f, 2.f, 3.f, float(blockIdx.x)};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (NT) {
      __builtin_nontemporal_store(x, reinterpret_cast<v4*>(p + size_t(k) * TPB * 4));
    } else {
      *reinterpret_cast<v4*>(p + size_t(k) * TPB * 4) = x;
    }
  }
}
// 12 B per lane: a 64-thread block writes K columns of 768 B
template <int K, bool NT>
__global__ void __launch_bounds__(64) kcol3(float* o) {
  float* p = o + size_t(blockIdx.x) * K * 192 + 3 * threadIdx.x;
  float v = float(blockIdx.x);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (NT) {
      __builtin_nontemporal_store(v, p + k * 192);
      __builtin_nontemporal_store(v, p + k * 192 + 1);
      __builtin_nontemporal_store(v, p + k * 192 + 2);
    } else {
      p[k * 192] = v, p[k * 192 + 1] = v, p[k * 192 + 2] = v;
    }
  }
}
// 12 B per lane, 256-thread block: wave w writes K columns (the block 4 K adjacent columns = 3 K KB)
template <int K, bool NT>
__global__ void __launch_bounds__(256) kcol3w(float* o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* p = o + (size_t(blockIdx.x) * 4 + wave) * K * 192 + 3 * lane;
  float v = float(blockIdx.x);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (NT) {
      __builtin_nontemporal_store(v, p + k * 192);
      __builtin_nontemporal_store(v, p + k * 192 + 1);
      __builtin_nontemporal_store(v, p + k * 192 + 2);
    } else {
      p[k * 192] = v, p[k * 192 + 1] = v, p[k * 192 + 2] = v;
    }
  }
}
int main() {
  const size_t n = size_t(4096) * 192 * 128; // floats = 402 MB
  float* buf;
  hipMalloc(&buf, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("%-34s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, n * 4 / ms / 1e6);
  };
#define RUNK(K_, NT_, T_) run("kfill K=" #K_ " nt=" #NT_ " tpb=" #T_, [&] { kfill<K_, NT_, T_><<<unsigned(n / 4 / T_ / K_), T_>>>(buf); })
  RUNK(1, false, 256); RUNK(2, false, 256); RUNK(3, false, 256); RUNK(4, false, 256); RUNK(8, false, 256); RUNK(24, false, 256);
  RUNK(1, true, 256); RUNK(2, true, 256); RUNK(4, true, 256); RUNK(8, true, 256); RUNK(24, true, 256);
  RUNK(1, false, 64); RUNK(2, false, 64); RUNK(4, false, 64); RUNK(8, false, 64); RUNK(24, false, 64); RUNK(96, false, 64);
  RUNK(1, false, 1024); RUNK(4, false, 1024);
#define RUNC(K_, NT_) run("kcol3 K=" #K_ " nt=" #NT_, [&] { kcol3<K_, NT_><<<unsigned(n / 192 / K_), 64>>>(buf); })
  RUNC(1, false); RUNC(2, false); RUNC(4, false); RUNC(8, false); RUNC(16, false); RUNC(32, false); RUNC(128, false);
  RUNC(1, true); RUNC(4, true); RUNC(16, true); RUNC(128, true);
#define RUNW(K_, NT_) run("kcol3w (256 thr) K=" #K_ " nt=" #NT_, [&] { kcol3w<K_, NT_><<<unsigned(n / 192 / K_ / 4), 256>>>(buf); })
  RUNW(1, false); RUNW(2, false); RUNW(4, false); RUNW(1, true); RUNW(2, true); RUNW(4, true);
  return 0;
}
