// store_cfg5.hip -- experiment: the store pattern of J-assembly at cfg5 (M = 900 rows = 3600 B per
// column, P = 300 columns, 300 units = 5 chunks of 64 lanes x 12 B), four waves per instance.
//   chunk-outer: for chunk { for column { store } }   (pieces of a column written far apart in time)
//   column-outer: for column { for chunk { store } }  (a column's 3600 B written back to back)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int M = 900, P = 300, U = 300;
template <bool NT>
__device__ __forceinline__ void st3(float* o, float v) {
  if (NT) {
    __builtin_nontemporal_store(v, o);
    __builtin_nontemporal_store(v, o + 1);
    __builtin_nontemporal_store(v, o + 2);
  } else {
    o[0] = v, o[1] = v, o[2] = v;
  }
}
template <bool NT, bool COLOUTER>
__global__ void __launch_bounds__(256) k(float* jac) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* jb = jac + size_t(blockIdx.x) * M * P;
  float v = float(blockIdx.x);
  if (COLOUTER) {
    for (int c = wave; c < P; c += 4) {
#pragma unroll
      for (int ch = 0; ch < 5; ++ch) {
        const int u = 64 * ch + lane;
        if (u < U) st3<NT>(jb + size_t(c) * M + 3 * u, v);
      }
    }
  } else {
    for (int ch = 0; ch < 5; ++ch) {
      const int u = 64 * ch + lane;
      for (int c = wave; c < P; c += 4) {
        if (u < U) st3<NT>(jb + size_t(c) * M + 3 * u, v);
      }
    }
  }
}
int main() {
  const int B = 8192;
  const size_t n = size_t(B) * M * P;
  float* buf;
  hipMalloc(&buf, n * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-28s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, n * 4 / ms / 1e6);
  };
  run("chunk-outer", [&] { k<false, false><<<B, 256>>>(buf); });
  run("chunk-outer nt", [&] { k<true, false><<<B, 256>>>(buf); });
  run("column-outer", [&] { k<false, true><<<B, 256>>>(buf); });
  run("column-outer nt", [&] { k<true, true><<<B, 256>>>(buf); });
  return 0;
}
