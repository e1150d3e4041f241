// Probe: issue rate / dependent latency of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per instruction, one wave per SIMD and
// two).  hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int CHAINS, bool F64>
__global__ void probe(long long* out, double seed) {
  v4d c[CHAINS];
  v4f cf[CHAINS];
  for (int i = 0; i < CHAINS; ++i) {
    c[i] = v4d{seed, seed, seed, seed};
    cf[i] = v4f{float(seed), float(seed), float(seed), float(seed)};
  }
  const double a = seed + threadIdx.x, b = seed * 0.5;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (F64) {
        c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
      } else {
        cf[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(float(a), float(b), cf[i], 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < CHAINS; ++i) {
    s += c[i][0] + c[i][1] + c[i][2] + c[i][3] + cf[i][0] + cf[i][1] + cf[i][2] + cf[i][3];
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = t1 - t0;
  }
  if (s == 12345.678) {
    out[1] = 1;
  }
}
template <int CHAINS, bool F64>
void run(const char* name, int threads, int blocks = 1) {
  long long* d;
  hipMalloc(&d, 16);
  hipLaunchKernelGGL((probe<CHAINS, F64>), dim3(blocks), dim3(threads), 0, 0, d, 1.0);
  hipLaunchKernelGGL((probe<CHAINS, F64>), dim3(blocks), dim3(threads), 0, 0, d, 1.0);
  long long h[2];
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-28s chains %d, %3d threads x %4d blocks: %6.1f cycles per instruction of a wave\n", name, CHAINS, threads, blocks, double(h[0]) / (64.0 * CHAINS));
  hipFree(d);
}
int main() {
  run<1, true>("v_mfma_f64_16x16x4_f64", 256);
  run<2, true>("v_mfma_f64_16x16x4_f64", 256);
  run<4, true>("v_mfma_f64_16x16x4_f64", 256);
  run<4, true>("v_mfma_f64_16x16x4_f64", 512);
  run<4, true>("v_mfma_f64_16x16x4_f64", 256, 512); // the whole chip, two workgroups per CU
  run<4, true>("v_mfma_f64_16x16x4_f64", 256, 2048);
  run<1, false>("v_mfma_f32_16x16x4_f32", 256);
  run<4, false>("v_mfma_f32_16x16x4_f32", 256);
  run<4, false>("v_mfma_f32_16x16x4_f32", 512);
  return 0;
}
