// store_pattern.hip -- experiment: what write bandwidth does the J-assembly STORE PATTERN reach on its
// own (no kinematics)?  One wave owns one instance's column-major M x P Jacobian (M = 192 floats =
// 768 B per column, P = 128 columns = 96 KB contiguous) and writes it column by column.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/store_pattern scripts/store_pattern.hip
// Variants (argv[1] selects a subset, default all):
//   0 fill          grid-stride 16 B per lane, perfectly sequential (the "fill" ceiling)
//   1 seq3          12 B per lane, columns ascending
//   2 perm3         12 B per lane, columns in a fixed pseudo-random order
//   3 seq4          16 B per lane on 48 lanes, columns ascending
//   4 seq3 nt / 5 perm3 nt / 6 seq4 nt    the same with non-temporal stores
//   7 seq3 nt, 4 instances per 256-thread block
//   8 seq4x2 nt     two columns per store instruction pair (1536 B in flight per lane pair)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);        \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

constexpr int M = 192, P = 128;

__global__ void fillKernel(float4* o, size_t n4) {
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    o[i] = float4{1.f, 2.f, 3.f, 4.f};
  }
}

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
template <bool NT>
__device__ __forceinline__ void st4(float* o, float v) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 x = {v, v, v, v};
  if (NT) {
    __builtin_nontemporal_store(x, reinterpret_cast<v4*>(o));
  } else {
    *reinterpret_cast<v4*>(o) = x;
  }
}

// WPB instances per block (one wave each)
template <bool NT, bool PERM, int WPB, int WORK>
__global__ void __launch_bounds__(64 * WPB) cols3(float* jac, const int* order, int B) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (b >= B) {
    return;
  }
  float* jb = jac + size_t(b) * M * P + 3 * lane;
  float v = float(b);
  for (int i = 0; i < P; ++i) {
    const int c = PERM ? order[i] : i;
#pragma unroll
    for (int w = 0; w < WORK; ++w) {
      v = __builtin_fmaf(v, 1.0001f, 0.5f);
    }
    st3<NT>(jb + size_t(c) * M, v);
  }
}

template <bool NT, bool PERM>
__global__ void __launch_bounds__(64) cols4(float* jac, const int* order, int B) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  float* jb = jac + size_t(b) * M * P + 4 * lane;
  float v = float(b);
  if (lane < 48) {
    for (int i = 0; i < P; ++i) {
      const int c = PERM ? order[i] : i;
      st4<NT>(jb + size_t(c) * M, v);
    }
  }
}

// all 64 lanes, 16 B each: one store instruction covers 1024 B = 1 1/3 columns; three instructions
// cover four columns (only possible when the four columns are adjacent in memory)
template <bool NT>
__global__ void __launch_bounds__(64) cols4full(float* jac, int B) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  float* jb = jac + size_t(b) * M * P + 4 * lane;
  float v = float(b);
  for (int i = 0; i < M * P / 256; ++i) {
    st4<NT>(jb + size_t(i) * 256, v);
  }
}


// seq3 with a per-instance rotation of the column order: instance b starts at column (b * K) % P
template <bool NT>
__global__ void __launch_bounds__(64) rot3(float* jac, int B, int K) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  float* jb = jac + size_t(b) * M * P + 3 * lane;
  float v = float(b);
  const int c0 = (b * K) % P;
  for (int i = 0; i < P; ++i) {
    const int c = (c0 + i) % P;
    st3<NT>(jb + size_t(c) * M, v);
  }
}

// limited occupancy: LDS bytes per 64-thread block chosen by the host
template <bool NT, bool PERM>
__global__ void __launch_bounds__(64) occ3(float* jac, const int* order, int B) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x;
  float* jb = jac + size_t(b) * M * P + 3 * lane;
  float v = float(b);
  if (B < 0) {
    lds[lane] = v;
  }
  for (int i = 0; i < P; ++i) {
    const int c = PERM ? order[i] : i;
    st3<NT>(jb + size_t(c) * M, v);
  }
}

// one 256-thread block per instance: wave w writes columns 4 i + w (four adjacent columns at a time)
template <bool NT>
__global__ void __launch_bounds__(256) wpi4(float* jac, int B) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x;
  float* jb = jac + size_t(b) * M * P + 3 * lane;
  float v = float(b);
  for (int i = wave; i < P; i += 4) {
    st3<NT>(jb + size_t(i) * M, v);
  }
}
// one 256-thread block per instance, 16 B per lane, 4 KB per round, sequential
template <bool NT>
__global__ void __launch_bounds__(256) blk4k(float* jac, int B) {
  const int b = blockIdx.x;
  float* jb = jac + size_t(b) * M * P + 4 * threadIdx.x;
  float v = float(b);
  for (int i = 0; i < M * P / 1024; ++i) {
    st4<NT>(jb + size_t(i) * 1024, v);
  }
}
// fill-like decomposition: grid = B * 8, each 64-thread block writes 16 adjacent columns (12 KB) and exits
template <bool NT>
__global__ void __launch_bounds__(64) piece3(float* jac, int B) {
  const int lane = threadIdx.x & 63;
  float* jb = jac + size_t(blockIdx.x) * M * 16 + 3 * lane;
  float v = float(blockIdx.x);
  for (int i = 0; i < 16; ++i) {
    st3<NT>(jb + size_t(i) * M, v);
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096;
  const int reps = 20;
  const size_t n = size_t(B) * M * P;
  float* jac;
  CK(hipMalloc(&jac, n * 4));
  std::vector<int> order(P);
  for (int i = 0; i < P; ++i) {
    order[i] = (i * 37 + 11) % P; // 37 coprime to 128
  }
  int* dOrder;
  CK(hipMalloc(&dOrder, P * 4));
  CK(hipMemcpy(dOrder, order.data(), P * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) {
      launch();
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
      launch();
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("B=%d %-28s %8.1f us  %7.0f GB/s\n", B, name, ms * 1e3, n * 4 / ms / 1e6);
  };
  run("fill 16B/lane grid-stride", [&] { fillKernel<<<256 * 8, 256>>>(reinterpret_cast<float4*>(jac), n / 4); });
  run("fill 16B/lane 1 pass", [&] { fillKernel<<<unsigned(n / 4 / 256), 256>>>(reinterpret_cast<float4*>(jac), n / 4); });
  run("seq3", [&] { cols3<false, false, 1, 0><<<B, 64>>>(jac, dOrder, B); });
  run("perm3", [&] { cols3<false, true, 1, 0><<<B, 64>>>(jac, dOrder, B); });
  run("seq4(48 lanes)", [&] { cols4<false, false><<<B, 64>>>(jac, dOrder, B); });
  run("seq3 nt", [&] { cols3<true, false, 1, 0><<<B, 64>>>(jac, dOrder, B); });
  run("perm3 nt", [&] { cols3<true, true, 1, 0><<<B, 64>>>(jac, dOrder, B); });
  run("seq4(48 lanes) nt", [&] { cols4<true, false><<<B, 64>>>(jac, dOrder, B); });
  run("perm4(48 lanes) nt", [&] { cols4<true, true><<<B, 64>>>(jac, dOrder, B); });
  run("seq3 nt 4 inst/block", [&] { cols3<true, false, 4, 0><<<(B + 3) / 4, 256>>>(jac, dOrder, B); });
  run("perm3 nt 4 inst/block", [&] { cols3<true, true, 4, 0><<<(B + 3) / 4, 256>>>(jac, dOrder, B); });
  run("full 64x16B nt", [&] { cols4full<true><<<B, 64>>>(jac, B); });
  run("full 64x16B", [&] { cols4full<false><<<B, 64>>>(jac, B); });
  run("perm3 nt +32 fma/col", [&] { cols3<true, true, 1, 32><<<B, 64>>>(jac, dOrder, B); });
  run("perm3 nt +128 fma/col", [&] { cols3<true, true, 1, 128><<<B, 64>>>(jac, dOrder, B); });
  run("perm3 +128 fma/col", [&] { cols3<false, true, 1, 128><<<B, 64>>>(jac, dOrder, B); });

  for (int K : {1, 3, 5, 11, 16, 21, 43, 64}) {
    char nm[64];
    snprintf(nm, 64, "rot3 nt K=%d", K);
    run(nm, [&] { rot3<true><<<B, 64>>>(jac, B, K); });
    snprintf(nm, 64, "rot3 K=%d", K);
    run(nm, [&] { rot3<false><<<B, 64>>>(jac, B, K); });
  }
  for (int kb : {10, 20, 40, 80}) {
    char nm[64];
    snprintf(nm, 64, "occ seq3 nt lds=%dKB", kb);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&occ3<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024));
    run(nm, [&] { occ3<true, false><<<B, 64, kb * 1024>>>(jac, dOrder, B); });
    snprintf(nm, 64, "occ perm3 lds=%dKB", kb);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&occ3<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024));
    run(nm, [&] { occ3<false, true><<<B, 64, kb * 1024>>>(jac, dOrder, B); });
  }
  run("wpi4 nt", [&] { wpi4<true><<<B, 256>>>(jac, B); });
  run("wpi4", [&] { wpi4<false><<<B, 256>>>(jac, B); });
  run("blk4k nt", [&] { blk4k<true><<<B, 256>>>(jac, B); });
  run("blk4k", [&] { blk4k<false><<<B, 256>>>(jac, B); });
  run("piece3 nt (12KB/block)", [&] { piece3<true><<<B * 8, 64>>>(jac, B); });
  run("piece3 (12KB/block)", [&] { piece3<false><<<B * 8, 64>>>(jac, B); });
  return 0;
}
