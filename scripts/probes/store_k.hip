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
// the two-kernel J-assembly's second kernel in miniature: a 256-thread block = 4 adjacent columns of one
// instance; every lane first loads its unit (5 floats from a 1.3 KB per-instance record that a first kernel
// would have written), every wave its column's joint state (8 floats, uniform address), then one 12-byte
// store per lane.  What bandwidth survives the dependent loads?
template <bool NT>
__global__ void __launch_bounds__(256) kcol3w_dep(float* o, const float* __restrict__ ub, const float* __restrict__ sb) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x >> 5, col = 4 * (blockIdx.x & 31) + wave;
  const float* u = ub + (size_t(b) * 64 + lane) * 5;
  const float vx = u[0], vy = u[1], vz = u[2], sg = u[3];
  const int tin = __float_as_int(u[4]);
  const float* st = sb + (size_t(b) * 72 + (col * 37) % 72) * 8; // wave-uniform
  const float tx = st[0], ty = st[1], tz = st[2], ax = st[3], ay = st[4], az = st[5];
  const float w = (tin & 1) ? 1.f : 0.f;
  const float ox = vx - tx, oy = vy - ty, oz = vz - tz;
  float* p = o + (size_t(b) * 128 + col) * 192 + 3 * lane;
  const float gx = sg * (ay * oz - az * oy) * w, gy = sg * (az * ox - ax * oz) * w, gz = sg * (ax * oy - ay * ox) * w;
  if (NT) {
    __builtin_nontemporal_store(gx, p);
    __builtin_nontemporal_store(gy, p + 1);
    __builtin_nontemporal_store(gz, p + 2);
  } else {
    p[0] = gx, p[1] = gy, p[2] = gz;
  }
}
// the same with 8 columns per block (2 per wave): half the blocks, the loads amortised over two stores
template <bool NT>
__global__ void __launch_bounds__(256) kcol3w_dep2(float* o, const float* __restrict__ ub, const float* __restrict__ sb) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.x >> 4, col0 = 8 * (blockIdx.x & 15) + 2 * wave;
  const float* u = ub + (size_t(b) * 64 + lane) * 5;
  const float vx = u[0], vy = u[1], vz = u[2], sg = u[3];
  const int tin = __float_as_int(u[4]);
  const float w = (tin & 1) ? 1.f : 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int col = col0 + k;
    const float* st = sb + (size_t(b) * 72 + (col * 37) % 72) * 8;
    const float tx = st[0], ty = st[1], tz = st[2], ax = st[3], ay = st[4], az = st[5];
    const float ox = vx - tx, oy = vy - ty, oz = vz - tz;
    float* p = o + (size_t(b) * 128 + col) * 192 + 3 * lane;
    const float gx = sg * (ay * oz - az * oy) * w, gy = sg * (az * ox - ax * oz) * w, gz = sg * (ax * oy - ay * ox) * w;
    if (NT) {
      __builtin_nontemporal_store(gx, p);
      __builtin_nontemporal_store(gy, p + 1);
      __builtin_nontemporal_store(gz, p + 2);
    } else {
      p[0] = gx, p[1] = gy, p[2] = gz;
    }
  }
}
// general shape: WPB waves per block, CPW columns per wave (block = WPB * CPW adjacent columns of one instance)
template <int WPB, int CPW, bool XCD>
__global__ void __launch_bounds__(64 * WPB) kcol3w_shape(float* o, const float* __restrict__ ub, const float* __restrict__ sb) {
  constexpr int CPB = WPB * CPW, G = 128 / CPB;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int b, cg;
  if (XCD) { // all blocks of an instance on one XCD (blocks go round-robin to the 8 XCDs)
    const int x = blockIdx.x & 7, r = blockIdx.x >> 3;
    cg = r % G;
    b = (r / G) * 8 + x;
  } else {
    b = blockIdx.x / G, cg = blockIdx.x % G;
  }
  const int col0 = CPB * cg + CPW * wave;
  const float* u = ub + (size_t(b) * 64 + lane) * 5;
  const float vx = u[0], vy = u[1], vz = u[2], sg = u[3];
  const int tin = __float_as_int(u[4]);
  const float w = (tin & 1) ? 1.f : 0.f;
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    const int col = col0 + k;
    const float* st = sb + (size_t(b) * 72 + (col * 37) % 72) * 8;
    const float tx = st[0], ty = st[1], tz = st[2], ax = st[3], ay = st[4], az = st[5];
    const float ox = vx - tx, oy = vy - ty, oz = vz - tz;
    float* p = o + (size_t(b) * 128 + col) * 192 + 3 * lane;
    __builtin_nontemporal_store(sg * (ay * oz - az * oy) * w, p);
    __builtin_nontemporal_store(sg * (az * ox - ax * oz) * w, p + 1);
    __builtin_nontemporal_store(sg * (ax * oy - ay * ox) * w, p + 2);
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
  float *ub, *sb;
  hipMalloc(&ub, size_t(4096) * 64 * 5 * 4);
  hipMalloc(&sb, size_t(4096) * 72 * 8 * 4);
  hipMemset(ub, 0x3f, size_t(4096) * 64 * 5 * 4); // 0x3f3f3f3f = 0.747f, odd as an int: non-zero data, w = 1
  hipMemset(sb, 0x3e, size_t(4096) * 72 * 8 * 4);
  run("kcol3w_dep (4 cols/block)", [&] { kcol3w_dep<false><<<4096 * 32, 256>>>(buf, ub, sb); });
  run("kcol3w_dep nt (4 cols/block)", [&] { kcol3w_dep<true><<<4096 * 32, 256>>>(buf, ub, sb); });
  run("kcol3w_dep2 (8 cols/block)", [&] { kcol3w_dep2<false><<<4096 * 16, 256>>>(buf, ub, sb); });
  run("kcol3w_dep2 nt (8 cols/block)", [&] { kcol3w_dep2<true><<<4096 * 16, 256>>>(buf, ub, sb); });
#define RUNS(WPB_, CPW_, X_) run("shape wpb=" #WPB_ " cpw=" #CPW_ " xcd=" #X_, [&] { kcol3w_shape<WPB_, CPW_, X_><<<4096 * (128 / (WPB_ * CPW_)), 64 * WPB_>>>(buf, ub, sb); })
  RUNS(4, 1, false); RUNS(4, 1, true); RUNS(4, 2, false); RUNS(4, 2, true); RUNS(4, 4, false); RUNS(4, 4, true); RUNS(4, 8, true);
  RUNS(2, 2, true); RUNS(2, 4, true); RUNS(2, 8, true); RUNS(1, 4, true); RUNS(1, 8, true); RUNS(1, 16, true); RUNS(8, 1, true); RUNS(8, 2, true);
  return 0;
}
