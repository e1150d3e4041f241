// Probe: on which SIMD does wave w of a four-wave workgroup land (gfx950)?  The one-launch solve runs its single-wave phases
// (triangular solves, the panels' elimination chain) on wave 0: if wave 0 of every workgroup sat on the same SIMD of its CU, the
// four co-resident workgroups' chains would share one vector ALU.  Workgroups of 256 threads with 40 KB of LDS (four per CU,
// like the kernel), long enough to be co-resident; every wave records HW_ID (s_getreg_b32): SIMD, CU, SE, wave slot, + XCC_ID.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/wave_simd_placement.hip -o /tmp/wave_simd && /tmp/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256, 4) probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float v = float(threadIdx.x);
  for (int i = 0; i < spin; ++i) { // stay resident for a while
    lds[threadIdx.x] = v;
    __syncthreads();
    v = lds[(threadIdx.x + 1) & 255] * 1.0001f;
    __syncthreads();
  }
  if (lane == 0) {
    out[(blockIdx.x * 4 + wave) * 2] = hw;
    out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc;
  }
  if (v == 12345.678f) {
    out[0] = 0;
  }
}
int main() {
  const int B = 4096;
  unsigned* d;
  hipMalloc(&d, B * 8 * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  hipLaunchKernelGGL(probe, dim3(B), dim3(256), 40 * 1024, 0, d, 2000);
  std::vector<unsigned> h(B * 8);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (gfx940: [14:13]), ...
  long simdOfWave[4][4] = {};
  std::map<unsigned, std::vector<int>> perCu; // (xcc, se, sh, cu) -> workgroups
  int distinct = 0;
  for (int b = 0; b < B; ++b) {
    unsigned mask = 0;
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2];
      const int simd = (hw >> 4) & 3;
      simdOfWave[w][simd]++;
      mask |= 1u << simd;
    }
    distinct += mask == 0xf;
    const unsigned hw0 = h[b * 8], xcc = h[b * 8 + 1] & 0xf;
    perCu[(xcc << 16) | (hw0 & 0xff00)].push_back(b);
  }
  printf("workgroups whose four waves sit on four distinct SIMDs: %d of %d\n", distinct, B);
  for (int w = 0; w < 4; ++w) {
    printf("wave %d on SIMD 0..3: %ld %ld %ld %ld\n", w, simdOfWave[w][0], simdOfWave[w][1], simdOfWave[w][2], simdOfWave[w][3]);
  }
  printf("CUs seen: %zu\n", perCu.size());
  int shown = 0;
  for (auto& kv : perCu) {
    if (shown++ >= 6) {
      break;
    }
    printf("cu key %06x: first workgroups", kv.first);
    for (size_t i = 0; i < kv.second.size() && i < 8; ++i) {
      const int b = kv.second[i];
      printf("  %d(w0: simd %u slot %u)", b, (h[b * 8] >> 4) & 3, h[b * 8] & 0xf);
    }
    printf("\n");
  }
  return 0;
}
