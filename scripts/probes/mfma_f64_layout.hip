// Register layout of v_mfma_f64_16x16x4_f64 on gfx950, measured: lane l feeds A[l % 16][l / 16] and B[l / 16][l % 16] (assumed, then
// checked by the result); which C[i][j] does register r of lane l hold?   hipcc --offload-arch=gfx950 mfma_f64_layout.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x, i = l & 15, k = l >> 4;
  // C[i][j] = (i + 1) + 100 (j + 1): k = 0 carries A = i + 1, B = 1; k = 1 carries A = 1, B = 100 (j + 1); k = 2, 3 are zero
  const double a = k == 0 ? double(i + 1) : (k == 1 ? 1.0 : 0.0);
  const double b = k == 0 ? 1.0 : (k == 1 ? 100.0 * double(i + 1) : 0.0);
  v4d c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    out[4 * l + r] = c[r];
  }
}
int main() {
  double* d;
  hipMalloc(&d, 256 * sizeof(double));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  double h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  bool std_layout = true;
  for (int l = 0; l < 64; ++l) {
    for (int r = 0; r < 4; ++r) {
      const int v = int(h[4 * l + r] + 0.5), j = v / 100 - 1, i = v % 100 - 1;
      if (l < 20 || (l & 15) == 0) {
        printf("lane %2d reg %d -> C[%2d][%2d]\n", l, r, i, j);
      }
      std_layout = std_layout && i == 4 * (l >> 4) + r && j == (l & 15);
    }
  }
  printf("layout C[4 (l / 16) + r][l %% 16]: %s\n", std_layout ? "YES" : "NO");
  return 0;
}
