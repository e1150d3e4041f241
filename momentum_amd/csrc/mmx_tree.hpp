// mmx_tree.hpp -- device helpers of the tree-structured normal equations (mmx_fused.hip).  The formulas are
// derived and validated against the explicit Jacobian in tests/tree_algebra_np.py.
#pragma once

#include "mmx_device.hpp"

namespace mmx {

// translationAxis column d of joint a = column d of parent.toLinear() (joint_state.cpp:36-42)
__device__ __forceinline__ F3 transAxisCol(const float* js, int parent, int d) {
  if (parent < 0) {
    return F3{d == 0 ? 1.f : 0.f, d == 1 ? 1.f : 0.f, d == 2 ? 1.f : 0.f};
  }
  const float* p = js + kJs * parent;
  const F3 c = qmatCol(Q4{p[3], p[4], p[5], p[6]}, d);
  return F3{c.x * p[7], c.y * p[7], c.z * p[7]};
}

constexpr int kC2 = 17; // second-order channels per joint: m0 | m1(3) | M2 (xx xy xz yy yz zz) | M2 of directions (6) | pad (odd stride)
constexpr int kC2Used = 16;
constexpr int kC1 = 7; // first-order channels per joint: F(3) | N(3) | D (odd stride)

// Tree sums as tiny exact-fp32 MFMA products with a 0/1 mask matrix built on the fly (joints are
// indexed by DFS position, so "m is in the subtree of k" is k <= m < k + subSize[k]):
//   kSubtree = true :  out[k][c] = sum over the loaded positions m in the subtree of k of in[m][c]
//                      (adjoint pass: subtree sums; only joints that carry units have non-zero rows)
//   kSubtree = false:  out[k][c] = sum over the ancestors-or-self a of k of in[a][c]
//                      (tangent pass: prefix sums along the parent chain)
// v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (exact products by 0/1), hence deterministic.
// Tiles of 16 rows x 16 channels are dealt to the four waves.
// UN: k-steps per trip -- their table and value reads go out together, then the chained MFMAs (pays for long k ranges:
// hundreds of joints; the 72-joint fused solve is faster with 1)
// kRange (subtree sums, optional): per row tile the range [lo, hi) of loaded-position indices that can fall into a
// subtree of one of the tile's rows (treeSumRanges below); everything outside multiplies by an exact zero.
// I: element type of the two integer tables (int32_t in global memory / the tree kernels' LDS, int16_t in the one-launch
// solve's LDS, where every float counts towards a fourth workgroup per CU)
template <int NC, bool kSubtree, int STRIDE = NC, int UN = 1, class I = int32_t> // NC channels per row, rows STRIDE floats apart
__device__ __forceinline__ void treeSumT(
    const I* subSize,
    const I* loadedPos,
    int numLoaded,
    const float* in,
    float* out,
    int J,
    int wave,
    int numWaves,
    int lane,
    const int32_t* kRange = nullptr) {
  const int K = kSubtree ? numLoaded : J;
  const int rowTiles = (J + 15) >> 4;
  constexpr int colTiles = (NC + 15) / 16;
  const int i = lane & 15, g = lane >> 4;
  for (int t = wave; t < rowTiles * colTiles; t += numWaves) {
    const int rt = t / colTiles, ct = t - rt * colTiles;
    const int r = 16 * rt + i; // row of the A operand this lane feeds
    const int rsz = r < J ? subSize[r] : 0;
    const int c = 16 * ct + i; // column of the B operand this lane feeds
    v4f acc{0.f, 0.f, 0.f, 0.f};
    // prefix sums: only positions up to the tile's last row can be ancestors of its rows
    const int kEnd = kSubtree ? (kRange != nullptr ? kRange[2 * rt + 1] : K) : (K < 16 * rt + 16 ? K : 16 * rt + 16);
    const int kBegin = kSubtree && kRange != nullptr ? kRange[2 * rt] : 0;
    for (int k0 = kBegin; k0 < kEnd; k0 += 4 * UN) {
      int pp[UN], sz[UN];
      float bv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int kk = k0 + 4 * u + g;
        pp[u] = kSubtree ? loadedPos[kk < kEnd ? kk : 0] : kk;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int kk = k0 + 4 * u + g;
        sz[u] = kSubtree ? 0 : subSize[kk < kEnd ? pp[u] : 0];
        bv[u] = (kk < kEnd && c < NC) ? in[STRIDE * pp[u] + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int kk = k0 + 4 * u + g, p = pp[u];
        const bool m = kSubtree ? (p >= r && p < r + rsz) : (r < J && p <= r && r < p + sz[u]);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((kk < kEnd && m) ? 1.f : 0.f, bv[u], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int orow = 16 * rt + 4 * g + q, ocol = 16 * ct + i;
      if (orow < J && ocol < NC) {
        out[STRIDE * orow + ocol] = acc[q];
      }
    }
  }
}

// kRange of treeSumT: row tile rt covers DFS positions 16 rt .. 16 rt + 15; a loaded position p contributes to one of its
// rows r iff r <= p < r + subSize[r], hence 16 rt <= p < max_r (r + subSize[r]).  One thread per row tile, two binary
// searches in the ascending loadedPos.  Call by >= (J + 15) / 16 threads; barrier afterwards.
template <class I = int32_t>
__device__ __forceinline__ void treeSumRanges(const I* subSize, const I* loadedPos, int numLoaded, int J, int tid, int32_t* kRange) {
  const int rowTiles = (J + 15) >> 4;
  if (tid < rowTiles) {
    int hi = 0;
    for (int r = 16 * tid; r < 16 * tid + 16 && r < J; ++r) {
      hi = max(hi, r + subSize[r]);
    }
    auto lowerBound = [&](int v) { // first index with loadedPos[index] >= v
      int a = 0, b = numLoaded;
      while (a < b) {
        const int m = (a + b) >> 1;
        if (loadedPos[m] < v) {
          a = m + 1;
        } else {
          b = m;
        }
      }
      return a;
    };
    kRange[2 * tid] = lowerBound(16 * tid);
    kRange[2 * tid + 1] = lowerBound(hi);
  }
}

// J^T y component of one (joint, dof) from the first-order subtree sums (tests/tree_algebra_np.py jt_times)
__device__ __forceinline__ float sourceGradient(int joint, int dof, int parent, const float* js, const float* sb) {
  const float* a = js + kJs * joint;
  const F3 ta{a[0], a[1], a[2]};
  const F3 Fv{sb[0], sb[1], sb[2]};
  if (dof < 3) {
    return dot(transAxisCol(js, parent, dof), Fv);
  }
  if (dof < 6) {
    const float* ax = a + 8 + 3 * (dof - 3);
    const F3 Nv{sb[3], sb[4], sb[5]};
    return dot(F3{ax[0], ax[1], ax[2]}, Nv - cross(ta, Fv));
  }
  return kLn2 * (sb[6] - dot(ta, Fv));
}

} // namespace mmx
