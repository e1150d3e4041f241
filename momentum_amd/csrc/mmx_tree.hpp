// mmx_tree.hpp -- device helpers of the tree-structured normal equations (shared by the fused
// solver, mmx_fused.hip, and the wave-per-instance pipeline, mmx_pipeline.hip).  The formulas are
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

constexpr int kC2 = 16; // second-order channels per joint: m0 | m1(3) | M2 (xx xy xz yy yz zz) | M2 of directions (6)
constexpr int kC1 = 8; // first-order channels per joint: F(3) | N(3) | D | pad

// Tree sums as tiny exact-fp32 MFMA products with a 0/1 mask matrix built on the fly (joints are
// indexed by DFS position, so "m is in the subtree of k" is k <= m < k + subSize[k]):
//   kSubtree = true :  out[k][c] = sum over the loaded positions m in the subtree of k of in[m][c]
//                      (adjoint pass: subtree sums; only joints that carry units have non-zero rows)
//   kSubtree = false:  out[k][c] = sum over the ancestors-or-self a of k of in[a][c]
//                      (tangent pass: prefix sums along the parent chain)
// v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (exact products by 0/1), hence deterministic.
// Tiles of 16 rows x 16 channels are dealt to the four waves.
template <int NC, bool kSubtree>
__device__ __forceinline__ void treeSumT(
    const int32_t* subSize,
    const int32_t* loadedPos,
    int numLoaded,
    const float* in,
    float* out,
    int J,
    int wave,
    int numWaves,
    int lane) {
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
    for (int k0 = 0; k0 < K; k0 += 4) {
      const int kk = k0 + g;
      float av = 0.f, bv = 0.f;
      if (kk < K) {
        const int p = kSubtree ? loadedPos[kk] : kk;
        const bool m = kSubtree ? (p >= r && p < r + rsz) : (r < J && p <= r && r < p + subSize[p]);
        av = m ? 1.f : 0.f;
        bv = c < NC ? in[NC * p + c] : 0.f;
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int orow = 16 * rt + 4 * g + q, ocol = 16 * ct + i;
      if (orow < J && ocol < NC) {
        out[NC * orow + ocol] = acc[q];
      }
    }
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

// The 16 floats the H assembly needs from one column source (joint, dof) (phase E of the fused
// kernel): G0(3) AX(3) TR(1) | AL(3) BV(3) BS(1) | GJ(1) pad(1).  m2 = the joint's second-order
// subtree sums (kC2 floats), m1 = its first-order subtree sums (kC1 floats).
__device__ __forceinline__ void
sourceTable(int joint, int dof, int parent, const float* js, const float* m2, const float* m1s, float* o) {
  const float* a = js + kJs * joint;
  const F3 ta{a[0], a[1], a[2]};
  const float m0 = m2[0];
  const F3 m1{m2[1], m2[2], m2[3]};
  F3 al, bv{0.f, 0.f, 0.f}, g0, ax;
  float bs = 0.f, tr;
  if (dof < 3) {
    al = transAxisCol(js, parent, dof);
    g0 = m0 * al;
    ax = cross(m1, al);
    tr = dot(al, m1);
  } else if (dof < 6) {
    const float* w = a + 8 + 3 * (dof - 3);
    const F3 om{w[0], w[1], w[2]};
    al = F3{0.f, 0.f, 0.f} - cross(om, ta);
    bv = om;
    g0 = m0 * al + cross(om, m1);
    // axial([om]x M) = tr(M) om - M om, for the point and the direction second moments
    const float t2 = (m2[4] + m2[7] + m2[9]) + (m2[10] + m2[13] + m2[15]);
    const F3 Mo{
        (m2[4] + m2[10]) * om.x + (m2[5] + m2[11]) * om.y + (m2[6] + m2[12]) * om.z,
        (m2[5] + m2[11]) * om.x + (m2[7] + m2[13]) * om.y + (m2[8] + m2[14]) * om.z,
        (m2[6] + m2[12]) * om.x + (m2[8] + m2[14]) * om.y + (m2[9] + m2[15]) * om.z};
    ax = cross(m1, al) + (t2 * om - Mo);
    tr = dot(al, m1);
  } else {
    al = F3{0.f, 0.f, 0.f} - kLn2 * ta;
    bs = kLn2;
    g0 = m0 * al + kLn2 * m1;
    ax = cross(m1, al);
    tr = dot(al, m1) + kLn2 * (m2[4] + m2[7] + m2[9]);
  }
  o[0] = g0.x, o[1] = g0.y, o[2] = g0.z;
  o[3] = ax.x, o[4] = ax.y, o[5] = ax.z;
  o[6] = tr;
  o[7] = al.x, o[8] = al.y, o[9] = al.z;
  o[10] = bv.x, o[11] = bv.y, o[12] = bv.z;
  o[13] = bs;
  o[14] = sourceGradient(joint, dof, parent, js, m1s);
  o[15] = 0.f;
}

// contraction of a (deep, anc) source pair: G0.AL + AX.BV + TR*BS
__device__ __forceinline__ float sourcePairTerm(const float* srcT, int deep, int anc) {
  const float4 d0v = *reinterpret_cast<const float4*>(srcT + kSrc * deep);
  const float4 d1v = *reinterpret_cast<const float4*>(srcT + kSrc * deep + 4);
  const float4 a1v = *reinterpret_cast<const float4*>(srcT + kSrc * anc + 4);
  const float4 a2v = *reinterpret_cast<const float4*>(srcT + kSrc * anc + 8);
  const float4 a3v = *reinterpret_cast<const float4*>(srcT + kSrc * anc + 12);
  // (G0 = d0v.xyz, AX = d0v.w d1v.xy, TR = d1v.z ; AL = a1v.w a2v.xy, BV = a2v.zw a3v.x, BS = a3v.y)
  return d0v.x * a1v.w + d0v.y * a2v.x + d0v.z * a2v.y + d0v.w * a2v.z + d1v.x * a2v.w + d1v.y * a3v.x + d1v.z * a3v.y;
}

} // namespace mmx
