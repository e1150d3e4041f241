// mmx_fused.hip -- the fused batched Gauss-Newton solve kernel (gfx950 / CDNA4, wave64).
//
// One workgroup (4 wavefronts) owns one skeleton instance for the WHOLE solve: all iterations of
// SolverT::solve (momentum/solver/solver.cpp:50-128) with GaussNewtonSolverT::doIteration
// (momentum/solver/gauss_newton_solver.cpp:224-280) run inside one launch, theta and every
// intermediate live in LDS / registers, and the dense Jacobian is never formed:
//
//   A  joint parameters = transform * theta + offsets, half-angle sin/cos, exp2       (parameter_transform.cpp:110-124)
//   B  forward kinematics by tree level                                               (joint_state.cpp:22-65)
//   C  constraint vectors ("units"): residual rows, sigma, error                      (position/orientation evalFunction)
//   D  per-joint SUBTREE sums of the units' moments (joints in DFS order => a subtree is a range)
//   E  per column source (joint, dof): alpha/B of d(unit)/d(dof) and its moment contractions
//   F  g = J^T r from the first-order sums (adjoint pass)
//   G  H = J^T J + lambda I from the second-order sums, each 16x16 tile computed straight into
//      the MFMA accumulator registers of the wave that owns it (O(1) work per entry, independent
//      of the number of constraint rows)
//   H  blocked right-looking Cholesky on 16x16 tiles: tiles stay in registers, finished L panels
//      go to LDS; trailing updates are v_mfma_f32_16x16x4_f32
//   I  forward / backward substitution with the LDS-resident factor
//   J  one refinement step of the corrected seminormal equations, rho = J^T (r - J d) - lambda d,
//      with J d by a tangent pass down the tree and J^T w by the adjoint pass (never forming J);
//      this makes the fp32 step agree with the reference's double-precision solve to ~1e-7
//   K  theta -= delta, error history, convergence test
//
// The formulas of D-G and J are derived and validated against the explicit Jacobian in
// tests/tree_algebra_np.py / tests/test_tree_algebra.py.
#include "mmx_device.hpp"
#include "mmx_kernels.hpp"
#include "mmx_mixed.hpp"
#include "mmx_tree.hpp"

#include <cfloat>
#include <cstdlib>

namespace mmx {

struct FusedLds {
  // ---- loaded once per launch (batch-shared integer tables and the parameter-transform CSR)
  // source SLOTS: slot c < NP is the primary source of column c (pad columns: weight 0), slots >= NP are
  // the further sources of multi-source columns: extras of column c = NP + mStart[c] .. NP + mStart[c+1] - 1
  int16_t* mStart; // [NP+1] column -> number of extra slots before it
  int* mTin; // [nsrc] DFS interval of the slot's joint: tin | tout << 16
  int* mInfo; // [nsrc] joint | dof << 12 | (parent + 1) << 16
  float* mW; // [nsrc]
  // ---- per iteration
  float* th; // [P] theta (full parameter space)
  float* js; // [kJs J] world t(3) q(4) s(1) | rotation axes (9)
  float* up; // [3 U] unit world vector
  float* ur; // [3 U] scaled residual rows r
  float* uy; // [3 U] sigma * (r or w): input of the adjoint pass
  float* us; // [U]   sigma
  float* own1; // [kC1 J] first-order sums over the joint's own units (by DFS position)
  float* sub1; // [kC1 J] ... over the joint's subtree
  float* g; // [NP]
  float* d0; // [NP]
  float* rho; // [NP]
  float* invDiag; // [NP] 1 / L(i,i)
  float* dfull; // [P]
  float* jd; // [7 J]
  float* tanOwn; // [kTan J]
  float* tanPre; // [kTan J]
  double* red; // [8]
  int* flags; // [4]: stop | not positive definite (this iteration) | status | unused
  // ---- rows of the further joint error functions and ellipsoid limits (kGen instantiations): dense in LDS
  float* gEv; // [GT][kGenEv] per constraint: vp(3) vn(3) sigma*dp(9) sigma*dn(9) tin row flags tinStop
  float* gRes; // [rowsGp] residual rows
  float* gW; // [rowsGp] r - J d of the refinement
  float* gJ; // [rowsGp][gst] their Jacobian rows over the solve columns (pad rows / columns zero)
  // ---- one region, two lives: assembly scratch (phases A-G), then the Cholesky factor (H-J)
  double* fkA; // [kFkCh][fkPad(J)] the two buffers of the pointer-jumping FK (mmx_device.hpp fkJumpRoundsD); fkB lies over
  double* fkB; // own2 / sub2, which phase D writes after FK is done
  float* own2; // [kC2 J]
  float* umom; // [kUmom U] per-unit moment contributions (phase D only)
  float* sub2; // [kC2 J]
  float* L; // [T][256] tiles; a diagonal tile holds L_kk (lower triangle) and L_kk^-T (strict upper triangle)
  // ---- aliases the refinement scratch (dfull, jd, tanOwn, tanPre): dead before phase J starts
  float* srcT; // [15][srcStride]: weighted D (7 channels) | weighted A (7) | weighted J^T r share (1), channel-major
};

// Views of the batch-shared tables after they were copied into LDS.  Plain local structs whose
// pointers come straight from the LDS carve, so the compiler keeps the LDS address space (ds_read)
// instead of falling back to flat loads.
struct RigView {
  int32_t J, P, R, numLevels, jumpRounds;
  const int32_t* parent;
  const float* preRot; // stays in global memory (read once per joint per iteration)
  const float* offset; // "
  const int32_t* ptOuter;
  const int32_t* ptInner;
  const float* ptValue;
  const float* ptOffsets; // "
  bool hasOffsets; // some entry of ptOffsets is non-zero (RigDev::ptOffsetsNonZero)
  const int32_t* levelOrder;
  const int32_t* levelStart;
  const int4* rowRec; // RigDev::ptRowRec (null: walk the CSR)
  int32_t numRowRec;
};
// I: int32_t where the tables stay in global memory or are LDS copies of the tree kernels; int16_t in the one-launch solve,
// whose LDS budget is what decides between three and four workgroups per CU (every index of a fused problem is below 4096)
template <class I>
struct FusedViewT {
  int32_t U, Kp;
  const I* subSize;
  const I* dfsJoint;
  const I* loadedPos;
  int32_t numLoaded;
  const I* colToSolve; // [P] compacted index of a parameter or -1
  const I* unitPos; // [U] DFS position of the joint a unit hangs on
  const I* posUnitStart;
  const I* posUnits;
  const I* solveList;
};
using FusedView = FusedViewT<int32_t>;
using FusedViewS = FusedViewT<int16_t>;

__host__ __device__ __forceinline__ size_t alignUp4(size_t x) {
  return (x + 3) & ~size_t(3);
}
// channel stride of the per-slot tables: >= nsrc and = 16 mod 32, so that the 16 slots x 2 channels a
// 32-lane group reads as MFMA operands (phase G) fall into 32 different banks
__host__ __device__ __forceinline__ int srcStrideFor(int nsrc) {
  int x = (nsrc + 15) & ~15;
  return (x & 31) == 16 ? x : x + 16;
}
constexpr int kSrcCh = 15; // D(7) | A(7) | g share (the tree kernels; the one-launch solve keeps the g share in uy's place: 14)

// ---------------------------------------------------------------------------------------------
// adjoint machinery: sums over a joint's own units, then over its subtree (= a DFS index range)
// ---------------------------------------------------------------------------------------------
// Two steps, both fully parallel: (1) one thread per unit writes the unit's moment contributions
// (NCH floats: kC1 first-order, then kC2 second-order channels) to a scratch array; (2) one thread per
// (loaded joint, channel) adds up the few units of that joint, in ascending unit order.  Joints
// without units are never read by the subtree sums (treeSumT walks loadedPos), so they are skipped.
__device__ __forceinline__ void firstOrderMoments(float* o, F3 p, float yx, float yy, float yz, bool point) {
  // N += p x y (points and directions share the channel: only the sum enters, see jt_times)
  o[3] = p.y * yz - p.z * yy;
  o[4] = p.z * yx - p.x * yz;
  o[5] = p.x * yy - p.y * yx;
  o[0] = point ? yx : 0.f;
  o[1] = point ? yy : 0.f;
  o[2] = point ? yz : 0.f;
  o[6] = point ? p.x * yx + p.y * yy + p.z * yz : 0.f;
}

// NCH channels per unit (kC1 first-order, then kC2Used second-order ones), stored with stride STRIDE (odd)
template <int NCH, int STRIDE, int kT = 256, class FV = FusedView, bool kSecondOnly = false>
__device__ __forceinline__ void gatherOwnSums(const FV& fd, const FusedLds& s, const float* umom, int tid) {
  for (int item = tid; item < fd.numLoaded * NCH; item += kT) {
    const int li = item / NCH, c = item - li * NCH;
    const int k = fd.loadedPos[li];
    float acc = 0.f;
    const int e1 = fd.posUnitStart[k + 1];
    for (int e = fd.posUnitStart[k]; e < e1; ++e) {
      acc += umom[STRIDE * fd.posUnits[e] + c];
    }
    if (kSecondOnly) { // (the channels are the second-order ones only)
      s.own2[kC2 * k + c] = acc;
    } else if (c < kC1) {
      s.own1[kC1 * k + c] = acc;
    } else {
      s.own2[kC2 * k + (c - kC1)] = acc;
    }
  }
}
constexpr int kUmom = 25; // stride of the per-unit moment scratch: 7 + 16 channels, odd

// phase D: first- and second-order own sums from up / uy / us
// kSecondOnly: the second-order channels only (the mixed-precision solve: g = J^T r is formed in double, mmx_mixed.hpp; uy is
// not read)
// (upD / usD: the units' vectors and weights in double, rounded here -- the mixed route keeps no single-precision copies)
template <int kT = 256, class FV = FusedView, bool kSecondOnly = false>
__device__ __forceinline__ void ownSums(const FV& fd, const FusedLds& s, float* umom, int U, int tid, const double* upD = nullptr, const double* usD = nullptr) {
  constexpr int NCH = kSecondOnly ? kC2Used : kC1 + kC2Used;
  for (int u = tid; u < U; u += kT) {
    const F3 p = kSecondOnly ? F3{float(upD[3 * u]), float(upD[3 * u + 1]), float(upD[3 * u + 2])} : F3{s.up[3 * u], s.up[3 * u + 1], s.up[3 * u + 2]};
    const bool point = u < fd.Kp;
    float* o = umom + kUmom * u;
    if (!kSecondOnly) {
      firstOrderMoments(o, p, s.uy[3 * u], s.uy[3 * u + 1], s.uy[3 * u + 2], point);
    }
    const float sg = kSecondOnly ? float(usD[u]) : s.us[u];
    const float s2 = sg * sg;
    float* o2 = kSecondOnly ? o : o + kC1;
#pragma unroll
    for (int c = 0; c < kC2Used; ++c) {
      o2[c] = 0.f;
    }
    const int q = point ? 4 : 10;
    o2[q + 0] = s2 * p.x * p.x;
    o2[q + 1] = s2 * p.x * p.y;
    o2[q + 2] = s2 * p.x * p.z;
    o2[q + 3] = s2 * p.y * p.y;
    o2[q + 4] = s2 * p.y * p.z;
    o2[q + 5] = s2 * p.z * p.z;
    if (point) {
      o2[0] = s2;
      o2[1] = s2 * p.x;
      o2[2] = s2 * p.y;
      o2[3] = s2 * p.z;
    }
  }
  __syncthreads();
  gatherOwnSums<NCH, kUmom, kT, FV, kSecondOnly>(fd, s, umom, tid);
}

constexpr int kFusedTreeUn = 4; // k-steps per trip of the tree sums (measured on BASELINE configs[1]: 1 -> 1.575e6, 4 -> 1.60e6, 8 -> 1.54e6 solves/s)
template <int NC, bool kSubtree, int STRIDE = NC, int UN = kFusedTreeUn, class I = int32_t>
__device__ __forceinline__ void treeSum(const FusedViewT<I>& fd, const float* in, float* out, int J, int wave, int lane, const int32_t* kRange = nullptr) {
  treeSumT<NC, kSubtree, STRIDE, UN, I>(fd.subSize, fd.loadedPos, fd.numLoaded, in, out, J, wave, 4, lane, kRange);
}

// ---------------------------------------------------------------------------------------------
// parameter-space rows (LimitErrorFunctionT on model parameters, ModelParametersErrorFunctionT):
// they need no joint state, so their contributions to the error, to g / H and to the refinement
// are evaluated on the fly from theta -- no extra LDS, nothing stored per row.
// ---------------------------------------------------------------------------------------------
struct ParamCol {
  float g, h;
};

// contributions of the rows to solve column c = parameter p: g_c = sum_l J(l,c) r_l and
// H_cc = sum_l J(l,c)^2.  With a step `d0` the residual is replaced by r - J d0 (refinement).
template <class I = int32_t>
__device__ __forceinline__ ParamCol paramRowsColumn(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    const float* th,
    const float* d0,
    const I* colToSolve,
    int P,
    int b,
    int c,
    int p) {
  ParamCol o{0.f, 0.f};
  if (fd.numLimits > 0 && pb.wLimit > 0.f) {
    const float tWeight = 1e+1f * pb.wLimit;
    const int k1 = fd.limStart[c + 1];
    for (int k = fd.limStart[c]; k < k1; ++k) {
      const LimitRow row = evalLimit(rig, pb.limits[fd.limOf[k]], th, pb.enabledMask, tWeight);
      float coef = 0.f, rr = row.r;
#pragma unroll
      for (int e = 0; e < kLimitEntries; ++e) {
        coef += row.idx[e] == p ? row.coef[e] : 0.f;
        if (d0 != nullptr) {
          const int sc = row.idx[e] >= 0 ? colToSolve[row.idx[e]] : -1;
          rr -= row.coef[e] * (sc >= 0 ? d0[sc] : 0.f); // parameters outside the solve list do not move
        }
      }
      o.g += coef * rr;
      o.h += coef * coef;
    }
  }
  if (pb.hasModel && pb.wModel > 0.f) {
    const float w = pb.mpWeights[size_t(b) * P + p];
    if (w > 0.f) {
      const float sWeight = sqrtf(pb.wModel * 1e-1f);
      const float a = sWeight * w;
      float rr = (w * (th[p] - pb.mpTarget[size_t(b) * P + p])) * sWeight;
      if (d0 != nullptr) {
        rr -= a * d0[c];
      }
      o.g += a * rr;
      o.h += a * a;
    }
  }
  return o;
}

// The same in double for the mixed-precision instantiation (LimitErrorFunctionT<double>, ModelParametersErrorFunctionT<double>):
// kOp = false: g_c = sum_l J(l,c) r_l and H_cc = sum_l J(l,c)^2 of the rows at theta `th`; kOp = true: the rows' part of the
// operator, (J_p^T J_p x)_c for the vector x over the solve columns (g) -- h unused.
struct ParamColD {
  double g, h;
};
template <bool kOp, class I>
__device__ __forceinline__ ParamColD paramRowsColumnD(
    const RigDev& rig, const ProblemDev& pb, const FusedDev& fd, const double* th, const double* x, const I* colToSolve, int P, int b, int c, int p) {
  ParamColD o{0.0, 0.0};
  if (fd.numLimits > 0 && pb.wLimit > 0.f) {
    const double tWeight = double(1e+1f * pb.wLimit); // kLimitWeight * weight_ (a float product in both instantiations)
    const int k1 = fd.limStart[c + 1];
    for (int k = fd.limStart[c]; k < k1; ++k) {
      const LimitRowT<double> row = evalLimit<double>(rig, pb.limits[fd.limOf[k]], th, pb.enabledMask, tWeight);
      double coef = 0.0, jx = 0.0;
#pragma unroll
      for (int e = 0; e < kLimitEntries; ++e) {
        coef += row.idx[e] == p ? row.coef[e] : 0.0;
        if (kOp) {
          const int sc = row.idx[e] >= 0 ? colToSolve[row.idx[e]] : -1;
          jx += row.coef[e] * (sc >= 0 ? x[sc] : 0.0); // parameters outside the solve list do not move
        }
      }
      o.g += coef * (kOp ? jx : row.r);
      o.h += coef * coef;
    }
  }
  if (pb.hasModel && pb.wModel > 0.f) {
    const double w = double(pb.mpWeights[size_t(b) * P + p]);
    if (w > 0.0) {
      const double sWeight = double(sqrtf(pb.wModel * 1e-1f)); // sWeight: a float in both instantiations (model_parameters_error_function.cpp:109)
      const double a = sWeight * w;
      o.g += a * (kOp ? a * x[c] : (w * (th[p] - double(pb.mpTarget[size_t(b) * P + p]))) * sWeight);
      o.h += a * a;
    }
  }
  return o;
}
// this thread's share of the parameter-space blocks' error at `th` in double (paramRowsError of mmx_device.hpp with T = double)
template <bool kJacobianRows>
__device__ __forceinline__ double paramRowsErrorD(const RigDev& rig, const ProblemDev& pb, int P, const double* th, int b, int tid) {
  double e = 0.0;
  if (pb.NL > 0 && pb.wLimit > 0.f) {
    const double tWeight = double(1e+1f * pb.wLimit);
    for (int l = tid; l < pb.NL; l += 256) {
      e += evalLimit<double>(rig, pb.limits[l], th, pb.enabledMask, tWeight).err;
    }
  }
  if (pb.hasModel && pb.wModel > 0.f) {
    const float* tp = pb.mpTarget + size_t(b) * P;
    const float* tw = pb.mpWeights + size_t(b) * P;
    double em = 0.0;
    for (int i = tid; i < P; i += 256) {
      if (pb.enabledMask[i] != 0) {
        const double w = double(tw[i]);
        if (!kJacobianRows || w > 0.0) {
          const double pd = w * (th[i] - double(tp[i]));
          em += pd * pd;
        }
      }
    }
    e += em * double(pb.wModel) * 1e-1;
  }
  return e;
}

// y = A x for a CSR matrix whose tables live in GLOBAL memory (the wide kernels; the fused solve keeps them in LDS):
// four rows per trip with their row bounds, then their first entries, requested together -- a row's walk is a chain
// of dependent L2 round trips otherwise.  Same products in the same order as the plain walk.
template <int kT = 256, typename Gather, typename Store> // kT: threads of the workgroup
__device__ __forceinline__ void csrRowsPrefetched(const int32_t* outer, const int32_t* inner, const float* value, int R, int tid, Gather x, Store out) {
  for (int r0 = tid; r0 < R; r0 += 4 * kT) {
    int ka[4], kb[4], in0[4];
    float v0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = min(r0 + kT * i, R - 1);
      ka[i] = outer[r], kb[i] = outer[r + 1];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool has = kb[i] > ka[i];
      in0[i] = has ? inner[ka[i]] : 0, v0[i] = has ? value[ka[i]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + kT * i;
      if (r < R) {
        decltype(x(0)) acc = 0; // (float; double in the mixed-precision solve)
        if (kb[i] > ka[i]) {
          acc += v0[i] * x(in0[i]);
          for (int k = ka[i] + 1; k < kb[i]; ++k) {
            acc += value[k] * x(inner[k]);
          }
        }
        out(r, acc);
      }
    }
  }
}

// y = A x from the records of A's non-empty rows (RigDev::ptRowRec): one 16-byte load per thread, then a row's first product
// from the record itself; the rare further entries of a row from the CSR arrays.  out(r, value) is called for EVERY row r < R
// exactly once (the empty ones with 0) by the thread that owns the record before it -- no zero-fill pass, no barrier.
template <int kT = 256, typename Gather, typename Store>
__device__ __forceinline__ void csrRowsFromRecords(const int4* rec, int numRec, const int32_t* inner, const float* value, int R, int tid, Gather x, Store out) {
  for (int t = tid; t < numRec; t += kT) {
    const int4 q = rec[t];
    const int row = q.x & 0xffff, span = int(uint32_t(q.x) >> 16), in0 = q.y & 0xffff, cnt = int(uint32_t(q.y) >> 16); // (unsigned fields: mmx_rig_create builds no records when one would not fit 16 bits)
    auto acc = __int_as_float(q.z) * x(in0); // (float; double in the mixed-precision solve)
    for (int k = q.w + 1; k < q.w + cnt; ++k) {
      acc += value[k] * x(inner[k]);
    }
    out(row, acc);
    for (int r = row + 1; r < row + span; ++r) {
      out(r, decltype(acc)(0));
    }
    if (t == 0) {
      for (int r = 0; r < row; ++r) {
        out(r, decltype(acc)(0));
      }
    }
  }
}

// Forward kinematics of the whole skeleton from the parameters in `th` into s.js: local transforms
// of all joints at once (parameter_transform.cpp:110-124, joint_state.cpp:44-62), world transforms
// by pointer jumping (skeleton_state.cpp:100-121 re-associated), optionally the rotation axes.
// Ends with a barrier.  Clobbers alt / jlA / jlB (assembly scratch = the Cholesky region).
template <bool kGlobalTables = false, int kT = 256> // the rig's CSR tables are read from global memory (prefetching walk); kT threads
__device__ __forceinline__ void
blockFk(const RigView& rig, const FusedLds& s, const float* th, int tid, bool withAxes, long long* clk = nullptr, long long* clkLast = nullptr) {
  auto stamp = [&](int slot) { // profiling aid (MMX_PHASE_CLOCKS): sub-phases of FK
    if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
      const long long now = clock64();
      clk[slot] += now - *clkLast;
      *clkLast = now;
    }
  };
  // joint parameters = transform * theta + offsets, one transform ROW per thread (parameter_transform.cpp:
  // 110-124; the same products in the same order as a per-joint walk): 7 J independent short CSR walks
  // instead of seven dependent ones per joint.  They land in the refinement scratch (jd), which is dead
  // whenever FK runs.
  if (kGlobalTables && rig.rowRec != nullptr) {
    csrRowsFromRecords<kT>(
        rig.rowRec, rig.numRowRec, rig.ptInner, rig.ptValue, rig.R, tid, [&](int c) { return th[c]; }, [&](int r, float acc) { s.jd[r] = acc + (rig.hasOffsets ? rig.ptOffsets[r] : 0.f); });
  } else if (kGlobalTables) {
    csrRowsPrefetched<kT>(
        rig.ptOuter, rig.ptInner, rig.ptValue, rig.R, tid, [&](int c) { return th[c]; }, [&](int r, float acc) { s.jd[r] = acc + (rig.hasOffsets ? rig.ptOffsets[r] : 0.f); });
  } else {
    for (int r = tid; r < rig.R; r += kT) {
      const float off = rig.hasOffsets ? rig.ptOffsets[r] : 0.f; // (a global load on the phase's critical path when it is taken)
      float acc = 0.f;
      const int k1 = rig.ptOuter[r + 1];
      for (int k = rig.ptOuter[r]; k < k1; ++k) {
        acc += rig.ptValue[k] * th[rig.ptInner[k]];
      }
      s.jd[r] = acc + off;
    }
  }
  // the joint's constants are requested before the barrier, so that their L2 round trip overlaps it
  float pre[4] = {0.f, 0.f, 0.f, 1.f}, off3[3] = {0.f, 0.f, 0.f};
  if (tid < rig.J) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      pre[d] = rig.preRot[4 * tid + d];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      off3[d] = rig.offset[3 * tid + d];
    }
  }
  __syncthreads();
  stamp(26);
  const int Jp = fkPad(rig.J);
  for (int j = tid; j < rig.J; j += kT) {
    float* slot = s.js + kJs * j;
    float loc[8];
    if (j == tid) {
      fkLocalFromParams(s.jd + 7 * j, pre, off3, loc, slot + 8);
    } else {
      fkLocalFromParams(s.jd + 7 * j, rig.preRot + 4 * j, rig.offset + 3 * j, loc, slot + 8);
    }
    if (rig.jumpRounds == 0) { // every joint is a root
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        slot[c] = loc[c];
      }
    } else {
      fkStoreLocalD(s.fkA, Jp, j, loc, rig.parent[j] + 1);
    }
  }
  __syncthreads();
  stamp(24);
  fkJumpRoundsD(s.js, s.fkA, s.fkB, rig.J, rig.jumpRounds, tid, kT);
  stamp(25);
  if (withAxes) {
    for (int j = tid; j < rig.J; j += kT) {
      fkAxesInPlaceP(rig, j, rig.parent[j], s.js);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// The double passes of the mixed-precision instantiation (kMix; mmx_mixed.hpp has the primitives and the rationale).
// ---------------------------------------------------------------------------------------------
// this thread's transform-row record when the rig has at most one record per thread (a value and a flag, not a nullable pointer:
// `has ? &rec : nullptr` keeps the record in scratch)
struct MixRowRec {
  int4 q{0, 0, 0, 0};
  bool has = false;
};
// y = transform * x in double: one transform row per thread from the rig's row records, else the prefetching CSR walk
// (the tables in global memory either way); out(r, value) for every row r < R
// regRec: this thread's row record, loaded once per solve (rig.numRowRec <= 256: a record per thread; the walk's L2 round trip
// -- twice per operator application and once per forward pass -- then only remains for rows with more than one entry)
template <typename Gather, typename Store>
__device__ __forceinline__ void mixTransformRows(const RigView& rig, int tid, Gather x, Store out, const MixRowRec& regRec = MixRowRec{}) {
  if (regRec.has) {
    if (tid < rig.numRowRec) {
      const int4 q = regRec.q;
      const int row = q.x & 0xffff, span = int(uint32_t(q.x) >> 16), in0 = q.y & 0xffff, cnt = int(uint32_t(q.y) >> 16);
      auto acc = __int_as_float(q.z) * x(in0);
      for (int k = q.w + 1; k < q.w + cnt; ++k) {
        acc += rig.ptValue[k] * x(rig.ptInner[k]);
      }
      out(row, acc);
      for (int r = row + 1; r < row + span; ++r) {
        out(r, decltype(acc)(0));
      }
      if (tid == 0) {
        for (int r = 0; r < row; ++r) {
          out(r, decltype(acc)(0));
        }
      }
    }
  } else if (rig.rowRec != nullptr) {
    csrRowsFromRecords<256>(rig.rowRec, rig.numRowRec, rig.ptInner, rig.ptValue, rig.R, tid, x, out);
  } else {
    csrRowsPrefetched<256>(rig.ptOuter, rig.ptInner, rig.ptValue, rig.R, tid, x, out);
  }
}

// Forward kinematics of `th` (double) into m.js: joint parameters (m.X), local transforms of all joints at once, then the
// world transforms ONE TREE LEVEL PER BARRIER in the reference's own association order, world_j = world_parent o local_j
// (skeleton_state.cpp:100-121) -- NOT the pointer jumping of the single-precision kernels: the joints' pre-rotations are
// float quaternions (|q|^2 = 1 +- 6e-8), and Eigen's q * v (transform.h:124-129) is a homomorphism only for unit quaternions, so
// a re-associated product of the same transforms differs from the sequential one by ~1e-7 -- invisible in single precision,
// the largest term left in double (measured: BASELINE configs[1] sat at 1.5e-7 ... 5e-7 of the oracle's double run whatever
// the CG tolerance, the 24-joint chain with identity pre-rotations at 2.5e-8 = the rounding of the float result).
// myLevel: the tree level of joint `tid` (one joint per thread, J <= 256); optionally the rotation axes.  Ends with a barrier.
__device__ __forceinline__ void blockFkD(const RigView& rig, const MixLds& m, const double* th, int tid, int myLevel, bool withAxes, const MixRowRec& regRec = MixRowRec{}) {
  mixTransformRows(
      rig, tid, [&](int c) { return th[c]; }, [&](int r, double acc) { m.X[r] = acc + (rig.hasOffsets ? double(rig.ptOffsets[r]) : 0.0); }, regRec);
  __syncthreads();
  double loc[8];
  const int j = tid;
  double* slot = m.js + kJsD * (j < rig.J ? j : 0);
  if (j < rig.J) {
    fkLocalFromParamsD(m.X + 7 * j, rig.preRot + 4 * j, rig.offset + 3 * j, loc, slot + 8);
    if (myLevel == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        slot[c] = loc[c];
      }
    }
  }
  __syncthreads();
  for (int lvl = 1; lvl < rig.numLevels; ++lvl) {
    if (j < rig.J && myLevel == lvl) {
      const double* w = m.js + kJsD * rig.parent[j];
      const DQ qp{w[3], w[4], w[5], w[6]};
      const D3 t = D3{w[0], w[1], w[2]} + dqrot(qp, w[7] * D3{loc[0], loc[1], loc[2]}); // transform.h:124-129
      const DQ q = dqmul(qp, DQ{loc[3], loc[4], loc[5], loc[6]});
      slot[0] = t.x, slot[1] = t.y, slot[2] = t.z, slot[3] = q.x, slot[4] = q.y, slot[5] = q.z, slot[6] = q.w, slot[7] = w[7] * loc[7];
    }
    __syncthreads();
  }
  if (withAxes) {
    if (j < rig.J) {
      fkAxesInPlaceD(rig.preRot + 4 * j, j, rig.parent[j], m.js);
    }
    __syncthreads();
  }
}

// phase C in double: units from the joint states in m.js (stored: m.up / m.uf / m.us); returns this thread's share of the error
template <bool kStore>
__device__ __forceinline__ double mixUnits(const ProblemDev& pb, const FusedLds& s, const MixLds& m, int b, int U, int tid) {
  double e = 0.0;
  for (int u = tid; u < U; u += 256) {
    const UnitD un = evalUnitD(pb, loadUnitInput(pb, b, u), m.js, u);
    if (kStore) {
      m.up[3 * u] = un.v.x, m.up[3 * u + 1] = un.v.y, m.up[3 * u + 2] = un.v.z;
      m.uf[3 * u] = un.f.x, m.uf[3 * u + 1] = un.f.y, m.uf[3 * u + 2] = un.f.z;
      m.us[u] = un.sigma;
    }
    e += un.werr;
  }
  return e;
}

// SkeletonSolverFunctionT<double>::getError of `th` (skeleton_solver_function.cpp:64-83: rounded through float, :82); every
// thread returns the same value.  kStore: also leaves what phases A-C of an iteration would leave (see blockError).
template <bool kStore>
__device__ __forceinline__ double blockErrorD(
    const RigView& rig, const ProblemDev& pb, const FusedLds& s, const MixLds& m, const double* th, int b, int U, int tid, int myLevel, double* unrounded = nullptr, const MixRowRec& regRec = MixRowRec{},
    const RigDev* rowsRig = nullptr, bool hasRows = false) { // hasRows: the problem has parameter-space rows (their getError share; rowsRig = the rig descriptor.
  // A separate flag, not a null pointer: `cond ? &rig : nullptr` at the call keeps the whole by-value descriptor in scratch)
  blockFkD(rig, m, th, tid, myLevel, kStore, regRec);
  double e = mixUnits<kStore>(pb, s, m, b, U, tid);
  if (hasRows) {
    e += paramRowsErrorD<false>(*rowsRig, pb, rig.P, th, b, tid);
  }
  const double tot = blockSumD(s.red, e, tid);
  if (unrounded != nullptr) {
    *unrounded = tot;
  }
  return double(float(tot));
}

// Adjoint pass in double: out[c] = (J^T y)_c for the solve columns c < n (0 for the pad columns), y_u = yOf(u, k) the adjoint
// input of unit u (k: DFS position of its joint).  Own sums per loaded joint (m.X) -> subtree sums (m.Y; the loaded positions
// inside a subtree are the index range lo..hi of the ascending loadedPos) -> per-slot gradients (m.X) -> columns.
// (Measured and not kept, round 6: the subtree sums walking the position-sorted UNIT list directly -- a root joint's sum is then
// a chain of 64 dependent LDS round trips: the operator went from 20 k to 35 k cycles.)
// Clobbers m.X and m.Y; yOf may read m.Y (the tangent pass's prefixes: consumed before the first barrier).  Ends WITHOUT a barrier.
template <class FV, typename YFn>
__device__ __forceinline__ void mixAdjoint(const FV& fd, const FusedLds& s, const MixLds& m, int J, int NP, int n, int nsrc, int tid, YFn yOf, double* out) {
  double *sub, *slot; // where the subtree sums / the per-slot gradients end up
  if (size_t(7 * fd.U) <= mixXDoubles(J, nsrc) && size_t(nsrc) <= mixYDoubles(J, 0)) {
    // one thread per UNIT writes its seven moments (the unit evaluation is the arithmetic of this pass: all units at once instead of
    // a loaded joint's units one after the other), then one thread per (loaded joint, channel) adds the joint's few units
    for (int u = tid; u < fd.U; u += 256) {
      const D3 pu{m.up[3 * u], m.up[3 * u + 1], m.up[3 * u + 2]};
      const D3 y = yOf(u, fd.unitPos[u], pu);
      const D3 Nv = dcross(pu, y); // points and directions share the channel (jt_times)
      const bool point = u < fd.Kp;
      double* o = m.X + 7 * u;
      o[0] = point ? y.x : 0.0, o[1] = point ? y.y : 0.0, o[2] = point ? y.z : 0.0;
      o[3] = Nv.x, o[4] = Nv.y, o[5] = Nv.z;
      o[6] = point ? ddot(pu, y) : 0.0;
    }
    __syncthreads();
    for (int item = tid; item < 7 * fd.numLoaded; item += 256) {
      const int li = item / 7, c = item - 7 * li;
      const int k = fd.loadedPos[li];
      double acc = 0.0;
      const int e1 = fd.posUnitStart[k + 1];
      for (int e = fd.posUnitStart[k]; e < e1; ++e) {
        acc += m.X[7 * fd.posUnits[e] + c];
      }
      m.Y[item] = acc;
    }
    __syncthreads();
    for (int item = tid; item < 7 * J; item += 256) {
      const int k = item / 7, c = item - 7 * k;
      double acc = 0.0;
      const int l1 = m.hi[k];
      for (int li = m.lo[k]; li < l1; ++li) {
        acc += m.Y[7 * li + c];
      }
      m.X[item] = acc;
    }
    sub = m.X, slot = m.Y;
  } else {
    for (int li = tid; li < fd.numLoaded; li += 256) {
      const int k = fd.loadedPos[li];
      D3 Fv{0.0, 0.0, 0.0}, Nv{0.0, 0.0, 0.0};
      double Dd = 0.0;
      const int e1 = fd.posUnitStart[k + 1];
      for (int e = fd.posUnitStart[k]; e < e1; ++e) {
        const int u = fd.posUnits[e];
        const D3 pu{m.up[3 * u], m.up[3 * u + 1], m.up[3 * u + 2]};
        const D3 y = yOf(u, k, pu);
        Nv = Nv + dcross(pu, y);
        if (u < fd.Kp) {
          Fv = Fv + y;
          Dd += ddot(pu, y);
        }
      }
      double* o = m.X + 7 * li;
      o[0] = Fv.x, o[1] = Fv.y, o[2] = Fv.z, o[3] = Nv.x, o[4] = Nv.y, o[5] = Nv.z, o[6] = Dd;
    }
    __syncthreads();
    for (int item = tid; item < 7 * J; item += 256) {
      const int k = item / 7, c = item - 7 * k;
      double acc = 0.0;
      const int l1 = m.hi[k];
      for (int li = m.lo[k]; li < l1; ++li) {
        acc += m.X[7 * li + c];
      }
      m.Y[7 * k + c] = acc;
    }
    sub = m.Y, slot = m.X;
  }
  __syncthreads();
  for (int e = tid; e < nsrc; e += 256) {
    const int info = s.mInfo[e];
    slot[e] = double(s.mW[e]) * sourceGradientD(info & 0xfff, (info >> 12) & 7, (info >> 16) - 1, m.js, sub + 7 * (s.mTin[e] & 0xffff));
  }
  __syncthreads();
  for (int c = tid; c < NP; c += 256) {
    double a = 0.0;
    if (c < n) {
      a = slot[c]; // the primary slot, then the extras
      const int e1 = NP + s.mStart[c + 1];
      for (int e = NP + s.mStart[c]; e < e1; ++e) {
        a += slot[e];
      }
    }
    out[c] = a;
  }
}

// Tangent pass in double for the step x (solve columns; xOf(model parameter) -> its entry or 0): joint-parameter step (m.X),
// per joint C = T - Om x t - ln2 sd t, W = Om, S = sd summed over the ancestor chain by pointer jumping -> m.Y[kTanD k ..]
// by DFS position k.  J <= 256.  Ends with a barrier.
template <class FV, typename XFn>
__device__ __forceinline__ void mixTangent(const RigView& rig, const FV& fd, const MixLds& m, const int16_t* parentPos, int J, int tid, XFn xOf, const MixRowRec& regRec = MixRowRec{}) {
  mixTransformRows(rig, tid, xOf, [&](int r, double a) { m.X[r] = a; }, regRec);
  __syncthreads();
  double acc[7];
  int target = -1;
  if (tid < J) {
    const int q = fd.dfsJoint[tid];
    const double* ja = m.js + kJsD * q;
    const double* d = m.X + 7 * q;
    const D3 ta{ja[0], ja[1], ja[2]};
    D3 Tv{0.0, 0.0, 0.0};
    if (d[0] != 0.0 || d[1] != 0.0 || d[2] != 0.0) {
      const int par = rig.parent[q];
      Tv = d[0] * transAxisColD(m.js, par, 0) + d[1] * transAxisColD(m.js, par, 1) + d[2] * transAxisColD(m.js, par, 2);
    }
    const D3 Om = d[3] * D3{ja[8], ja[9], ja[10]} + d[4] * D3{ja[11], ja[12], ja[13]} + d[5] * D3{ja[14], ja[15], ja[16]};
    const D3 C = Tv - dcross(Om, ta) - (kLn2D * d[6]) * ta;
    acc[0] = C.x, acc[1] = C.y, acc[2] = C.z, acc[3] = Om.x, acc[4] = Om.y, acc[5] = Om.z, acc[6] = d[6];
    target = parentPos[tid];
    double* o = m.Y + kTanD * tid;
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      o[c] = acc[c];
    }
    reinterpret_cast<int*>(o + 7)[0] = target;
  }
  __syncthreads();
  for (int r = 0; r < rig.jumpRounds; ++r) {
    int next = -1;
    if (target >= 0) {
      const double* t = m.Y + kTanD * target;
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        acc[c] += t[c];
      }
      next = reinterpret_cast<const int*>(t + 7)[0];
    }
    __syncthreads();
    if (target >= 0) {
      double* o = m.Y + kTanD * tid;
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        o[c] = acc[c];
      }
      reinterpret_cast<int*>(o + 7)[0] = next;
    }
    target = next;
    __syncthreads();
  }
}

// q = (J^T S^2 J + mu I) p in double (tangent pass down, adjoint pass up); p, q: [NP] over the solve columns.  Ends with a barrier.
template <class FV>
__device__ __forceinline__ void mixApply(
    const RigView& rig, const FV& fd, const FusedLds& s, const MixLds& m, const int16_t* parentPos, int J, int NP, int n, int nsrc, double mu, const double* p, double* q, int tid, const MixRowRec& regRec = MixRowRec{}) {
  mixTangent(
      rig, fd, m, parentPos, J, tid,
      [&](int c) {
        const int cs = fd.colToSolve[c];
        return cs >= 0 ? p[cs] : 0.0;
      },
      regRec);
  mixAdjoint(
      fd, s, m, J, NP, n, nsrc, tid,
      [&](int u, int k, const D3& pu) {
        const double* pre = m.Y + kTanD * k;
        D3 v = dcross(D3{pre[3], pre[4], pre[5]}, pu);
        if (u < fd.Kp) {
          v = D3{pre[0], pre[1], pre[2]} + v + (kLn2D * pre[6]) * pu;
        }
        const double sg = m.us[u];
        return (sg * sg) * v;
      },
      q);
  for (int c = tid; c < n; c += 256) { // (the thread that wrote q[c])
    q[c] += mu * p[c];
  }
  __syncthreads();
}

constexpr int kGenEv = 29; // words per constraint record (odd stride)

// Which instantiations of the one-launch solve are set up for FOUR workgroups per CU (128 registers, the transform's CSR walked
// from global memory, the register-lean forms of the triangular solves / tile products): up to six blocks, reference rows, no
// trust region -- the per-rule ones and the generic rule (line searches).
template <int NB, bool kTR, bool kGen, int kRule>
struct FusedFour {
  static constexpr bool value = NB <= 6 && !kGen && !kTR;
};

// Sizes of the one-launch solve's lifetime-shared LDS areas (fusedSolveKernel's carve and fusedLdsBytes agree through this)
struct FusedLayout {
  size_t uyFloats; // uy (C-D) | slots' gradient shares (E-F) | partial cells (G) | rho, invDiag (end of G .. K)
  size_t auxOff; // where the latter three start inside it: 0, or behind uy when uy must survive the factorisation (separateUy)
  size_t cellOff; // where the partial cells start (counted from auxOff)
  size_t t9; // aligned kTan J: tanOwn at 0, jd and tanPre at t9, dfull at 2 t9 of the arena
  size_t arenaFloats; // srcT (E-G, 14 channels) | the refinement's buffers + dfull (J-K)
  size_t umomFloats; // the unit moments' place, also the first-order subtree sums' (D-E)
  size_t regionFloats; // assembly scratch (A-G) | the factor's tiles
  size_t genFloats; // kGen: records, residual rows, w, J_g
  size_t total;
};
// separateUy: the trust region re-runs phases D-G for every value of its damping WITHOUT re-evaluating the units (phase C), so
// its uy must outlive rho / invDiag
// csrFloats: LDS copy of the parameter transform's CSR (row pointer, columns, values as 32-bit words) in the instantiations
// that run three or fewer workgroups per CU -- they have the room, and an L2 walk costs them 4 %; 0: read from global memory
// mix: the mixed-precision instantiation (kMix; mmx_mixed.hpp): theta, joint states, units, g and the CG vectors in double
// INSTEAD of their single-precision twins, the double scratch X | Y in the arena
__host__ __device__ inline FusedLayout fusedLayout(int NB, int J, int P, int U, int nsrc, int n, int numCells, bool cellsBehindRho, int GT, int genRows, bool separateUy = false, size_t csrFloats = 0, bool mix = false) {
  auto a4 = [](size_t x) { return (x + 3) & ~size_t(3); };
  auto h4 = [&](size_t x) { return a4((x + 1) / 2); }; // 16-bit entries
  auto mx = [](size_t x, size_t y) { return x > y ? x : y; };
  const size_t T = size_t(NB) * (NB + 1) / 2, NP = 16 * size_t(NB);
  FusedLayout l;
  l.cellOff = cellsBehindRho ? 2 * NP : 0;
  const size_t aux = a4(mx(2 * NP, mx(size_t(nsrc), l.cellOff + size_t(numCells))));
  l.auxOff = separateUy ? a4(3 * size_t(U)) : 0;
  l.uyFloats = separateUy ? l.auxOff + aux : mx(a4(3 * size_t(U)), aux);
  l.t9 = a4(size_t(kTan) * J);
  l.arenaFloats = mx(size_t(kSrcCh - 1) * size_t(srcStrideFor(nsrc)), 2 * l.t9 + a4(P));
  if (mix) {
    l.arenaFloats = mx(size_t(kSrcCh - 1) * size_t(srcStrideFor(nsrc)), a4(2 * (mixXDoubles(J, nsrc) + mixYDoubles(J, P))));
  }
  l.umomFloats = mx(a4(size_t(kUmom) * U), a4(size_t(kC1) * J));
  const size_t scratch = fkBufFloats(J) + 2 * a4(size_t(kC2) * J) + l.umomFloats;
  l.regionFloats = mx(scratch, T * 256);
  const size_t rowsGp = a4(size_t(genRows));
  l.genFloats = GT > 0 ? a4(size_t(kGenEv) * GT) + 2 * a4(rowsGp) + a4(rowsGp * size_t(srcStrideFor(int(NP)))) : 0;
  const size_t meta = h4(NP + 1) + 3 * a4(nsrc) + a4(J) + 4 * h4(J) + h4(size_t(J) + 1) + 2 * h4(U) + h4(n) + h4(P);
  size_t fixed = a4(P) + a4(size_t(kJs) * J) + 2 * a4(3 * size_t(U)) + a4(U) + 2 * a4(NP) + l.uyFloats + 16 + 8;
  if (mix) { // uy's place | red, flags | the double arrays | lo, hi
    fixed = l.uyFloats + 40 + 8 + a4(2 * mixPersistentDoubles(J, P, U, int(NP))) + 2 * h4(J);
  }
  l.total = meta + fixed + l.genFloats + l.arenaFloats + l.regionFloats + csrFloats;
  return l;
}

// error of the further joint error functions + ellipsoid limits at the joint states in js (this thread's share)
__device__ __forceinline__ double generalRowsError(const ProblemDev& pb, const float* js, int b, int tid) {
  double e = 0.0;
  for (int g = tid; g < pb.G; g += 256) {
    const JointBlockDev k = jointBlockOf(pb, b, pb.genBlock[g]);
    e += double(evalJointConstraint(k, js, pb.genJoint[g], size_t(b) * size_t(k.count) + size_t(g - k.first)).werr);
  }
  if (pb.wLimit > 0.f) {
    const float tWeightE = 1e+1f * pb.wLimit;
    for (int q = tid; q < pb.NE; q += 256) {
      e += double(evalEllipsoid(pb.ellipsoids[q], js, tWeightE).werr);
    }
  }
  return e;
}

// The further joint error functions and the ellipsoid limits at the joint states in js: per constraint a record
// (v_p, v_n, sigma df/dv_p, sigma df/dv_n, DFS position, first row, flags, stop position) in gEv, the residual rows in
// gRes (rows counted from the first row after the 3 U position / orientation rows); returns this thread's error share.
__device__ __forceinline__ double generalRowsEvaluate(const ProblemDev& pb, const float* js, int b, int U, int tid, float* gEv, float* gRes) {
  double e = 0.0;
  int* evi = reinterpret_cast<int*>(gEv);
  for (int g = tid; g < pb.G; g += 256) {
    const JointBlockDev k = jointBlockOf(pb, b, pb.genBlock[g]);
    const int i = g - k.first;
    const JointEval o = evalJointConstraint(k, js, pb.genJoint[g], size_t(b) * size_t(k.count) + size_t(i));
    const int row = k.rowStart + o.nrows * i - 3 * U;
    e += double(o.werr);
    for (int q = 0; q < o.nrows; ++q) {
      gRes[row + q] = o.sigma * o.f[q];
    }
    const float sg = fabsf(o.sigma) <= 1e-9f ? 0.f : o.sigma; // early termination (joint_error_function-inl.h:216): the rows stay zero
    float* w = gEv + kGenEv * g;
    w[0] = o.vp.x, w[1] = o.vp.y, w[2] = o.vp.z;
    w[3] = o.vn.x, w[4] = o.vn.y, w[5] = o.vn.z;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      w[6 + q] = sg * o.dp[q];
      w[15 + q] = sg * o.dn[q];
    }
    evi[kGenEv * g + 24] = pb.genTin[g];
    evi[kGenEv * g + 25] = row;
    evi[kGenEv * g + 26] = o.nrows | (o.hasPoint ? 16 : 0) | (o.hasDir ? 32 : 0);
    evi[kGenEv * g + 27] = -1;
  }
  const float tWeightE = 1e+1f * pb.wLimit;
  for (int q = tid; q < pb.NE; q += 256) { // LimitType::Ellipsoid (limit_error_function.cpp:702-790), see jointBlocksKernel
    const EllipsoidDev ct = pb.ellipsoids[q];
    const int row = pb.rowsJoint - 3 * pb.NE + 3 * q - 3 * U, g = pb.G + q;
    EllipsoidEval o = evalEllipsoid(ct, js, tWeightE);
    if (!(pb.wLimit > 0.f)) {
      o.jwgt = o.werr = 0.f;
    }
    e += double(o.werr);
    gRes[row] = o.diff.x * o.jwgt, gRes[row + 1] = o.diff.y * o.jwgt, gRes[row + 2] = o.diff.z * o.jwgt;
    float* w = gEv + kGenEv * g;
    w[0] = o.position.x, w[1] = o.position.y, w[2] = o.position.z;
    w[3] = w[4] = w[5] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      w[6 + k] = (k == 0 || k == 4 || k == 8) ? o.jwgt : 0.f;
      w[15 + k] = 0.f;
    }
    evi[kGenEv * g + 24] = ct.tinParent;
    evi[kGenEv * g + 25] = row;
    evi[kGenEv * g + 26] = 3 | 16;
    evi[kGenEv * g + 27] = ct.tinStop;
  }
  return e;
}

// J_g: entry (row of constraint g, solve column c) gathered from the column's source slots -- the walk of
// joint_error_function-inl.h:228-294 turned around as in jointBlocksKernel (constraints fastest).  slot(e): the slot's
// DFS interval, joint, dof, parent and weight; extras(c, e0, e1): the extra slots of column c besides its primary slot c.
struct GenSlot {
  int tin, tout, joint, dof, parent;
  float weight;
};
template <typename SlotFn, typename ExtrasFn>
__device__ __forceinline__ void generalRowsGather(const float* js, const float* gEv, float* gJ, int gst, int GT, int n, int tid, SlotFn slot, ExtrasFn extras) {
  const int* evi = reinterpret_cast<const int*>(gEv);
  for (int item = tid; item < GT * n; item += 256) {
    const int c = item / GT, g = item - c * GT;
    const float* w = gEv + kGenEv * g;
    const int tin = evi[kGenEv * g + 24], row = evi[kGenEv * g + 25], fl = evi[kGenEv * g + 26], tinStop = evi[kGenEv * g + 27];
    const bool hasPoint = (fl & 16) != 0, hasDir = (fl & 32) != 0;
    const F3 vp{w[0], w[1], w[2]}, vn{w[3], w[4], w[5]};
    float acc[3] = {0.f, 0.f, 0.f};
    auto addSlot = [&](int e) {
      const GenSlot sl = slot(e);
      if (!(sl.tin <= tin && tin < sl.tout)) {
        return; // the slot's joint is not an ancestor of the constraint's joint
      }
      if (tinStop >= 0 && sl.tin <= tinStop && tinStop < sl.tout) {
        return; // ellipsoid limit: the walk stopped before this joint
      }
      const float* a = js + kJs * sl.joint;
      F3 gp{0.f, 0.f, 0.f}, gn{0.f, 0.f, 0.f};
      if (sl.dof >= 3 && sl.dof < 6) {
        const float* ax = a + 8 + 3 * (sl.dof - 3);
        const F3 axis{ax[0], ax[1], ax[2]};
        if (hasPoint) {
          gp = cross(axis, vp - F3{a[0], a[1], a[2]});
        }
        if (hasDir) {
          gn = cross(axis, vn);
        }
      } else if (hasPoint) {
        gp = sl.dof < 3 ? transAxisCol(js, sl.parent, sl.dof) : kLn2 * (vp - F3{a[0], a[1], a[2]});
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float jc = (w[6 + 3 * q] * gp.x + w[7 + 3 * q] * gp.y + w[8 + 3 * q] * gp.z) +
            (w[15 + 3 * q] * gn.x + w[16 + 3 * q] * gn.y + w[17 + 3 * q] * gn.z);
        acc[q] += jc * sl.weight;
      }
    };
    addSlot(c);
    int e0, e1;
    extras(c, e0, e1);
    for (int e = e0; e < e1; ++e) {
      addSlot(e);
    }
    gJ[row * gst + c] = acc[0];
    if ((fl & 15) == 3) {
      gJ[(row + 1) * gst + c] = acc[1];
      gJ[(row + 2) * gst + c] = acc[2];
    }
  }
}

// SkeletonSolverFunctionT::getError (skeleton_solver_function.cpp:64-83) of the parameters in
// `th`: FK without derivatives + sum of w * |f|^2, rounded through float like the reference (:82).
// Every thread returns the same value.  Clobbers the FK scratch / js / red.
// kStore: the evaluation also leaves everything phases A-C of an iteration would leave for `th` (rotation axes, the
// units' vectors / residuals / weights) and reports the unrounded sum -- when the trial is accepted, the next iteration
// starts from it instead of repeating forward kinematics and the unit evaluation.
template <bool kGen = false, bool kStore = false, bool kGlobalCsr = true, class FV = FusedViewS>
__device__ __forceinline__ double blockError(
    const RigDev& rigDev,
    const RigView& rig,
    const ProblemDev& pb,
    const FV& fd,
    const FusedLds& s,
    const float* th,
    int b,
    int tid,
    double* unrounded = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  blockFk<kGlobalCsr>(rig, s, th, tid, kStore); // (the transform's CSR: from global memory where the instantiation keeps no LDS copy)
  double e = 0.0;
  for (int u = tid; u < fd.U; u += 256) {
    const Unit un = evalUnit(pb, s.js, b, u);
    if (kStore) {
      s.up[3 * u] = un.v.x, s.up[3 * u + 1] = un.v.y, s.up[3 * u + 2] = un.v.z;
      const float rx = un.sigma * un.f.x, ry = un.sigma * un.f.y, rz = un.sigma * un.f.z;
      s.ur[3 * u] = rx, s.ur[3 * u + 1] = ry, s.ur[3 * u + 2] = rz;
      s.uy[3 * u] = un.sigma * rx, s.uy[3 * u + 1] = un.sigma * ry, s.uy[3 * u + 2] = un.sigma * rz;
      s.us[u] = un.sigma;
    }
    e += double(un.werr);
  }
  if (pb.M > pb.rowsJoint) {
    e += paramRowsError<false>(rigDev, pb, rig.P, th, b, tid);
  }
  if (kGen) {
    e += generalRowsError(pb, s.js, b, tid);
  }
  e = waveReduceSum(e);
  if (lane == 0) {
    s.red[wave] = e;
  }
  __syncthreads();
  const double tot = (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]);
  __syncthreads();
  if (unrounded != nullptr) {
    *unrounded = tot;
  }
  return double(float(tot));
}

__device__ __forceinline__ float blockSumF(const FusedLds& s, float v, int tid) {
  v = waveReduceSumF(v);
  if ((tid & 63) == 0) {
    s.red[4 + (tid >> 6)] = double(v);
  }
  __syncthreads();
  const float tot = float((s.red[4] + s.red[5]) + (s.red[6] + s.red[7]));
  __syncthreads();
  return tot;
}

// ---------------------------------------------------------------------------------------------
// (L L^T) x = b with the factor in LDS, by wave 0 alone and without a single workgroup barrier
// inside.  Off-diagonal tiles hold L_jk; a diagonal tile holds L_kk in its lower triangle and the
// strict upper triangle of L_kk^-T above it (written by the panel factorisation), invDiag[i] =
// 1 / L(i,i) = the diagonal of L_kk^-T.  With the inverse of the diagonal blocks at hand a block
// step is two short dot products instead of a 16-step substitution chain:
//   forward :  y_k = L_kk^-1 (b_k - sum_{j<k} L_kj y_j)        backward:  x_k = L_kk^-T (y_k - sum_{j>k} L_jk^T x_j)
// Lane 4 i + g owns row i of the current block and a quarter g of every dot product; quarters are
// added with quad-permute DPP moves; the 16 right-hand sides of a block are gathered from their
// quads with ds_bpermute (one LDS-crossbar trip instead of a store and a load).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float quadSum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); // lanes ^ 1
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)); // lanes ^ 2
  return v;
}

// left-looking form (one partial sum, the tiles of block row k read when block k is due): for the
// wide systems, where NB accumulators per direction and NB^2 / 2 unrolled tile products do not pay
template <int NB>
__device__ __forceinline__ void solveLLtLeft(const float* L, const float* invDiag, float* x, int tid) {
  if (tid < 64) {
    const int i = tid >> 2, g = tid & 3;
#pragma unroll
    for (int k = 0; k < NB; ++k) { // forward
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < k; ++j) {
        acc = dot4(ldsRow4(L + 256 * tileIndex(k, j), i, g), *reinterpret_cast<const float4*>(x + 16 * j + 4 * g), acc);
      }
      acc = quadSum(acc);
      float* xk = x + 16 * k;
      const float rhs = xk[i] - acc; // the same value in the four lanes of quad i
      const float invd = invDiag[16 * k + i];
      const float* Dk = L + 256 * tileIndex(k, k);
      float p = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 4 * t + g; // L_kk^-1 (i, c) = L_kk^-T (c, i)
        const float m = Dk[tileAddr(c, i)];
        const float rc = __shfl(rhs, 4 * c, 64); // right-hand side of row c straight from its quad
        p += (c < i ? m : (c == i ? invd : 0.f)) * rc;
      }
      p = quadSum(p);
      if (g == 0) {
        xk[i] = p;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
#pragma unroll
    for (int k = NB - 1; k >= 0; --k) { // backward
      float acc = 0.f;
#pragma unroll
      for (int j = k + 1; j < NB; ++j) {
        const float* Tj = L + 256 * tileIndex(j, k);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int c = 4 * t + g;
          acc += Tj[tileAddr(c, i)] * x[16 * j + c]; // L(16 j + c, 16 k + i)
        }
      }
      acc = quadSum(acc);
      float* xk = x + 16 * k;
      const float rhs = xk[i] - acc;
      const float invd = invDiag[16 * k + i];
      const float4 row = ldsRow4(L + 256 * tileIndex(k, k), i, g); // L_kk^-T (i, 4g..4g+3)
      const int c0 = 4 * g;
      float p = (c0 > i ? row.x : (c0 == i ? invd : 0.f)) * __shfl(rhs, 4 * c0, 64);
      p += (c0 + 1 > i ? row.y : (c0 + 1 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 1), 64);
      p += (c0 + 2 > i ? row.z : (c0 + 2 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 2), 64);
      p += (c0 + 3 > i ? row.w : (c0 + 3 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 3), 64);
      p = quadSum(p);
      if (g == 0) {
        xk[i] = p;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  __syncthreads();
}

// The left-looking form with the dependent chain cut to the NEWEST block: the sums over the blocks finished two steps ago
// and earlier are formed a step ahead (their LDS reads fly under the current step's chain), the newest block's solution comes
// over in registers (ds_bpermute from its quads) instead of through its store -- one partial sum and four values more than the
// plain form, against the 2 NB partial sums of the right-looking one.  Forward: the blocks in solveLLtLeft's order (each block's
// four products packed: dot4pk); backward: the blocks are added from the last one down (the plain form: from k + 1 up).  While the kernel spilled
// (default code generation, 128 registers) this lost 0-5 %; without spills (momentum_amd/build.py SOLVE_KERNEL_FLAGS) it gains
// 0.6 % on BASELINE configs[1] and 1.4 % on cfg3 (profiles/r05_exp_fused.txt).
template <int NB>
__device__ __forceinline__ void solveLLtLeftAhead(const float* L, const float* invDiag, float* x, int tid) {
  if (tid < 64) {
    const int i = tid >> 2, g = tid & 3;
    float pre = 0.f; // block k's sum over j < k - 1
    float4 yq{0.f, 0.f, 0.f, 0.f}; // y_{k-1}[4g .. 4g+3]
#pragma unroll
    for (int k = 0; k < NB; ++k) { // forward
      float preNext = 0.f; // block k + 1's sum over j < k: independent of this step
      if (k + 1 < NB) {
#pragma unroll
        for (int j = 0; j < k; ++j) {
          preNext = dot4pk(ldsRow4(L + 256 * tileIndex(k + 1, j), i, g), *reinterpret_cast<const float4*>(x + 16 * j + 4 * g), preNext);
        }
      }
      float acc = pre;
      if (k > 0) {
        acc = dot4pk(ldsRow4(L + 256 * tileIndex(k, k - 1), i, g), yq, acc);
      }
      acc = quadSum(acc);
      float* xk = x + 16 * k;
      const float rhs = xk[i] - acc; // the same value in the four lanes of quad i
      const float invd = invDiag[16 * k + i];
      const float* Dk = L + 256 * tileIndex(k, k);
      float p = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 4 * t + g; // L_kk^-1 (i, c) = L_kk^-T (c, i)
        const float m = Dk[tileAddr(c, i)];
        const float rc = __shfl(rhs, 4 * c, 64); // right-hand side of row c straight from its quad
        p += (c < i ? m : (c == i ? invd : 0.f)) * rc;
      }
      p = quadSum(p);
      if (g == 0) {
        xk[i] = p;
      }
      if (k + 1 < NB) {
        yq = float4{__shfl(p, 16 * g, 64), __shfl(p, 16 * g + 4, 64), __shfl(p, 16 * g + 8, 64), __shfl(p, 16 * g + 12, 64)};
      }
      pre = preNext;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    pre = 0.f; // block k's sum over j > k + 1
    float xq[4] = {0.f, 0.f, 0.f, 0.f}; // x_{k+1}[4t + g]
#pragma unroll
    for (int k = NB - 1; k >= 0; --k) { // backward
      float preNext = 0.f; // block k - 1's sum over j > k
      if (k > 0) {
#pragma unroll
        for (int j = NB - 1; j > k; --j) {
          const float* Tj = L + 256 * tileIndex(j, k - 1);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c = 4 * t + g;
            preNext += Tj[tileAddr(c, i)] * x[16 * j + c]; // L(16 j + c, 16 (k - 1) + i)
          }
        }
      }
      float acc = pre;
      if (k + 1 < NB) {
        const float* Tj = L + 256 * tileIndex(k + 1, k);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc += Tj[tileAddr(4 * t + g, i)] * xq[t];
        }
      }
      acc = quadSum(acc);
      float* xk = x + 16 * k;
      const float rhs = xk[i] - acc;
      const float invd = invDiag[16 * k + i];
      const float4 row = ldsRow4(L + 256 * tileIndex(k, k), i, g); // L_kk^-T (i, 4g..4g+3)
      const int c0 = 4 * g;
      float p = (c0 > i ? row.x : (c0 == i ? invd : 0.f)) * __shfl(rhs, 4 * c0, 64);
      p += (c0 + 1 > i ? row.y : (c0 + 1 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 1), 64);
      p += (c0 + 2 > i ? row.z : (c0 + 2 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 2), 64);
      p += (c0 + 3 > i ? row.w : (c0 + 3 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 3), 64);
      p = quadSum(p);
      if (g == 0) {
        xk[i] = p;
      }
      if (k > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          xq[t] = __shfl(p, 4 * (4 * t + g), 64);
        }
      }
      pre = preNext;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  __syncthreads();
}

// solveLLtLeftAhead in two pieces (round 6): the forward substitution of block k by ONE wave without a workgroup barrier --
// the same operations in the same order as solveLLtLeftAhead's k-th forward step (the older blocks' sum, then the newest
// block's products, each as packed pairs), the previous blocks' solutions read from x in LDS instead of coming over in
// registers: bit-identical results -- and the backward sweep on its own.  Block k needs the block rows 0..k of the factor only,
// so a wave the panel factorisation leaves idle solves block k - 1 while the others eliminate panel k (phase H): the forward
// half of the first solve of an iteration disappears from the critical path.
template <int NB>
__device__ __forceinline__ void solveForwardBlockAhead(const float* L, const float* invDiag, float* x, int k, int lane) {
  const int i = lane >> 2, g = lane & 3;
  float acc = 0.f;
  for (int j = 0; j + 1 < k; ++j) {
    acc = dot4pk(ldsRow4(L + 256 * tileIndex(k, j), i, g), *reinterpret_cast<const float4*>(x + 16 * j + 4 * g), acc);
  }
  if (k > 0) {
    acc = dot4pk(ldsRow4(L + 256 * tileIndex(k, k - 1), i, g), *reinterpret_cast<const float4*>(x + 16 * (k - 1) + 4 * g), acc);
  }
  acc = quadSum(acc);
  float* xk = x + 16 * k;
  const float rhs = xk[i] - acc; // the same value in the four lanes of quad i
  const float invd = invDiag[16 * k + i];
  const float* Dk = L + 256 * tileIndex(k, k);
  float p = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = 4 * t + g; // L_kk^-1 (i, c) = L_kk^-T (c, i)
    const float m = Dk[tileAddr(c, i)];
    const float rc = __shfl(rhs, 4 * c, 64); // right-hand side of row c straight from its quad
    p += (c < i ? m : (c == i ? invd : 0.f)) * rc;
  }
  p = quadSum(p);
  if (g == 0) {
    xk[i] = p;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
// the last forward block (kFirst = NB - 1: the ones before it were solved during the factorisation) and the backward sweep of
// solveLLtLeftAhead, by wave 0; ends with a barrier
template <int NB>
__device__ __forceinline__ void solveLLtAheadTail(const float* L, const float* invDiag, float* x, int tid) {
  if (tid < 64) {
    solveForwardBlockAhead<NB>(L, invDiag, x, NB - 1, tid);
    const int i = tid >> 2, g = tid & 3;
    float pre = 0.f; // block k's sum over j > k + 1
    float xq[4] = {0.f, 0.f, 0.f, 0.f}; // x_{k+1}[4t + g]
#pragma unroll
    for (int k = NB - 1; k >= 0; --k) { // backward
      float preNext = 0.f; // block k - 1's sum over j > k
      if (k > 0) {
#pragma unroll
        for (int j = NB - 1; j > k; --j) {
          const float* Tj = L + 256 * tileIndex(j, k - 1);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c = 4 * t + g;
            preNext += Tj[tileAddr(c, i)] * x[16 * j + c]; // L(16 j + c, 16 (k - 1) + i)
          }
        }
      }
      float acc = pre;
      if (k + 1 < NB) {
        const float* Tj = L + 256 * tileIndex(k + 1, k);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc += Tj[tileAddr(4 * t + g, i)] * xq[t];
        }
      }
      acc = quadSum(acc);
      float* xk = x + 16 * k;
      const float rhs = xk[i] - acc;
      const float invd = invDiag[16 * k + i];
      const float4 row = ldsRow4(L + 256 * tileIndex(k, k), i, g); // L_kk^-T (i, 4g..4g+3)
      const int c0 = 4 * g;
      float p = (c0 > i ? row.x : (c0 == i ? invd : 0.f)) * __shfl(rhs, 4 * c0, 64);
      p += (c0 + 1 > i ? row.y : (c0 + 1 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 1), 64);
      p += (c0 + 2 > i ? row.z : (c0 + 2 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 2), 64);
      p += (c0 + 3 > i ? row.w : (c0 + 3 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 3), 64);
      p = quadSum(p);
      if (g == 0) {
        xk[i] = p;
      }
      if (k > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          xq[t] = __shfl(p, 4 * (4 * t + g), 64);
        }
      }
      pre = preNext;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  __syncthreads();
}

template <int NB>
__device__ __forceinline__ void solveLLtRight(const float* L, const float* invDiag, float* x, int tid) {
  if (tid < 64) {
    const int i = tid >> 2, g = tid & 3;
    // Right-looking inside the wave: as soon as a block of the solution is known it is folded into
    // the partial sums of all blocks still to come (acc[m]: row i of block m, quarter g).  Only the
    // newest contribution sits on the dependent chain; the older ones fill its LDS wait slots.
    float acc[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      acc[m] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) { // forward: y_k = L_kk^-1 (b_k - sum_{j<k} L_kj y_j)
      const float rhs = x[16 * k + i] - quadSum(acc[k]); // the same value in the four lanes of quad i
      const float invd = invDiag[16 * k + i];
      const float* Dk = L + 256 * tileIndex(k, k);
      float p = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 4 * t + g; // L_kk^-1 (i, c) = L_kk^-T (c, i)
        const float m = Dk[tileAddr(c, i)];
        const float rc = __shfl(rhs, 4 * c, 64); // right-hand side of row c straight from its quad
        p += (c < i ? m : (c == i ? invd : 0.f)) * rc;
      }
      p = quadSum(p);
      if (k + 1 < NB) {
        const float4 yq{__shfl(p, 16 * g, 64), __shfl(p, 16 * g + 4, 64), __shfl(p, 16 * g + 8, 64), __shfl(p, 16 * g + 12, 64)}; // y_k[4g..4g+3]
#pragma unroll
        for (int m = k + 1; m < NB; ++m) {
          acc[m] = dot4(ldsRow4(L + 256 * tileIndex(m, k), i, g), yq, acc[m]);
        }
      }
      acc[k] = p; // block k of y, kept for the backward sweep (every lane of quad i holds y_{16k+i})
    }
    float bacc[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
      bacc[m] = 0.f;
    }
#pragma unroll
    for (int k = NB - 1; k >= 0; --k) { // backward: x_k = L_kk^-T (y_k - sum_{j>k} L_jk^T x_j)
      const float rhs = acc[k] - quadSum(bacc[k]);
      const float invd = invDiag[16 * k + i];
      const float4 row = ldsRow4(L + 256 * tileIndex(k, k), i, g); // L_kk^-T (i, 4g..4g+3)
      const int c0 = 4 * g;
      float p = (c0 > i ? row.x : (c0 == i ? invd : 0.f)) * __shfl(rhs, 4 * c0, 64);
      p += (c0 + 1 > i ? row.y : (c0 + 1 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 1), 64);
      p += (c0 + 2 > i ? row.z : (c0 + 2 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 2), 64);
      p += (c0 + 3 > i ? row.w : (c0 + 3 == i ? invd : 0.f)) * __shfl(rhs, 4 * (c0 + 3), 64);
      p = quadSum(p);
      if (g == 0) {
        x[16 * k + i] = p;
      }
      if (k > 0) {
        // x_k[4t + g] for t = 0..3: the rows this lane pairs with L(16k + 4t + g, 16m + i)
        const float xq[4] = {__shfl(p, 4 * g, 64), __shfl(p, 4 * (4 + g), 64), __shfl(p, 4 * (8 + g), 64), __shfl(p, 4 * (12 + g), 64)};
#pragma unroll
        for (int m = 0; m < k; ++m) {
          const float* Tk = L + 256 * tileIndex(k, m);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            bacc[m] += Tk[tileAddr(4 * t + g, i)] * xq[t]; // L(16 k + c, 16 m + i)
          }
        }
      }
    }
  }
  __syncthreads();
}

// kLeft: the left-looking form also for the small systems -- the instantiations that run four workgroups per CU at 128 registers
// (round 5: the right-looking form's 2 NB partial sums are what the allocator spills there; 1.73 -> 1.80e6 solves/s on BASELINE
// configs[1], profiles/r05_exp_fused.txt; at three workgroups per CU and 168 registers the right-looking form stays ahead)
template <int NB, bool kLeft = false>
__device__ __forceinline__ void solveLLt(const float* L, const float* invDiag, float* x, int tid) {
  if (NB <= 8 && !kLeft) {
    solveLLtRight<NB>(L, invDiag, x, tid);
  } else if (NB <= 8) { // (kLeft: the chain cut to the newest block, see solveLLtLeftAhead)
    solveLLtLeftAhead<NB>(L, invDiag, x, tid);
  } else {
    solveLLtLeft<NB>(L, invDiag, x, tid);
  }
}

// Per-instance constraint parents (mmx_problem_set_instance_parents): the tables that say which units hang on which
// joint, built per element instead of copied from the batch-shared ones -- units sorted by (DFS position of their joint,
// unit index) with a counting rank, so that the own sums add in the same deterministic order as in the shared case.
// All four tables in LDS; returns the number of loaded positions (the same value in every thread); ends with a barrier.
template <class I>
__device__ __forceinline__ int buildInstanceUnitTables(
    const ProblemDev& pb, const FusedDev& fd, int b, int J, int U, int tid, I* unitPos, I* posUnitStart, I* posUnits, I* loadedPos) {
  for (int u = tid; u < U; u += 256) {
    int joint;
    if (u < fd.Kp) {
      joint = pb.instPosParent != nullptr ? pb.instPosParent[size_t(b) * fd.Kp + u] : fd.unitJoint[u];
    } else {
      const int co = (u - fd.Kp) / 3;
      joint = pb.instOriParent != nullptr ? pb.instOriParent[size_t(b) * pb.Ko + co] : fd.unitJoint[u];
    }
    unitPos[u] = I(pb.jointTin[joint]);
  }
  __syncthreads();
  for (int k = tid; k <= J; k += 256) { // units on positions before k
    int cnt = 0;
    for (int u = 0; u < U; ++u) {
      cnt += unitPos[u] < k ? 1 : 0;
    }
    posUnitStart[k] = I(cnt);
  }
  for (int u = tid; u < U; u += 256) {
    const int pu = unitPos[u];
    int rank = 0;
    for (int v = 0; v < U; ++v) {
      const int pv = unitPos[v];
      rank += (pv < pu || (pv == pu && v < u)) ? 1 : 0;
    }
    posUnits[rank] = I(u);
  }
  __syncthreads();
  for (int k = tid; k < J; k += 256) { // loaded positions, ascending
    if (posUnitStart[k + 1] > posUnitStart[k]) {
      int rank = 0;
      for (int q = 0; q < k; ++q) {
        rank += posUnitStart[q + 1] > posUnitStart[q] ? 1 : 0;
      }
      loadedPos[rank] = I(k);
    }
  }
  int numLoaded = 0; // every thread computes the same value
  for (int k = 0; k < J; ++k) {
    numLoaded += posUnitStart[k + 1] > posUnitStart[k] ? 1 : 0;
  }
  __syncthreads();
  return numLoaded;
}

// MODE 0: production; 1: also dump H / g of the first iteration (parity hook); 2: per-phase clocks
// Workgroups per CU follow the LDS footprint (tiles: 1 KB each): three up to NB = 6, two up to NB = 8,
// one beyond -- the register budget is set to match, so the wide systems do not spill.
// kTR: the TrustRegionQRT step rule (its re-solve loops are compiled into that instantiation only)
// kGen: rows of the further joint error functions (plane / aim / fixed axis / normal / ...) and of the ellipsoid limits:
// evaluated per iteration into a small dense block J_g (LDS), added to g, H (matrix-core rank-k update of the tiles),
// the refinement residual and the trial errors.  More LDS, so at most two workgroups per CU.
// kRule: -1 = the step rule, the line search and the parameter-space rows are run-time options (one instantiation for all
// of them); 0 / 1 = the instantiation for GaussNewtonSolverT without a line search (the BASELINE metric) / for the LM
// schedule, both for problems without limit / model-parameter rows: the other rules' branches -- each inlines a whole
// trial evaluation, a further copy of phases A-C --, the on-the-fly evaluation of the parameter-space rows and the state
// they carry from iteration to iteration are compiled out (cfg2: 15.0 k / 19.1 k instead of 27.0 k instructions, 6 / 9
// instead of 36 spilled vector registers, 352 / 380 instead of 456 spilled scalar registers).  (2 = a line search only:
// compiles, but spills MORE than the generic instantiation -- 68 vector registers -- and is not instantiated.)
// Round 3, one box: plain Gauss-Newton +2.1 %, LM schedule +2.5 % over the generic instantiation; parity unchanged.
// kArgLazy: the five descriptor structs in ONE device-resident struct behind a single pointer (the problem's, written by
// stashFusedArgsKernel in stream order before the solve), every field an s_load where it is used, instead of ~200 SGPRs of
// by-value arguments loaded at the entry and kept in VGPR lanes across the whole kernel (444 spilled SGPRs: v_writelane /
// v_readlane).  At 168 registers the form lost 2 % (round 4: profiles/r04_exp_argptr.txt); at 128 it frees vector registers:
// spilled VGPRs 75 -> 36 (LM schedule), 129 -> 94 (generic rule), and the line-search line gains 4 %, the LM one 1 %, plain
// Gauss-Newton nothing (profiles/r05_exp_fused.txt).  (Reading them through the kernel-argument segment's own pointer instead
// was tried: the compiler knows that segment dereferenceable and hoists the loads back to the entry -- 50 / 67 / 104 spilled.)
// The four-workgroup production instantiations take it for problems with a shared rig and shared weights (the per-element
// selections write into the by-value copies: those problems keep the by-value form).
struct FusedArgs {
  RigDev rig;
  ProblemDev pb;
  FusedDev fd;
  SolveStateDev st;
  FusedParams fp;
};
static __global__ void stashFusedArgsKernel(FusedArgs a, FusedArgs* dst) {
  *dst = a;
}
// kMix: the MIXED-precision instantiation (mmx_gn_options::precision == MMX_PRECISION_MIXED; mmx_mixed.hpp): theta, forward
// kinematics, units, g = J^T r and the linear solve's residual in double -- all O(J + U) tree passes --, H / factor / triangular
// solves in single precision as the preconditioner of a conjugate-gradient iteration in double.  Generic rule only (kRule = -1,
// no parameter-space rows, no trust region, no general rows); three workgroups per CU up to six blocks.
template <int NB, int MODE, bool kTR, bool kGen = false, int kRule = -1, bool kArgLazy = false, bool kMix = false>
// Workgroups per CU the register budget is set for (the LDS footprint decides what actually runs): FOUR for the per-rule
// instantiations up to six blocks (round 5: the lifetime-shared carve brings BASELINE configs[1] to 40.2 KB; 128 VGPRs cost
// a workgroup 4.6 % of its latency -- measured with the LDS still at 52 KB, profiles/r05_exp_fused.txt -- and buy a third
// more of them per CU), three for the generic ones (their trial evaluations spill at 128), two up to eight blocks and for
// the general rows, one beyond.
__global__ void __launch_bounds__(256, (NB <= 6 && !kGen ? (FusedFour<NB, kTR, kGen, kRule>::value && !kMix ? 4 : 3) : (NB <= 8 ? 2 : 1))) fusedSolveKernel(
    const FusedArgs* __restrict__ argsDev, // kArgLazy: the descriptors (the by-value ones are not read); else unused
    const RigDev rigArg, // (by value, read-only: the per-element copies are taken BELOW the element-list gate -- a modified by-value
    const ProblemDev pbArg, // argument is copied to scratch at entry, 90 KB of stores per workgroup that an empty one paid too)
    FusedDev fdV,
    float* __restrict__ theta, // [B][P] in/out
    SolveStateDev stV,
    FusedParams fpV,
    float* __restrict__ dbgH, // [B][n*n] or null: H = J^T J (no lambda) of the FIRST iteration
    float* __restrict__ dbgG, // [B][n] or null
    long long* __restrict__ dbgClk, // [32] or null: per-phase cycle counts of block 0 (profiling aid)
    MixSelect sel) { // kMix: the elements to solve and where their initial parameters are (MMX_PRECISION_AUTO's second pass); else unused
  static_assert(!kMix || (!kTR && !kGen && kRule == -1 && !kArgLazy), "the mixed-precision instantiation: generic rule, reference rows");
  constexpr int T = NB * (NB + 1) / 2; // lower-triangle tiles
  constexpr int NP = 16 * NB; // padded system size
  // lookahead: the left-looking updates of block column k + 1 by the columns before k ride under panel k's elimination
  // chain (waves without a panel row); NB <= 8: wave 3 never holds one (round 3, measured on one box: +2.2 % on cfg2)
  constexpr bool kLook = NB <= 8;
  long long clkLast = 0;
  // (the clocked instantiation only) dbgClk[31] = slot + 1: every workgroup ENDS the first time it reaches that stamp -- counter
  // passes over launches cut after successive phases attribute a per-kernel PMC (LDS bank conflicts) to the phases by
  // difference (scripts/lds_conflicts_by_phase.sh; the stamps used are the workgroup-uniform ones)
  const int dbgStop = (MODE == 2 && dbgClk != nullptr) ? int(dbgClk[31]) : 0;
#define MMX_CLK(slot)                                             \
  if (MODE == 2 && blockIdx.x == 0 && threadIdx.x == 0) {         \
    const long long now_ = clock64();                             \
    dbgClk[slot] += now_ - clkLast;                               \
    clkLast = now_;                                               \
  }                                                               \
  if (MODE == 2 && dbgStop == (slot) + 1) {                       \
    __builtin_amdgcn_endpgm();                                    \
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // MMX_PRECISION_AUTO's second pass (kMix with an element list): workgroup i takes element map[i]; the ones beyond *count leave at
  // once.  (Round 6 measured a FIXED grid walking the list instead -- nothing to pay when the list is empty: AUTO on an unmarked
  // batch - 3.5 % instead of - 7 % at 65 536 --, but the kernel arguments then stay live across the whole body for the next element:
  // 34 instead of 8 spilled VGPRs, 504 instead of 309 spilled SGPRs, and MMX_PRECISION_MIXED itself 3.5 % slower.  Not kept.)
  int bSel = blockIdx.x;
  if (kMix && sel.map != nullptr) {
    if (int(blockIdx.x) >= *sel.count) {
      return;
    }
    bSel = sel.map[blockIdx.x];
  }
  const int b = bSel, tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  RigDev rigV;
  ProblemDev pbV;
  if (!kArgLazy) { // (the element's selections go into copies of the by-value descriptors)
    rigV = rigArg;
    pbV = pbArg;
    selectInstanceRig(rigV, b);
    selectInstanceWeights(pbV, b);
  }
  const RigDev& rig = kArgLazy ? argsDev->rig : rigV;
  const ProblemDev& pb = kArgLazy ? argsDev->pb : pbV;
  const FusedDev& fd = kArgLazy ? argsDev->fd : fdV;
  const SolveStateDev& st = kArgLazy ? argsDev->st : stV;
  const FusedParams& fp = kArgLazy ? argsDev->fp : fpV;
  const int J = rig.J, P = rig.P, U = fd.U, n = fd.n, nsrc = fd.nsrc;
  const int kR = rig.R;

  // ---- LDS carve (every offset a multiple of 4 floats); must match fusedLdsBytes().  Round 5: laid out by LIFETIME so that
  // the 72-joint problems fit a quarter of a CU's LDS (BASELINE configs[1]: 52.2 -> 40.2 KB, four workgroups per CU):
  //   * the integer tables are 16-bit (every index of a fused problem is below 4096), the parameter transform's CSR stays
  //     in global memory (read by the prefetching walk csrRowsPrefetched: one L2 round trip per use);
  //   * first-order own / subtree sums of phases D-E live in the assembly scratch (over the FK buffer / the unit moments),
  //     those of the refinement in the refinement's arena;
  //   * the arena (phases J-K: jd | tanOwn | tanPre | own sums | subtree sums | per-slot gradients, each over a dead
  //     predecessor; dfull) shares its place with the slot tables srcT of phases E-G;
  //   * uy (dead after phase D) shares its place with the slots' gradient shares (E-F), the split entries' partial
  //     cells (G) and rho | invDiag (from the end of G on).
  FusedLds s;
  int16_t *lParentPos, *lSubSize, *lPosUnitStart, *lPosUnits, *lUnitJoint, *lSolveList, *lDfsJoint, *lLoadedPos, *lColToSolve;
  int *lParent, *lPtOuter = nullptr, *lPtInner = nullptr;
  float *srcGu, *cells, *arena, *lPtValue = nullptr;
  // the instantiations whose register budget is set for four workgroups per CU read the transform's CSR from global memory
  // (their LDS has no room for it); the others keep their LDS copy
  constexpr bool kFour = FusedFour<NB, kTR, kGen, kRule>::value && !kMix;
  constexpr bool kCsrLds = !kFour && !kMix; // (the mixed instantiation walks the transform in double from global memory: mixTransformRows)
  // the register-lean forms of three routines in the four-workgroup instantiations (A/B variants: the full forms back, one each)
  constexpr bool kLeanSolve = kFour;
  constexpr bool kRide = kFour && NB >= 2; // the first solve's forward substitution rides under the panel factorisation (wave 3); A/B: r06_exp_fused.txt item 1
  constexpr bool kLeanOps = kFour;
  constexpr bool kLeanAcc = kFour;
  const int kNnz = fd.nnz;
  const FusedLayout lay = fusedLayout(NB, J, P, U, nsrc, n, fd.numCells, kRule < 0, kGen ? fd.GT : 0, kGen ? fd.genRows : 0, kTR, 0, kMix);
  MixLds m{};
  {
    float* p = smem;
    auto take = [&](size_t count) {
      float* r = p;
      p += alignUp4(count);
      return r;
    };
    auto takeS = [&](size_t count) { return reinterpret_cast<int16_t*>(take((count + 1) / 2)); };
    s.mStart = takeS(NP + 1);
    s.mTin = reinterpret_cast<int*>(take(nsrc));
    s.mInfo = reinterpret_cast<int*>(take(nsrc));
    s.mW = take(nsrc);
    lParent = reinterpret_cast<int*>(take(J));
    lParentPos = takeS(J);
    lSubSize = takeS(J);
    lPosUnitStart = takeS(J + 1);
    lPosUnits = takeS(U);
    lUnitJoint = takeS(U);
    lSolveList = takeS(n);
    lDfsJoint = takeS(J);
    lLoadedPos = takeS(J);
    lColToSolve = takeS(P);
    if (kCsrLds) {
      lPtOuter = reinterpret_cast<int*>(take(kR + 1));
      lPtInner = reinterpret_cast<int*>(take(kNnz));
      lPtValue = take(kNnz);
    }
    if (kMix) { // (theta, joint states, residual rows, g and the step live in double: below)
      s.th = s.js = s.up = s.us = s.ur = s.g = s.d0 = nullptr;
      m.lo = takeS(J);
      m.hi = takeS(J);
    } else {
      s.th = take(P);
      s.js = take(size_t(kJs) * J);
      s.up = take(3 * size_t(U));
      s.ur = take(3 * size_t(U));
      s.us = take(U);
      s.g = take(NP);
      s.d0 = take(NP);
    }
    s.uy = take(lay.uyFloats);
    s.rho = s.uy + lay.auxOff, s.invDiag = s.rho + NP, srcGu = s.rho, cells = s.rho + lay.cellOff;
    s.red = reinterpret_cast<double*>(take(kMix ? 40 : 16)); // (kMix: five sums per reduction round)
    s.flags = reinterpret_cast<int*>(take(8)); // [4 .. 6] as floats: the precision estimate's accumulators (estAcc below)
    s.gEv = s.gRes = s.gW = s.gJ = nullptr;
    if (kGen) {
      const int rowsGp = (fd.genRows + 3) & ~3;
      s.gEv = take(size_t(kGenEv) * size_t(fd.GT));
      s.gRes = take(rowsGp);
      s.gW = take(rowsGp);
      s.gJ = take(size_t(rowsGp) * size_t(srcStrideFor(NP)));
    }
    arena = take(lay.arenaFloats); // srcT (phases E-G)  |  the refinement's buffers and dfull (phases J-K)
    if (kMix) { // ... | the double scratch X, Y (everywhere else)
      m.X = reinterpret_cast<double*>(arena);
      m.Y = m.X + mixXDoubles(J, nsrc);
    }
    s.srcT = arena;
    s.tanOwn = arena, s.jd = arena + lay.t9, s.tanPre = arena + lay.t9, s.dfull = arena + 2 * lay.t9;
    float* region = p;
    s.fkA = reinterpret_cast<double*>(take(fkBufFloats(J)));
    s.fkB = reinterpret_cast<double*>(p); // over own2 / sub2 (2 kC2 J >= fkBufFloats(J) floats)
    s.own2 = take(size_t(kC2) * J);
    s.sub2 = take(size_t(kC2) * J);
    s.umom = take(lay.umomFloats);
    s.own1 = region, s.sub1 = s.umom; // phases D-E: over the FK buffer (dead after FK) / the unit moments (dead after the own sums)
    s.L = region;
    if (kMix) { // the double arrays, behind everything else (every offset so far is a multiple of 16 bytes)
      double* d = reinterpret_cast<double*>(smem + (lay.total - alignUp4(2 * mixPersistentDoubles(J, P, U, NP))));
      m.th = d, d += P;
      m.js = d, d += size_t(kJsD) * J;
      m.up = d, d += 3 * size_t(U);
      m.uf = d, d += 3 * size_t(U);
      m.us = d, d += U;
      m.g = d, d += NP;
      m.x = d, d += NP;
      m.r = d, d += NP;
      m.p = d, d += NP;
      m.q = d;
    }
  }

  float* thg = theta + size_t(b) * P;
  if (kMix) {
    const float* th0 = sel.thetaInit != nullptr ? sel.thetaInit + size_t(b) * P : thg;
    for (int i = tid; i < P; i += 256) {
      m.th[i] = double(th0[i]);
    }
  } else {
    for (int i = tid; i < P; i += 256) {
      s.th[i] = thg[i];
    }
  }
  for (int c = tid; c <= NP; c += 256) {
    s.mStart[c] = int16_t(fd.srcStart[c]);
  }
  for (int e = tid; e < nsrc; e += 256) {
    const ColumnSourceDev cs = fd.srcs[e];
    s.mTin[e] = cs.tin | (cs.tout << 16);
    s.mInfo[e] = cs.joint | (cs.dof << 12) | ((cs.parent + 1) << 16);
    s.mW[e] = cs.weight;
  }
  for (int i = tid; i < J; i += 256) {
    lParent[i] = rig.parent[i];
    lSubSize[i] = int16_t(fd.subSize[i]);
  }
  for (int i = tid; i <= J; i += 256) {
    lPosUnitStart[i] = int16_t(fd.posUnitStart[i]);
  }
  if (kCsrLds) {
    for (int i = tid; i <= kR; i += 256) {
      lPtOuter[i] = rig.ptOuter[i];
    }
    for (int i = tid; i < kNnz; i += 256) {
      lPtInner[i] = rig.ptInner[i];
      lPtValue[i] = rig.ptValue[i];
    }
  }
  for (int i = tid; i < U; i += 256) {
    lPosUnits[i] = int16_t(fd.posUnits[i]);
    lUnitJoint[i] = int16_t(pb.unitTin[i]); // DFS position of the unit's joint
  }
  for (int i = tid; i < n; i += 256) {
    lSolveList[i] = int16_t(fd.solveList[i]);
  }
  for (int i = tid; i < J; i += 256) {
    lDfsJoint[i] = int16_t(fd.dfsJoint[i]);
    lLoadedPos[i] = int16_t(i < fd.numLoaded ? fd.loadedPos[i] : 0);
  }
  for (int i = tid; i < P; i += 256) {
    lColToSolve[i] = -1;
  }
  if (kGen) { // pad rows / pad columns of J_g stay zero for the whole launch
    const int rowsGp = (fd.genRows + 3) & ~3;
    for (int i = tid; i < rowsGp * srcStrideFor(NP); i += 256) {
      s.gJ[i] = 0.f;
    }
    for (int i = tid; i < rowsGp; i += 256) {
      s.gRes[i] = 0.f;
      s.gW[i] = 0.f;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    lColToSolve[fd.solveList[i]] = int16_t(i);
  }
  // Per-instance constraint parents (mmx_problem_set_instance_parents): the tables that say which units
  // hang on which joint are built here, per element, instead of copied from the batch-shared ones --
  // units sorted by (DFS position of their joint, unit index) with a counting rank, so the summation
  // order of the own sums is the same deterministic one as in the shared case.  Once per solve.
  int numLoadedInst = fd.numLoaded;
  if (pb.instPosParent != nullptr || pb.instOriParent != nullptr) {
    __syncthreads();
    numLoadedInst = buildInstanceUnitTables(pb, fd, b, J, U, tid, lUnitJoint, lPosUnitStart, lPosUnits, lLoadedPos);
  }
  // parentPos[k] = DFS position of the parent of the joint at DFS position k (-1 for a root), built
  // through a joint -> position scratch map (alt is free until the first FK)
  {
    int* posOf = reinterpret_cast<int*>(s.fkA);
    for (int k = tid; k < J; k += 256) {
      posOf[fd.dfsJoint[k]] = k;
    }
    __syncthreads();
    for (int k = tid; k < J; k += 256) {
      const int par = rig.parent[fd.dfsJoint[k]];
      lParentPos[k] = int16_t(par >= 0 ? posOf[par] : -1);
    }
  }
  // from here on the kernel reads the batch-shared tables through these LDS-backed views
  RigView rv;
  rv.J = J, rv.P = P, rv.R = kR, rv.numLevels = rig.numLevels, rv.jumpRounds = rig.jumpRounds;
  rv.parent = lParent, rv.preRot = asGlobal(rig.preRot), rv.offset = asGlobal(rig.offset);
  // (asGlobal: read out of the lazily loaded descriptors the pointers are generic and their loads flat)
  rv.ptOuter = kCsrLds ? lPtOuter : asGlobal(rig.ptOuter), rv.ptInner = kCsrLds ? lPtInner : asGlobal(rig.ptInner), rv.ptValue = kCsrLds ? lPtValue : asGlobal(rig.ptValue); // (LDS copy, or global: csrRowsPrefetched)
  rv.ptOffsets = asGlobal(rig.ptOffsets);
  rv.hasOffsets = rig.ptOffsetsNonZero != 0;
  rv.levelOrder = nullptr, rv.levelStart = nullptr; // (the pointer-jumping FK needs neither)
  rv.rowRec = asGlobal(rig.ptRowRec), rv.numRowRec = rig.numRowRec;
  FusedViewS fv;
  fv.U = U, fv.Kp = fd.Kp;
  fv.dfsJoint = lDfsJoint, fv.loadedPos = lLoadedPos, fv.numLoaded = numLoadedInst, fv.colToSolve = lColToSolve;
  fv.subSize = lSubSize, fv.unitPos = lUnitJoint, fv.posUnitStart = lPosUnitStart, fv.posUnits = lPosUnits;
  fv.solveList = lSolveList;
  // the precision estimate's accumulators, kept by thread 0 in LDS (not in registers: they are touched once per iteration):
  // [0] largest w = kPivotFloor (H_jj + mu) / d_jj of the solve, [1] largest w x sqrt(error of the iteration / error of the
  // first) over the iterations, [2] the first iteration's error (< 0: none yet)
  float* estAcc = reinterpret_cast<float*>(s.flags + 4);
  if (tid == 0) {
    s.flags[0] = 0; // stop
    s.flags[1] = 0; // not positive definite (this iteration)
    s.flags[2] = 0; // status
    estAcc[0] = estAcc[1] = 0.f, estAcc[2] = -1.f;
  }
  int4 mixRec{0, 0, 0, 0}; // kMix: this thread's transform-row record (RigDev::ptRowRec), kept for the whole solve
  const bool mixHasRec = kMix && rv.rowRec != nullptr && rv.numRowRec <= 256;
  if (mixHasRec && tid < rv.numRowRec) {
    mixRec = rv.rowRec[tid];
  }
  const MixRowRec mixRecP{mixRec, mixHasRec};
  int mixLevel = 0; // kMix: the tree level of joint `tid` (blockFkD composes one level per barrier)
  if (kMix && tid < J) {
    for (int a = lParent[tid]; a >= 0; a = lParent[a]) {
      ++mixLevel;
    }
  }
  if (kMix) { // per DFS position: the index range of the (ascending) loaded positions inside the joint's subtree
    for (int k = tid; k < J; k += 256) {
      const int end = k + lSubSize[k];
      int lo = 0, hi = 0;
      for (int li = 0; li < numLoadedInst; ++li) {
        const int pp = lLoadedPos[li];
        lo += pp < k ? 1 : 0;
        hi += pp < end ? 1 : 0;
      }
      m.lo[k] = int16_t(lo), m.hi[k] = int16_t(hi);
    }
  }
  const bool hasParamRows = kRule >= 0 ? false : (pb.M > pb.rowsJoint); // limit / model-parameter rows present (uniform)
  double lastError = DBL_MAX; // solver.cpp:84-85 (kept by thread 0)
  float lambda = fp.lambda; // constant for GaussNewtonSolverT, adapted by the LM schedule
  double lambdaD = double(fp.lambda); // kMix: the schedule's damping as GaussNewtonSolverT<double> carries it (lambda = its rounding: what is factored)
  bool mixUnconverged = false; // kMix: some iteration's conjugate gradients stopped at the step limit
  int mixApplied = 0; // kMix: operator applications of the whole solve (mmx_problem_solve_diagnostics [2]: per iteration)
  float trRadius = fp.trustRadius; // TrustRegionQRT::curTrustRegionRadius_ (initializeSolver, trust_region_qr.cpp:38-41)
  double curError = DBL_MAX;
  int itersDone = 0;
  float pivotWorst = 0.f; // largest kPivotFloor (H_jj + mu) / d_jj of the solve, per diagonal lane (the precision estimate's input)
  float refineWorst = 0.f; // largest |last correction|^2 / |step|^2 of a refinement
  // the joint states / units in LDS already belong to s.th (left by an accepted trial of the line search or the LM
  // schedule, blockError<kStore>); stateError = the error an evaluation of phases A-C would report for it
  // (measured: line search 1.39 -> 1.52e6, LM schedule 1.41 -> 1.53e6 solves/s at cfg2 / cfg3)
  constexpr bool kReuse = !kGen && !kTR && kRule != 0;
  const int stepRule = kRule < 0 ? fp.stepRule : (kRule == 1 ? 1 : 0);
  const int doLineSearch = kRule < 0 || kRule == 2 ? fp.doLineSearch : 0;
  bool stateValid = false;
  double stateError = 0.0;
  // (the parameter transform's CSR in global memory is walked twice per iteration by the instantiations without an LDS copy.  Two
  // ways of hiding the walk's L2 latency in registers -- the rows' bounds kept and the first entries requested a phase ahead;
  // the bounds only -- were built and measured in round 5 and each LOST 7 % at 128 registers: profiles/r05_exp_fused.txt item 4.)
  __syncthreads();

  if (MODE == 2) {
    clkLast = clock64();
  }
  const int tidOuter = tid;
  for (int it = 0; it < fp.maxIterations; ++it) {
    // Opaque per-iteration copies of the thread index: everything derived from them (tile
    // addresses, row assignments, run boundaries ...) is recomputed where it is used instead of
    // being hoisted out of the iteration loop, where those per-thread invariants overflowed the
    // register budget of three workgroups per CU and were reloaded from scratch at ~140 sites.
    int tid = tidOuter;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the constraint payload of this thread's unit is requested before FK so that its HBM latency
    // hides behind it (one round trip per iteration instead of two) -- except in the instantiations that run four
    // workgroups per CU at 128 registers: eleven registers across FK cost them more than the round trip (an L2 hit from the
    // second iteration on): 1.716 -> 1.726e6 solves/s on BASELINE configs[1], profiles/r05_exp_fused.txt
    constexpr bool kLateUnit = kFour || kMix;
    const UnitInput uin0 = kLateUnit ? UnitInput{} : loadUnitInput(pb, b, tid < U ? tid : U);
    // TrustRegionQRT::doIteration (momentum/character_solver/trust_region_qr.cpp:52-270) wraps what follows
    // in up to ten trial steps (:157); every other step rule passes through once.
    float trLambda = 1e-10f; // :86, only grows within an iteration (:213-224)
    int trustStep = 0;
    bool trNoStep = false; // the step is not worth taking (:164) or could not be computed: the parameters stay
    float trDn2 = 0.f, trDg = 0.f, trMu = 0.f; // |step|^2, step . J^T r, damping of the step on the table
    bool notPd = false; // (kept false since the pivot floor: the factorisation always completes and the step is always taken, as in
                        // the reference, which never looks at LLT::info(); the branches it guards are dead code the compiler removes)
    bool badPivot = false; // a raw pivot was not positive: reported as MMX_SOLVE_NOT_PD
    bool floored = false; // the factor's damping floor exceeded the caller's damping: reported as MMX_SOLVE_DAMPING_FLOORED
    for (;;) {
    int tidT = tid;
    if (kTR) { // (no per-thread invariant of the body is to live across the re-solve loops either)
      asm volatile("" : "+v"(tidT));
    }
    {
    const int tid = tidT;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (kReuse && stateValid) {
      curError = stateError;
    } else if (kMix) {
      // ================= A-C in double (mmx_mixed.hpp)
      blockFkD(rv, m, m.th, tid, mixLevel, true, mixRecP);
      double e = mixUnits<true>(pb, s, m, b, U, tid);
      if (hasParamRows) {
        e += paramRowsErrorD<true>(rig, pb, P, m.th, b, tid);
      }
      e = waveReduceSum(e);
      if (lane == 0) {
        s.red[wave] = e;
      }
      __syncthreads();
      curError = (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]);
      MMX_CLK(1)
    } else {
    // ================= A+B: forward kinematics (local transforms, pointer-jumping composition, rotation axes)
    blockFk<!kCsrLds>(rv, s, s.th, tid, true, MODE == 2 ? dbgClk : nullptr, &clkLast);
    MMX_CLK(1)
    // ================= C: units (need only the world transforms, not the axes)
    {
      double e = 0.0;
      for (int u = tid; u < U; u += 256) {
        const Unit un = evalUnitFrom(pb, (!kLateUnit && u == tid) ? uin0 : loadUnitInput(pb, b, u), s.js, u);
        s.up[3 * u] = un.v.x, s.up[3 * u + 1] = un.v.y, s.up[3 * u + 2] = un.v.z;
        const float rx = un.sigma * un.f.x, ry = un.sigma * un.f.y, rz = un.sigma * un.f.z;
        s.ur[3 * u] = rx, s.ur[3 * u + 1] = ry, s.ur[3 * u + 2] = rz;
        s.uy[3 * u] = un.sigma * rx, s.uy[3 * u + 1] = un.sigma * ry, s.uy[3 * u + 2] = un.sigma * rz;
        s.us[u] = un.sigma;
        e += double(un.werr);
      }
      if (hasParamRows) {
        e += paramRowsError<true>(rig, pb, P, s.th, b, tid);
      }
      if (kGen) { // the further joint error functions and the ellipsoid limits: records for the rows of J_g, residual rows
        e += generalRowsEvaluate(pb, s.js, b, U, tid, s.gEv, s.gRes);
      }
      e = waveReduceSum(e);
      if (lane == 0) {
        s.red[wave] = e;
      }
    }
    __syncthreads();
    curError = (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]); // every thread: the same value
    } // (phases A-C)
    if (kGen) {
      generalRowsGather(
          s.js, s.gEv, s.gJ, srcStrideFor(NP), fd.GT, n, tid,
          [&](int e) {
            const int span = s.mTin[e], info = s.mInfo[e];
            return GenSlot{span & 0xffff, span >> 16, info & 0xfff, (info >> 12) & 7, (info >> 16) - 1, s.mW[e]};
          },
          [&](int c, int& e0, int& e1) { e0 = NP + s.mStart[c], e1 = NP + s.mStart[c + 1]; });
      __syncthreads();
    }
    MMX_CLK(2)
    if (kMix) { // ================= F first, in double: g = J^T S^2 f (the arena is free: the slot tables come later)
      __syncthreads(); // (s.red of the error sum / a reused trial state: every thread is past its reads)
      mixAdjoint(
          fv, s, m, J, NP, n, nsrc, tid,
          [&](int u, int, const D3&) {
            const double sg = m.us[u];
            return (sg * sg) * D3{m.uf[3 * u], m.uf[3 * u + 1], m.uf[3 * u + 2]};
          },
          m.g);
      __syncthreads();
      MMX_CLK(5)
    }
    int newtonIter = 0, pdRetries = 0;
    for (;;) { // one pass per value of the damping (the trust region's Newton updates change it, :180-231)
    int tidS = tid;
    if (kTR) {
      asm volatile("" : "+v"(tidS));
    }
    {
    const int tid = tidS;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the reference seeds R with lambda ON its diagonal (online_householder_qr.cpp:133-140) and appends
    // sqrt(lambda_new - lambda) I rows: R^T R = J^T J + (1e-20 + lambda - 1e-10) I
    const float mu = kTR ? 1e-20f + (trLambda - 1e-10f) : lambda;
    // ================= D: own + subtree sums
    if (kMix) {
      ownSums<256, FusedViewS, true>(fv, s, s.umom, U, tid, m.up, m.us); // (second moments only: they feed H, the preconditioner)
    } else {
      ownSums(fv, s, s.umom, U, tid);
    }
    __syncthreads();
    MMX_CLK(15)
    constexpr int kUnD = kFusedTreeUn;
    if (!kMix) {
      treeSum<kC1, true, kC1, kUnD>(fv, s.own1, s.sub1, J, wave, lane);
    }
    treeSum<kC2Used, true, kC2, kUnD>(fv, s.own2, s.sub2, J, wave, lane);
    __syncthreads();
    MMX_CLK(3)
    // ================= E: per-slot tables (weight folded in), channel-major
    const int sst = srcStrideFor(nsrc);
    float* srcD = s.srcT; // [7][sst]  G0(3) AX(3) TR
    float* srcA = s.srcT + 7 * sst; // [7][sst]  AL(3) BV(3) BS
    float* srcG = srcGu; // [nsrc]    the slot's share of g = J^T r (in uy's place: dead since phase D, rho / invDiag not yet written)
    for (int e = tid; e < nsrc; e += 256) {
      ColumnSourceDev cs;
      {
        const int info = s.mInfo[e];
        cs.joint = info & 0xfff;
        cs.dof = (info >> 12) & 7;
        cs.parent = (info >> 16) - 1;
        cs.tin = s.mTin[e] & 0xffff;
      }
      float jaMix[kJs]; // kMix: the slot's joint state rounded to single precision (H is the preconditioner)
      if (kMix) {
        const double* ad = m.js + kJsD * cs.joint;
#pragma unroll
        for (int q = 0; q < kJs; ++q) {
          jaMix[q] = float(ad[q]);
        }
      }
      const float* a = kMix ? jaMix : s.js + kJs * cs.joint;
      const float* sb = s.sub2 + kC2 * cs.tin;
      const F3 ta{a[0], a[1], a[2]};
      const float m0 = sb[0];
      const F3 m1{sb[1], sb[2], sb[3]};
      F3 al, bv{0.f, 0.f, 0.f}, g0, ax;
      float bs = 0.f, tr;
      if (cs.dof < 3) {
        if (kMix) {
          const D3 ad = transAxisColD(m.js, cs.parent, cs.dof);
          al = F3{float(ad.x), float(ad.y), float(ad.z)};
        } else {
          al = transAxisCol(s.js, cs.parent, cs.dof);
        }
        g0 = m0 * al;
        ax = cross(m1, al);
        tr = dot(al, m1);
      } else if (cs.dof < 6) {
        const float* w = a + 8 + 3 * (cs.dof - 3);
        const F3 om{w[0], w[1], w[2]};
        al = F3{0.f, 0.f, 0.f} - cross(om, ta);
        bv = om;
        g0 = m0 * al + cross(om, m1);
        // axial([om]x M) = tr(M) om - M om, for the point and the direction second moments
        const float t2 = (sb[4] + sb[7] + sb[9]) + (sb[10] + sb[13] + sb[15]);
        const F3 Mo{
            (sb[4] + sb[10]) * om.x + (sb[5] + sb[11]) * om.y + (sb[6] + sb[12]) * om.z,
            (sb[5] + sb[11]) * om.x + (sb[7] + sb[13]) * om.y + (sb[8] + sb[14]) * om.z,
            (sb[6] + sb[12]) * om.x + (sb[8] + sb[14]) * om.y + (sb[9] + sb[15]) * om.z};
        ax = cross(m1, al) + (t2 * om - Mo);
        tr = dot(al, m1);
      } else {
        al = F3{0.f, 0.f, 0.f} - kLn2 * ta;
        bs = kLn2;
        g0 = m0 * al + kLn2 * m1;
        ax = cross(m1, al);
        tr = dot(al, m1) + kLn2 * (sb[4] + sb[7] + sb[9]);
      }
      const float w = s.mW[e]; // pad slots: 0
      float* d = srcD + e;
      d[0] = w * g0.x, d[sst] = w * g0.y, d[2 * sst] = w * g0.z;
      d[3 * sst] = w * ax.x, d[4 * sst] = w * ax.y, d[5 * sst] = w * ax.z;
      d[6 * sst] = w * tr;
      float* o = srcA + e;
      o[0] = w * al.x, o[sst] = w * al.y, o[2 * sst] = w * al.z;
      o[3 * sst] = w * bv.x, o[4 * sst] = w * bv.y, o[5 * sst] = w * bv.z;
      o[6 * sst] = w * bs;
      if (!kMix) {
        srcG[e] = w * sourceGradient(cs.joint, cs.dof, cs.parent, s.js, s.sub1 + kC1 * cs.tin);
      }
    }
    __syncthreads();
    MMX_CLK(4)
    // ================= F: g = J^T r (compacted), padded with zeros  (kMix: done above, in double)
    for (int c = tid; c < (kMix ? 0 : NP); c += 256) {
      float acc = srcG[c]; // pad columns: weight 0
      const int e1 = NP + s.mStart[c + 1];
      for (int e = NP + s.mStart[c]; e < e1; ++e) {
        acc += srcG[e];
      }
      if (hasParamRows && c < n) {
        const ParamCol pc = paramRowsColumn(rig, pb, fd, s.th, nullptr, lColToSolve, P, b, c, lSolveList[c]);
        acc += pc.g;
        s.rho[c] = pc.h; // parked until the tiles of H exist (phase G)
      }
      if (kGen && c < n) {
        const int gst = srcStrideFor(NP);
        for (int r = 0; r < fd.genRows; ++r) {
          acc += s.gJ[r * gst + c] * s.gRes[r];
        }
      }
      s.g[c] = acc;
      s.d0[c] = acc;
    }
    if (kMix && hasParamRows) { // the parameter-space rows' share of g in double; their diagonal parked (rounded) until the tiles of H exist
      for (int c = tid; c < n; c += 256) {
        const ParamColD pc = paramRowsColumnD<false>(rig, pb, fd, m.th, nullptr, lColToSolve, P, b, c, lSolveList[c]);
        m.g[c] += pc.g;
        s.rho[c] = float(pc.h);
      }
    }
    MMX_CLK(5)
    // ================= G: H = J^T J from the moment contractions.  Entry (r, c) of the primary slots is
    //   w_r w_c (G0.AL + AX.BV + TR BS)(deep, ancestor),   deep = the slot whose joint lies below the other's,
    // i.e. one of the two 7-term products D_r . A_c or A_r . D_c, selected by the DFS intervals -- so every
    // 16 x 16 tile is two matrix-core products (K = 7 padded to 8: two v_mfma_f32_16x16x4_f32 each) of the
    // channel-major slot tables, masked, and stored straight into the LDS tile (O(1) work per entry,
    // independent of the number of constraint rows; the scratch region became free at the barrier after
    // E).  The pairs that involve an extra source of a multi-source column are added afterwards from
    // host-built term records.
    {
      const int i = lane & 15, gq = lane >> 4;
      const int k1 = gq < 3 ? 4 + gq : 6; // second MFMA: channel 4 + gq; channel 7 is zero (clamped address, value dropped)
      // operands of a tile: rows of block I on the A side, rows of block Jc on the B side
      struct TileOps {
        float dI0, aI0, dJ0, aJ0, dI1, aI1, dJ1, aJ1;
        int spanC, spanR[4];
      };
      auto loadOps = [&](int I, int Jc) {
        TileOps o;
        const int ri = 16 * I + i, ci = 16 * Jc + i;
        o.dI0 = srcD[gq * sst + ri], o.aI0 = srcA[gq * sst + ri];
        o.dJ0 = srcD[gq * sst + ci], o.aJ0 = srcA[gq * sst + ci];
        o.dI1 = srcD[k1 * sst + ri], o.aI1 = srcA[k1 * sst + ri];
        o.dJ1 = srcD[k1 * sst + ci], o.aJ1 = srcA[k1 * sst + ci];
        o.spanC = s.mTin[ci];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o.spanR[q] = s.mTin[16 * I + 4 * gq + q];
        }
        return o;
      };
      // the wave's tiles t = wave, wave + 4, ... in (I, Jc) form, advanced without a square root
      int I = 0, Jc = wave;
      auto normalise = [&]() {
        while (Jc > I) {
          Jc -= I + 1;
          ++I;
        }
      };
      normalise();
      TileOps cur = loadOps(I < NB ? I : 0, I < NB ? Jc : 0);
      for (int t = wave; t < T; t += 4) {
        const int tI = I, tJ = Jc;
        Jc += 4;
        normalise();
        const bool more = t + 4 < T; // wave-uniform
        // the next tile's LDS reads fly while this one multiplies -- except at four workgroups per CU, where the thirteen registers
        // of a second operand set cost more than the reads' latency (other workgroups fill it): + 0.4 % (r05_exp_fused.txt)
        TileOps nxt;
        if (!kLeanOps) {
          nxt = loadOps(more ? I : tI, more ? Jc : tJ);
        }
        const float z = gq == 3 ? 0.f : 1.f;
        v4f P{0.f, 0.f, 0.f, 0.f}, Q{0.f, 0.f, 0.f, 0.f};
        P = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.dI0, cur.aJ0, P, 0, 0, 0); // P[r][c] = D_r . A_c : row slot deep
        Q = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.aI0, cur.dJ0, Q, 0, 0, 0); // Q[r][c] = A_r . D_c : column slot deep
        P = __builtin_amdgcn_mfma_f32_16x16x4f32(z * cur.dI1, cur.aJ1, P, 0, 0, 0);
        Q = __builtin_amdgcn_mfma_f32_16x16x4f32(z * cur.aI1, cur.dJ1, Q, 0, 0, 0);
        const int tinC = cur.spanC & 0xffff, toutC = cur.spanC >> 16;
        float* Tc = s.L + 256 * tileIndex(tI, tJ);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int tinR = cur.spanR[q] & 0xffff, toutR = cur.spanR[q] >> 16;
          const bool rowDeep = tinC <= tinR && tinR < toutC; // the column's joint is the row's joint or above it
          const bool colDeep = tinR <= tinC && tinC < toutR;
          Tc[tileAddr(4 * gq + q, i)] = rowDeep ? P[q] : (colDeep ? Q[q] : 0.f);
        }
        if (kLeanOps) {
          nxt = loadOps(more ? I : tI, more ? Jc : tJ);
        }
        cur = nxt;
      }
    }
    __syncthreads();
    MMX_CLK(12)
    if (fd.termRounds > 0) {
      float h = 0.f;
      constexpr int kRecTrip = 8;
      for (int k0 = 0; k0 < fd.termRounds; k0 += kRecTrip) {
        uint2 rec[kRecTrip];
#pragma unroll
        for (int k = 0; k < kRecTrip; ++k) {
          const uint4* rp = asGlobal(fd.gTerms) + (k0 + k) * 256 + tid;
          rec[k] = *reinterpret_cast<const uint2*>(rp); // x: deep | anc << 12 | flags ; y: destination
        }
        if (MODE == 2) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          MMX_CLK(21)
        }
#pragma unroll
        for (int k = 0; k < kRecTrip; ++k) {
          const uint32_t x = rec[k].x;
          if (x & (1u << 26)) {
            const int deep = x & 0xfff, anc = (x >> 12) & 0xfff;
            float dv[7], av[7]; // all fourteen reads in flight together: one LDS round trip per record
#pragma unroll
            for (int ch = 0; ch < 7; ++ch) {
              dv[ch] = srcD[ch * sst + deep];
              av[ch] = srcA[ch * sst + anc];
            }
            float hj = 0.f;
#pragma unroll
            for (int ch = 0; ch < 7; ++ch) {
              hj += dv[ch] * av[ch];
            }
            h = (x & (1u << 24)) ? hj : h + hj;
            if (x & (1u << 25)) {
              const uint32_t y = rec[k].y;
              if (y & (1u << 30)) {
                cells[y & 0xffff] = h; // partial cell of a split entry (in uy's place; the gradient shares are consumed)
              } else {
                s.L[y] += h; // one thread per entry
              }
            }
          }
        }
        MMX_CLK(11)
      }
      __syncthreads();
    }
    if (fd.numComb > 0) { // entries that were split into chunks: add the partial cells, fixed order
      for (int i = tid; i < fd.numComb; i += 256) {
        const int dest = fd.comb[3 * i], first = fd.comb[3 * i + 1], cnt = fd.comb[3 * i + 2];
        float v = s.L[dest];
        for (int c = 0; c < cnt; ++c) {
          v += cells[first + c];
        }
        s.L[dest] = v;
      }
      __syncthreads();
    }
    if (hasParamRows) { // J^T J of the parameter-space rows: diagonal entries, then the shared off-diagonal ones
      for (int c = tid; c < n; c += 256) {
        s.L[256 * tileIndex(c >> 4, c >> 4) + tileAddr(c & 15, c & 15)] += s.rho[c];
      }
      if (pb.wLimit > 0.f) {
        const float tWeight = 1e+1f * pb.wLimit;
        for (int d = tid; d < fd.numPairDests; d += 256) {
          float accp = 0.f;
          const int k1 = fd.pairStart[d + 1];
          for (int k = fd.pairStart[d]; k < k1; ++k) {
            float ca = 0.f, cb = 0.f; // the row's entries in the two columns of this H entry
            if (kMix) { // (theta lives in double there; H is the preconditioner: rounded)
              const LimitRowT<double> row = evalLimit<double>(rig, pb.limits[fd.pairLim[k]], m.th, pb.enabledMask, double(tWeight));
#pragma unroll
              for (int e = 0; e < kLimitEntries; ++e) {
                const int sc = row.idx[e] >= 0 ? lColToSolve[row.idx[e]] : -1;
                ca += sc == fd.pairCols[2 * d] ? float(row.coef[e]) : 0.f;
                cb += sc == fd.pairCols[2 * d + 1] ? float(row.coef[e]) : 0.f;
              }
            } else {
              const LimitRow row = evalLimit(rig, pb.limits[fd.pairLim[k]], s.th, pb.enabledMask, tWeight);
#pragma unroll
              for (int e = 0; e < kLimitEntries; ++e) {
                const int sc = row.idx[e] >= 0 ? lColToSolve[row.idx[e]] : -1;
                ca += sc == fd.pairCols[2 * d] ? row.coef[e] : 0.f;
                cb += sc == fd.pairCols[2 * d + 1] ? row.coef[e] : 0.f;
              }
            }
            accp += ca * cb;
          }
          s.L[fd.pairDest[d]] += accp;
        }
      }
      __syncthreads();
    }
    if (kGen) { // + J_g^T J_g: a rank-genRows update of every tile on the matrix cores (operands straight from J_g)
      const int i = lane & 15, gq = lane >> 4;
      const int gst = srcStrideFor(NP), steps = (fd.genRows + 3) >> 2;
      for (int t = wave; t < T; t += 4) {
        int I, Jc;
        tileDecode(t, I, Jc);
        v4f acc{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < steps; ++k) {
          const float* rowp = s.gJ + (4 * k + gq) * gst;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rowp[16 * I + i], rowp[16 * Jc + i], acc, 0, 0, 0);
        }
        float* Tc = s.L + 256 * t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Tc[tileAddr(4 * gq + q, i)] += acc[q];
        }
      }
      __syncthreads();
    }
    if (MODE == 1 && it == 0) { // parity hook: H = J^T J without lambda, both triangles
      for (int e = tid; e < n * n; e += 256) {
        const int row = e / n, col = e - row * n;
        const int hi = row > col ? row : col, lo = row > col ? col : row;
        dbgH[size_t(b) * n * n + e] = s.L[256 * tileIndex(hi >> 4, lo >> 4) + tileAddr(hi & 15, lo & 15)];
      }
      __syncthreads();
    }
    // diagonal: + lambda (gauss_newton_solver.cpp:248); padded rows/cols form an identity block.  The FACTOR is damped
    // by at least kFactorDamping of the mean diagonal (mmx_device.hpp): weak damping -- the trust region starts from
    // (almost) none -- is something the reference's QR of J can take and an fp32 Cholesky of J^T J cannot when J is rank
    // deficient; the refinement (which measures its residual with the true mu through J) moves the step towards the
    // weakly damped one wherever J determines it.
    float muFactor = mu;
    { // (every wave sums the trace for itself: no barrier, no LDS round trip)
      float tr = 0.f;
      for (int c = tid & 63; c < n; c += 64) {
        tr += s.L[256 * tileIndex(c >> 4, c >> 4) + tileAddr(c & 15, c & 15)];
      }
      muFactor = fmaxf(mu, (kMix ? kMixFactorDamping : kFactorDamping) * waveReduceSumF(tr) / float(n > 0 ? n : 1));
      floored = floored || muFactor > mu;
    }
    // Every wave has summed the UNDAMPED diagonal before any thread damps its entry in place (round 5: without this barrier a
    // fast wave's writes below raced a slow wave's trace reads above -- harmless while lambda exceeds the floor, since
    // muFactor is then lambda whatever the trace, but under weak damping the waves could factor with floors that differed in
    // the last bits from run to run: scripts/diag_determinism.py found 586 of 11 264 instance-solves of BASELINE configs[1]'s
    // shape at lambda = 1e-7 not bit-reproducible, none from lambda = 1e-3 up)
    __syncthreads();
    for (int c = tid; c < NP; c += 256) {
      float* dg = s.L + 256 * tileIndex(c >> 4, c >> 4) + tileAddr(c & 15, c & 15);
      const float hd = c < n ? *dg + muFactor : 1.f;
      *dg = hd;
      s.invDiag[c] = kPivotFloor * hd; // the row's pivot floor until the panel pass of its block leaves 1 / l_cc here
    }
    if (MODE == 1 && it == 0) {
      for (int c = tid; c < n; c += 256) {
        dbgG[size_t(b) * n + c] = s.g[c];
      }
    }
    if (tid == 0) {
      s.flags[1] = 0;
    }
    __syncthreads(); // srcT / moments are dead from here on: the region becomes the factor
    MMX_CLK(6)

    // ================= H: blocked left-looking Cholesky on the LDS-resident tiles
    for (int k = 0; k < NB; ++k) {
      // (u) bring block column k up to date: tile(I,k) -= sum_{j<k} L(I,j) L(k,j)^T.  A tile is
      //     read once, takes all its 4 k MFMAs (two accumulators: even / odd j) and is written once.
      // tile (I, kc) -= sum_{j0 <= j < j1} L(I,j) L(kc,j)^T
      auto updateTile = [&](int I, int kc, int j0, int j1) {
        float* Tc = s.L + 256 * tileIndex(I, kc);
        v4f c0, c1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          c0[r] = Tc[tileAddr(4 * (lane >> 4) + r, lane & 15)];
        }
        for (int j = j0; j < j1; ++j) {
          const float4 av = ldsRow4(s.L + 256 * tileIndex(I, j), lane & 15, lane >> 4);
          const float4 bv = ldsRow4(s.L + 256 * tileIndex(kc, j), lane & 15, lane >> 4);
          if ((j & 1) && !kLeanAcc) { // (one accumulator at four workgroups per CU: four registers less, + 1.5 %; r05_exp_fused.txt)
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.x, bv.x, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.y, bv.y, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.z, bv.z, c1, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.w, bv.w, c1, 0, 0, 0);
          } else {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.x, bv.x, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.y, bv.y, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.z, bv.z, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.w, bv.w, c0, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Tc[tileAddr(4 * (lane >> 4) + r, lane & 15)] = c0[r] + c1[r];
        }
      };
      if (k > 0) {
        // (lookahead: the contributions of the block columns before k - 1 were taken during panel k - 1 by the
        // waves that had no panel row, see below -- only column k - 1's is left)
        const int jFirst = (kLook && k >= 2) ? k - 1 : 0;
        for (int I = k + wave; I < NB; I += 4) {
          updateTile(I, k, jFirst, k);
        }
        __syncthreads();
      }
      MMX_CLK(14)
      // (b+c) panel factorisation: every wave holds the 16 rows of the diagonal block in lanes
      //     0..15 (redundantly) and 48 rows of the panel below it in lanes 16..63, one row of 16
      //     values per lane.  Sixteen elimination steps factor the diagonal block AND solve the
      //     panel rows against it at the same time; the pivot row's entries are broadcast with
      //     v_readlane, so there is no LDS round trip and no barrier inside the dependent chain.
      {
        float* Dk = s.L + 256 * tileIndex(k, k);
        // lanes 16..63 of the four waves enumerate "virtual" panel rows: the first 16 are rows of
        // the identity, which the elimination turns into L_kk^-T (used by solveLLt), then come
        // the real rows below the diagonal block
        const bool diagLane = lane < 16;
        const int vrow = 48 * wave + (lane - 16);
        const bool identLane = !diagLane && vrow < 16;
        const int prow = 16 * k + vrow; // = 16 (k + 1) + (vrow - 16)
        const bool panelLane = !diagLane && !identLane && prow < NP;
        // waves whose 48 virtual rows all lie beyond the matrix only wait (wave-uniform branch)
        const bool waveWorks = wave == 0 || 16 * k + 48 * wave < NP;
        if (kRide && wave == 3 && k >= 1) {
          // (round 6) the first solve's forward substitution rides along: block k - 1 of y = L^-1 g needs the block rows 0 .. k - 1
          // of the factor, final since the barrier after panel k - 1 -- wave 3 never holds a panel row up to six blocks
          solveForwardBlockAhead<NB>(s.L, s.invDiag, s.d0, k - 1, lane);
        }
        if (kLook && !waveWorks && k >= 1 && k + 1 < NB) {
          // lookahead: the waves without a panel row bring block column k + 1 up to date with the finished columns
          // j < k while the others run the elimination chain (disjoint tiles: column k is the chain's)
          const int firstIdle = (NP - 16 * k + 47) / 48; // waves 0 .. firstIdle - 1 hold the virtual rows
          for (int I = k + 1 + (wave - firstIdle); I < NB; I += 4 - firstIdle) {
            updateTile(I, k + 1, 0, k);
          }
        }
        float* Tl = panelLane ? s.L + 256 * tileIndex(prow >> 4, k) : Dk;
        const int trow = diagLane ? lane : (panelLane ? (prow & 15) : vrow);
        float a[16];
        auto loadRows = [&]() {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = (diagLane || panelLane) ? ldsRow4(Tl, trow, q) : float4{0.f, 0.f, 0.f, 0.f};
            a[4 * q] = v.x, a[4 * q + 1] = v.y, a[4 * q + 2] = v.z, a[4 * q + 3] = v.w;
          }
          if (identLane) {
            // (the opaque copy keeps the compiler from hoisting these sixteen per-lane constants out
            // of the iteration loop, where they ended up in scratch and were reloaded for every panel)
            int vr = vrow;
            asm volatile("" : "+v"(vr));
#pragma unroll
            for (int c = 0; c < 16; ++c) {
              a[c] = c == vr ? 1.f : 0.f;
            }
          }
        };
        if (waveWorks) {
          loadRows();
        }
        MMX_CLK(22)
        float invd = 0.f;
        bool bad = false;
        if (waveWorks) {
          // the sixteen elimination steps; kGuard: a pivot at rounding level drops its column from this iteration's step
          // (see kPivotFloor): 1 / l_jj = 0
          auto chain = [&](const bool kGuard, const float floorRow) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const float djj = readLaneF(a[j], j);
              bad = bad || !(djj > 0.f);
              float inv = __builtin_amdgcn_rsqf(djj); // 1 / l_jj ; l_jj = d_jj * inv
              if (kGuard) {
                // (kMix: what is factored is only the PRECONDITIONER of the conjugate gradients: a pivot at rounding level is
                // FLOORED, not dropped -- a dropped column would take its parameter out of the Krylov space, and the noise a
                // floored pivot lets through is the iteration's to remove)
                const float fl = readLaneF(floorRow, j);
                inv = djj > fl ? inv : (kMix ? __builtin_amdgcn_rsqf(fl) : 0.f);
              }
              a[j] *= inv;
              if (lane == j) {
                invd = inv;
              }
              panelRowUpdate(a, j);
            }
          };
          // The threshold costs three instructions per step on the kernel's longest dependent chain (measured: 3 % of the
          // headline), and it only ever acts on rank-deficient systems with a lambda below the rounding of H: so the chain
          // runs unguarded, every diagonal lane then checks ITS pivot from the 1 / l_jj it kept (d_jj = 1 / invd^2; a
          // non-positive pivot left a NaN there), and only a panel that fails is reloaded (its tiles are still untouched
          // in LDS) and eliminated again with the guard.  Every wave factors the diagonal block redundantly from the same
          // values, so all of them take the same branch.
          chain(false, 0.f);
          const float floorRow = s.invDiag[16 * k + (lane & 15)]; // kPivotFloor * (H_rr + lambda) of the diagonal lanes' rows
          // the precision estimate's input: the smallest d_jj / (H_jj + mu) of the solve, kept per lane as the largest
          // kPivotFloor (H_jj + mu) / d_jj (one instruction off the chain; the lanes are joined once, after the last
          // iteration).  Every wave factors the diagonal block redundantly: wave 0's lanes 0..15 are the ones read.
          pivotWorst = floorRow * invd * invd > pivotWorst ? floorRow * invd * invd : pivotWorst; // (a NaN pivot keeps the old value: MMX_SOLVE_NOT_PD reports it)
          if (__any(diagLane && !(floorRow * invd * invd < 1.f))) {
            loadRows();
            chain(true, floorRow);
          }
        }
        __syncthreads(); // every wave has read the diagonal block (long ago) before wave 0 overwrites it
        if (waveWorks) {
          MMX_CLK(23)
          // Stores.  The diagonal tile takes L_kk (columns <= row) from the diagonal lane of a row and
          // L_kk^-T (columns > row) from the identity lane of the same row, 16 lanes further up:
          // ds_swizzle (xor 16) brings the partner's value over, so that the diagonal lane writes the
          // whole row and every row leaves as four unpredicated 16-byte stores.
          const bool stores = panelLane || (diagLane && wave == 0);
          float v[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const float partner = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[c]), 0x401F));
            v[c] = (diagLane && c > lane) ? partner : a[c];
          }
          if (stores) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              *reinterpret_cast<float4*>(Tl + trow * 16 + (((q ^ (trow >> 2)) & 3) << 2)) =
                  float4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            }
          }
          if (diagLane && wave == 0) {
            s.invDiag[16 * k + lane] = invd;
            if (bad) {
              s.flags[1] = 1;
            }
          }
        }
        __syncthreads();
        // panels taller than the 176 rows of one pass: the remaining rows solve against the finished L_kk
        for (int r = 16 * (k + 1) + 176 + tid; r < NP; r += 256) {
          float* Tr = s.L + 256 * tileIndex(r >> 4, k);
          float x[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = ldsRow4(Tr, r & 15, q);
            x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
          }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            float sum = x[j];
#pragma unroll
            for (int c = 0; c < j; ++c) {
              sum -= x[c] * Dk[tileAddr(j, c)];
            }
            x[j] = sum * s.invDiag[16 * k + j];
          }
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            Tr[tileAddr(r & 15, c)] = x[c];
          }
        }
        if (NP - 16 * (k + 1) > 176) {
          __syncthreads();
        }
      }
      MMX_CLK(13)
    }
    __syncthreads();
    badPivot = s.flags[1] != 0;
    MMX_CLK(7)

    // ================= I: d0 = (L L^T)^-1 g
    if (kMix) {
      // ---- I + J in mixed precision: (J^T S^2 J + mu I) x = g by conjugate gradients in double, preconditioned by the
      // single-precision factor (z = (L L^T)^-1 r, rounded through float on the way in and out); the operator goes through the
      // tree in double (mixApply), never through the rounded H.  With a good preconditioner (mu well above the rounding of H:
      // BASELINE's damping) the first direction is the single-precision step itself and the loop ends after ONE operator
      // application with x = alpha z_0 + z_1 -- the single-precision solve's refinement step with an exact residual; where the
      // factor is a poor preconditioner (damping floor above mu, cond x eps ~ 1e-2) the iteration continues as flexible
      // (Polak-Ribiere) CG.  Stop: the predicted size of the NEXT correction, |z|^2 / |last step| x |z|, below mixTol |x|.
      for (int c = tid; c < NP; c += 256) {
        const double gc = c < n ? m.g[c] : 0.0;
        m.x[c] = 0.0;
        m.r[c] = gc;
        s.rho[c] = float(gc);
      }
      __syncthreads();
      solveLLt<NB, false>(s.L, s.invDiag, s.rho, tid);
      double rz = 0.0;
      for (int c = tid; c < n; c += 256) {
        const double z = double(s.rho[c]);
        m.p[c] = z;
        rz += m.r[c] * z;
      }
      rz = blockSumD(s.red, rz, tid);
      MMX_CLK(16)
      const double tol2 = double(fp.mixTol) * double(fp.mixTol);
      bool converged = !(rz > 0.0); // (g = 0: the step is zero)
      for (int k = 0; k < fp.mixMaxCg && !converged; ++k) {
        mixApply(rv, fv, s, m, lParentPos, J, NP, n, nsrc, double(mu), m.p, m.q, tid, mixRecP);
        if (hasParamRows) { // + J_p^T J_p p of the parameter-space rows (the thread that wrote q[c])
          for (int c = tid; c < n; c += 256) {
            m.q[c] += paramRowsColumnD<true>(rig, pb, fd, m.th, m.p, lColToSolve, P, b, c, lSolveList[c]).g;
          }
          __syncthreads();
        }
        ++mixApplied;
        MMX_CLK(17)
        double pq = 0.0;
        for (int c = tid; c < n; c += 256) {
          pq += m.p[c] * m.q[c];
        }
        pq = blockSumD(s.red, pq, tid);
        if (!(pq > 0.0)) {
          break;
        }
        const double alpha = rz / pq;
        double sums[5] = {0.0, 0.0, 0.0, 0.0, 0.0}; // |alpha p|^2, |x|^2, r.z, z.q, |z|^2
        for (int c = tid; c < NP; c += 256) {
          float rc = 0.f;
          if (c < n) {
            const double dx = alpha * m.p[c], xn = m.x[c] + dx, rn = m.r[c] - alpha * m.q[c];
            m.x[c] = xn, m.r[c] = rn;
            sums[0] += dx * dx, sums[1] += xn * xn;
            rc = float(rn);
          }
          s.rho[c] = rc;
        }
        __syncthreads();
        MMX_CLK(19)
        solveLLt<NB, false>(s.L, s.invDiag, s.rho, tid);
        MMX_CLK(20)
        for (int c = tid; c < n; c += 256) {
          const double z = double(s.rho[c]);
          sums[2] += m.r[c] * z, sums[3] += z * m.q[c], sums[4] += z * z;
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          sums[q] = waveReduceSum(sums[q]);
        }
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            s.red[4 * q + wave] = sums[q];
          }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; ++q) {
          sums[q] = (s.red[4 * q] + s.red[4 * q + 1]) + (s.red[4 * q + 2] + s.red[4 * q + 3]);
        }
        __syncthreads();
        // the correction z (x alpha when the factor over-damps the directions the iteration is moving in: alpha > 1) would be
        // followed by one of about |z| x (|z| / |last step|): small enough => take z and stop
        const double am2 = alpha > 1.0 ? alpha * alpha : 1.0;
        if (am2 * sums[4] * am2 * sums[4] <= tol2 * sums[1] * sums[0]) {
          for (int c = tid; c < n; c += 256) {
            m.x[c] += double(s.rho[c]);
          }
          converged = true;
        } else {
          const double beta = -alpha * sums[3] / rz; // Polak-Ribiere: z'.(r' - r) / (z.r), r' - r = -alpha q
          for (int c = tid; c < n; c += 256) {
            m.p[c] = double(s.rho[c]) + beta * m.p[c];
          }
          rz = sums[2];
          converged = !(rz > 0.0);
          // no prospect of meeting the tolerance within the step limit (six applications in and the next correction still above
          // 1e-3 of the step: cond x eps ~ 1, the factor is no preconditioner): stop here, the element is reported
          if (k >= 5 && am2 * sums[4] > 1e-6 * sums[1]) {
            __syncthreads();
            break;
          }
        }
        __syncthreads();
        MMX_CLK(18)
      }
      mixUnconverged = mixUnconverged || !converged;
    } else if (kRide) { // (blocks 0 .. NB - 2 of the forward substitution were solved under the panels: the last block and the backward sweep)
      solveLLtAheadTail<NB>(s.L, s.invDiag, s.d0, tid);
    } else if (!notPd) {
      solveLLt<NB, kLeanSolve>(s.L, s.invDiag, s.d0, tid);
    }
    MMX_CLK(8)
    // ================= J: one refinement step through the tree (tangent + adjoint passes)
    // One step is enough when lambda dominates the rounding of H (the BASELINE metric); with a
    // small lambda, heavy weights or a robust loss (rows scaled by 1/c^2) the fp32 factor is a
    // weaker preconditioner, so a further step is taken while the last correction was still
    // larger than 1e-3 of the step (error after k steps ~ ratio^(k+1)).  The test is on block-wide
    // sums, hence uniform.
    // (refinement steps allowed: default 3, mmx_tuning::max_refinement_steps.  Round 6 measured leaving the refinement out of the
    // FIRST three / five iterations only -- + 5.6 % / + 10 % -- and every instance left the bound, median 3.4e-5: ten damped
    // iterations do not attenuate an early step's error at all; profiles/r06_exp_fused.txt)
    const int nRefine = (!notPd && !kMix) ? fp.refine : 0;
    float prevCorr2 = FLT_MAX;
    for (int rf = 0; rf < nRefine; ++rf) {
      // joint-parameter delta jd = transform * delta (delta gathered through the solve map)
      {
        auto xd = [&](int c) {
          const int cs = fv.colToSolve[c];
          return cs >= 0 ? s.d0[cs] : 0.f;
        };
        auto od = [&](int r, float a) { s.jd[r] = a; };
        if (kCsrLds) { // one transform row per thread from the LDS copy
          for (int r = tid; r < rv.R; r += 256) {
            float a = 0.f;
            const int k1 = rv.ptOuter[r + 1];
            for (int k = rv.ptOuter[r]; k < k1; ++k) {
              a += rv.ptValue[k] * xd(rv.ptInner[k]);
            }
            od(r, a);
          }
        } else if (rv.rowRec != nullptr) {
          csrRowsFromRecords<256>(rv.rowRec, rv.numRowRec, rv.ptInner, rv.ptValue, rv.R, tid, xd, od);
        } else {
          csrRowsPrefetched<256>(rv.ptOuter, rv.ptInner, rv.ptValue, rv.R, tid, xd, od);
        }
      }
      __syncthreads();
      MMX_CLK(16)
      if (kGen) { // w_g = r_g - J_g d (consumed by the rho loop below, several barriers later)
        const int gst = srcStrideFor(NP);
        for (int r = tid; r < fd.genRows; r += 256) {
          float a = s.gRes[r];
          for (int c = 0; c < n; ++c) {
            a -= s.gJ[r * gst + c] * s.d0[c];
          }
          s.gW[r] = a;
        }
      }
      // tangent pass: per joint (stored by DFS position) C = T - Om x t - ln2 sd t, W = Om, S = sd
      for (int k = tid; k < J; k += 256) {
        const int q = fv.dfsJoint[k];
        const float* ja = s.js + kJs * q;
        const float* d = s.jd + 7 * q;
        const F3 ta{ja[0], ja[1], ja[2]};
        F3 Tv{0.f, 0.f, 0.f};
        if (d[0] != 0.f || d[1] != 0.f || d[2] != 0.f) {
          const int par = rv.parent[q];
          Tv = d[0] * transAxisCol(s.js, par, 0) + d[1] * transAxisCol(s.js, par, 1) + d[2] * transAxisCol(s.js, par, 2);
        }
        const F3 Om = d[3] * F3{ja[8], ja[9], ja[10]} + d[4] * F3{ja[11], ja[12], ja[13]} + d[5] * F3{ja[14], ja[15], ja[16]};
        const F3 C = Tv - cross(Om, ta) - (kLn2 * d[6]) * ta;
        float* o = s.tanOwn + kTan * k;
        o[0] = C.x, o[1] = C.y, o[2] = C.z, o[3] = Om.x, o[4] = Om.y, o[5] = Om.z, o[6] = d[6], o[7] = 0.f;
      }
      __syncthreads();
      // ... summed over each joint's ancestor chain (prefix sums down the tree)
      if (J <= 256) {
        // pointer jumping with the running sum and the jump target in registers: per round a thread
        // reads its target's sum and target (slot 7 of the row), then every thread publishes its own
        float acc[7];
        int target = -1;
        if (tid < J) {
#pragma unroll
          for (int c = 0; c < 7; ++c) {
            acc[c] = s.tanOwn[kTan * tid + c];
          }
          target = lParentPos[tid];
          s.tanPre[kTan * tid + 7] = __int_as_float(target);
#pragma unroll
          for (int c = 0; c < 7; ++c) {
            s.tanPre[kTan * tid + c] = acc[c];
          }
        }
        __syncthreads();
        for (int r = 0; r < rv.jumpRounds; ++r) {
          int next = -1;
          if (target >= 0) {
            const float* t = s.tanPre + kTan * target;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
              acc[c] += t[c];
            }
            next = __float_as_int(t[7]);
          }
          __syncthreads();
          if (target >= 0) {
            float* o = s.tanPre + kTan * tid;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
              o[c] = acc[c];
            }
            o[7] = __int_as_float(next);
          }
          target = next;
          __syncthreads();
        }
      } else {
        treeSum<7, false, kTan>(fv, s.tanOwn, s.tanPre, J, wave, lane);
        __syncthreads();
      }
      // w = r - J d, y = sigma w per unit, then the first-order own sums.  The arena's two halves take turns (each buffer
      // over a dead predecessor): with U <= J the per-unit contributions go where tanOwn was, their per-joint sums where
      // tanPre was, the subtree sums over the contributions, the per-slot gradients over the own sums; with U > J the own
      // sums are formed directly where tanOwn was and the two halves swap roles.
      const bool unitPath = U <= J;
      float* refOwn = unitPath ? arena + lay.t9 : arena;
      float* refSub = unitPath ? arena : arena + lay.t9;
      FusedLds sr = s;
      sr.own1 = refOwn;
      if (unitPath) {
        for (int u = tid; u < U; u += 256) {
          const float* pre = s.tanPre + kTan * fv.unitPos[u];
          const F3 p{s.up[3 * u], s.up[3 * u + 1], s.up[3 * u + 2]};
          const bool point = u < fv.Kp;
          F3 v = cross(F3{pre[3], pre[4], pre[5]}, p);
          if (point) {
            v = F3{pre[0], pre[1], pre[2]} + v + (kLn2 * pre[6]) * p;
          }
          const float sg = s.us[u];
          firstOrderMoments(
              refSub + kC1 * u, p, sg * (s.ur[3 * u] - sg * v.x), sg * (s.ur[3 * u + 1] - sg * v.y), sg * (s.ur[3 * u + 2] - sg * v.z), point);
        }
        __syncthreads();
        gatherOwnSums<kC1, kC1, 256, FusedViewS>(fv, sr, refSub, tid);
      } else {
        for (int k = tid; k < J; k += 256) {
          const int e0 = fv.posUnitStart[k], e1 = fv.posUnitStart[k + 1];
          float a1[kC1] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (e1 > e0) {
            const float* pre = s.tanPre + kTan * k;
            for (int e = e0; e < e1; ++e) {
              const int u = fv.posUnits[e];
              const F3 p{s.up[3 * u], s.up[3 * u + 1], s.up[3 * u + 2]};
              const bool point = u < fv.Kp;
              F3 v = cross(F3{pre[3], pre[4], pre[5]}, p);
              if (point) {
                v = F3{pre[0], pre[1], pre[2]} + v + (kLn2 * pre[6]) * p;
              }
              const float sg = s.us[u];
              float o[kC1];
              firstOrderMoments(o, p, sg * (s.ur[3 * u] - sg * v.x), sg * (s.ur[3 * u + 1] - sg * v.y), sg * (s.ur[3 * u + 2] - sg * v.z), point);
#pragma unroll
              for (int c = 0; c < kC1; ++c) {
                a1[c] += o[c];
              }
            }
          }
#pragma unroll
          for (int c = 0; c < kC1; ++c) {
            refOwn[kC1 * k + c] = a1[c];
          }
        }
      }
      __syncthreads();
      MMX_CLK(17)
      treeSum<kC1, true>(fv, refOwn, refSub, J, wave, lane);
      __syncthreads();
      MMX_CLK(18)
      // J^T w per slot in parallel (where the own sums were: consumed), then per column the sum of its slots
      const bool perSource = nsrc <= kTan * J;
      float* perS = refOwn;
      if (perSource) {
        for (int e = tid; e < nsrc; e += 256) {
          const int info = s.mInfo[e];
          perS[e] = s.mW[e] * sourceGradient(info & 0xfff, (info >> 12) & 7, (info >> 16) - 1, s.js, refSub + kC1 * (s.mTin[e] & 0xffff));
        }
        __syncthreads();
      }
      for (int c = tid; c < NP; c += 256) {
        float a = 0.f;
        if (c < n) {
          auto slotShare = [&](int sl) {
            if (perSource) {
              return perS[sl];
            }
            const int info = s.mInfo[sl];
            return s.mW[sl] * sourceGradient(info & 0xfff, (info >> 12) & 7, (info >> 16) - 1, s.js, refSub + kC1 * (s.mTin[sl] & 0xffff));
          };
          a = slotShare(c); // the primary slot, then the extras
          const int e1 = NP + s.mStart[c + 1];
          for (int e = NP + s.mStart[c]; e < e1; ++e) {
            a += slotShare(e);
          }
          if (hasParamRows) {
            a += paramRowsColumn(rig, pb, fd, s.th, s.d0, lColToSolve, P, b, c, lSolveList[c]).g;
          }
          if (kGen) {
            const int gst = srcStrideFor(NP);
            for (int r = 0; r < fd.genRows; ++r) {
              a += s.gJ[r * gst + c] * s.gW[r];
            }
          }
          a -= mu * s.d0[c];
        }
        s.rho[c] = a;
      }
      __syncthreads();
      MMX_CLK(19)
      solveLLt<NB, kLeanSolve>(s.L, s.invDiag, s.rho, tid);
      MMX_CLK(20)
      float c2 = 0.f, d2 = 0.f;
      for (int c = tid; c < n; c += 256) {
        const float cr = s.rho[c], dn = s.d0[c] + cr;
        s.d0[c] = dn;
        c2 += cr * cr;
        d2 += dn * dn;
      }
      c2 = waveReduceSumF(c2);
      d2 = waveReduceSumF(d2);
      if (lane == 0) {
        s.red[4 + wave] = double(c2);
        s.red[wave] = double(d2); // (the error sums of phase C were consumed long ago)
      }
      __syncthreads();
      const float corr2 = float((s.red[4] + s.red[5]) + (s.red[6] + s.red[7]));
      const float step2 = float((s.red[0] + s.red[1]) + (s.red[2] + s.red[3]));
      refineWorst = fmaxf(refineWorst, corr2 * __builtin_amdgcn_rcpf(fmaxf(step2, 1e-37f))); // (mmx_problem_solve_diagnostics; uniform)
      __syncthreads();
      // a correction is only taken when it is a contraction (kRefineMax2, mmx_kernels.hip: where the fp32 factor is no
      // preconditioner any more -- pivots at rounding level -- the refinement diverges): undo it and stop
      if (corr2 > 0.25f * step2 || corr2 > prevCorr2) {
        for (int c = tid; c < n; c += 256) {
          s.d0[c] -= s.rho[c];
        }
        __syncthreads();
        break;
      }
      prevCorr2 = corr2;
      if (!(corr2 > 1e-6f * step2)) {
        break;
      }
    }
    MMX_CLK(9)
    if (!kTR) {
      break;
    }
    // ---- trust region: is this step the one to try, or does the damping have to grow first?
    if (notPd) { // (cannot happen in the reference's QR) more damping, a bounded number of times
      trLambda = fmaxf(4.f * trLambda, 1e-6f);
      if (++pdRetries > 16) {
        trNoStep = true;
        break;
      }
      continue;
    }
    {
      float p0 = 0.f, p1 = 0.f;
      for (int c = tid; c < n; c += 256) {
        p0 += s.d0[c] * s.d0[c];
        p1 += s.d0[c] * s.g[c];
      }
      trDn2 = blockSumF(s, p0, tid);
      trDg = blockSumF(s, p1, tid);
      trMu = mu;
    }
    if (newtonIter == 0 && 2.f * trDg < FLT_EPSILON * (1.f + float(curError))) { // :164 (gradientSub_ = 2 J^T r)
      trNoStep = true;
      break;
    }
    if (newtonIter < 3 && sqrtf(trDn2) >= 1.05f * trRadius) { // :180-181
      // Newton step on lambda (Nocedal & Wright eq. 4.44, :191-204): p_l = -(step), |q_l|^2 = p_l^T (R^T R)^-1 p_l
      for (int c = tid; c < NP; c += 256) {
        s.rho[c] = c < n ? s.d0[c] : 0.f;
      }
      __syncthreads();
      solveLLt<NB, kLeanSolve>(s.L, s.invDiag, s.rho, tid);
      float pq = 0.f;
      for (int c = tid; c < n; c += 256) {
        pq += s.d0[c] * s.rho[c];
      }
      const float q2 = blockSumF(s, pq, tid);
      if (q2 >= FLT_EPSILON) { // :198
        const float pn = sqrtf(trDn2);
        const float deltaLambda = (trDn2 / q2) * ((pn - trRadius) / trRadius);
        if (deltaLambda > 0.f) { // :207: lambda only ever grows
          trLambda += deltaLambda;
          ++newtonIter;
          continue; // factor and solve again with the larger damping (the reference appends rows to its QR)
        }
      }
    }
    break;
    }
    } // linear solves
    // ================= K: theta -= delta ; bookkeeping of SolverT::solve (solver.cpp:92-119)
    if (kMix) {
      // ---- the step rules of phase K as the DOUBLE instantiation takes them (oracle: solveGaussNewton<double>): theta, the
      // trial parameters (m.Y) and the errors in double; the LM schedule's damping in double, its rounding is what is factored
      if (stepRule == 1) {
        double part = 0.0;
        for (int c = tid; c < n; c += 256) {
          part += m.x[c] * m.g[c] + lambdaD * m.x[c] * m.x[c];
        }
        const double predicted = blockSumD(s.red, part, tid); // decrease of |r - J d|^2 = d.g + lambda d.d
        for (int i = tid; i < P; i += 256) {
          m.Y[i] = m.th[i];
        }
        __syncthreads();
        for (int c = tid; c < n; c += 256) {
          m.Y[fv.solveList[c]] -= m.x[c];
        }
        __syncthreads();
        double eFull = 0.0;
        const double eNew = blockErrorD<true>(rv, pb, s, m, m.Y, b, U, tid, mixLevel, &eFull, mixRecP, &rig, hasParamRows);
        const double rho = predicted > 0.0 ? (curError - eNew) / predicted : -1.0;
        if (st.stepHistory != nullptr && tid == 0) {
          double* sh = st.stepHistory + (size_t(b) * fp.maxIterations + it) * 2;
          sh[0] = lambdaD;
          sh[1] = rho;
        }
        stateValid = false; // (a rejected trial leaves the trial's joint states behind: this theta is evaluated again)
        if (rho > 0.0) {
          for (int i = tid; i < P; i += 256) {
            m.th[i] = m.Y[i];
          }
          stateValid = !hasParamRows; // (the trial's error is getError's: with parameter-space rows the iteration's is getJacobian's)
          stateError = eFull;
        }
        if (!(rho >= 0.25)) {
          lambdaD = fmin(lambdaD * double(fp.lmUp), double(fp.lmLambdaMax));
        } else if (rho > 0.75) {
          lambdaD = fmax(lambdaD * double(fp.lmDown), double(fp.lmLambdaMin));
        }
        lambda = float(lambdaD);
      } else if (doLineSearch) { // both backtracking rules (gauss_newton_solver.cpp:283-313, subset_gauss_newton_solver.cpp:117-142)
        const double scaledError = 1e-3 * curError;
        double gd = 0.0;
        if (doLineSearch == 2) {
          double part = 0.0;
          for (int c = tid; c < n; c += 256) {
            part += m.g[c] * m.x[c];
          }
          gd = blockSumD(s.red, part, tid);
        }
        float scale = 1.f;
        for (int ls = 0; ls < 10; ++ls) {
          for (int i = tid; i < P; i += 256) {
            m.Y[i] = m.th[i];
          }
          __syncthreads();
          for (int c = tid; c < n; c += 256) {
            m.Y[fv.solveList[c]] -= double(scale) * m.x[c];
          }
          __syncthreads();
          const double eNew = blockErrorD<true>(rv, pb, s, m, m.Y, b, U, tid, mixLevel, &stateError, mixRecP, &rig, hasParamRows);
          if ((curError - eNew) >= (doLineSearch == 2 ? double(1e-4f * scale) * gd : double(scale) * scaledError)) {
            break;
          }
          scale *= 0.5f;
        }
        for (int i = tid; i < P; i += 256) {
          m.th[i] = m.Y[i];
        }
        stateValid = !hasParamRows; // the last trial evaluated IS the new theta
      } else {
        for (int c = tid; c < n; c += 256) {
          m.th[fv.solveList[c]] -= m.x[c]; // skeleton_solver_function.cpp:158
        }
        stateValid = false;
      }
      break;
    }
    if (kTR) {
      if (trNoStep) {
        break;
      }
      // trial step, gain ratio against the quadratic model e - 2 g.p + p^T (J^T J + 1e-20 I) p, where
      // p^T J^T J p = g.p - mu |p|^2 because (J^T J + mu I) p = g  (:240-247)
      for (int i = tid; i < P; i += 256) {
        s.dfull[i] = s.th[i];
      }
      __syncthreads();
      for (int c = tid; c < n; c += 256) {
        s.dfull[fv.solveList[c]] -= s.d0[c];
      }
      __syncthreads();
      const double eNew = blockError<kGen, false, !kCsrLds>(rig, rv, pb, fv, s, s.dfull, b, tid);
      const float predicted = trDg + (trMu - 1e-20f) * trDn2; // e - model
      const float rho = float((curError - eNew) / double(predicted));
      if (rho < 0.25f) { // :256-262 (lambda > 0 always holds: it starts at 1e-10)
        trRadius = 0.25f * trRadius;
      } else if (rho > 0.75f) {
        trRadius = fminf(2.f * trRadius, 10.f);
      }
      if (rho > 0.f) { // :265
        for (int i = tid; i < P; i += 256) {
          s.th[i] = s.dfull[i];
        }
        break;
      }
      if (++trustStep >= 10) { // :157: every trial rejected, the parameters stay (:268-269)
        break;
      }
      __syncthreads();
      continue; // the trial evaluation overwrote the joint states and the factor: this theta once more
    }
    if (!notPd && stepRule == 1) {
      // ---- LM gain-ratio schedule, the lambda form of TrustRegionQRT's radius rule
      // (momentum/character_solver/trust_region_qr.cpp:244-268); identical to the oracle's
      // restatement (oracle/mmx_oracle.hpp solveGaussNewton, stepRule 1)
      float part = 0.f;
      for (int c = tid; c < n; c += 256) {
        part += s.d0[c] * s.g[c] + lambda * s.d0[c] * s.d0[c];
      }
      const float predicted = blockSumF(s, part, tid); // decrease of |r - J d|^2 = d.g + lambda d.d
      for (int i = tid; i < P; i += 256) {
        s.dfull[i] = s.th[i];
      }
      __syncthreads();
      for (int c = tid; c < n; c += 256) {
        s.dfull[fv.solveList[c]] -= s.d0[c];
      }
      __syncthreads();
      double eFull = 0.0;
      const double eNew = blockError<kGen, kReuse, !kCsrLds>(rig, rv, pb, fv, s, s.dfull, b, tid, &eFull);
      const float rho = predicted > 0.f ? float((curError - eNew) / double(predicted)) : -1.f;
      if (st.stepHistory != nullptr && tid == 0) {
        double* sh = st.stepHistory + (size_t(b) * fp.maxIterations + it) * 2;
        sh[0] = double(lambda);
        sh[1] = double(rho);
      }
      stateValid = false; // (a rejected trial leaves the trial's joint states behind: this theta is evaluated again)
      if (rho > 0.f) {
        for (int i = tid; i < P; i += 256) {
          s.th[i] = s.dfull[i];
        }
        stateValid = kReuse && !hasParamRows;
        stateError = eFull;
      }
      if (!(rho >= 0.25f)) {
        lambda = fminf(lambda * fp.lmUp, fp.lmLambdaMax);
      } else if (rho > 0.75f) {
        lambda = fmaxf(lambda * fp.lmDown, fp.lmLambdaMin);
      }
    } else if (notPd && stepRule == 1) {
      if (st.stepHistory != nullptr && tid == 0) {
        double* sh = st.stepHistory + (size_t(b) * fp.maxIterations + it) * 2;
        sh[0] = double(lambda);
        sh[1] = -1.0; // no step was taken: the schedule treats it as a rejected one
      }
      lambda = fminf(lambda * fp.lmUp, fp.lmLambdaMax);
    } else if (!notPd && doLineSearch) {
      // ---- GaussNewtonSolverT::updateParameters with doLineSearch (gauss_newton_solver.cpp:283-313):
      // Armijo backtracking, c1 = 1e-3, tau = 0.5, at most 10 trial steps; the last trial stays
      // or SubsetGaussNewtonSolverT / GaussNewtonSolverQRT (subset_gauss_newton_solver.cpp:117-142,
      // gauss_newton_solver_qr.cpp:126-149): c_1 = 1e-4 against the directional derivative J^T r . delta
      const float scaledError = 1e-3f * float(curError);
      double gd = 0.0;
      if (doLineSearch == 2) {
        float part = 0.f;
        for (int c = tid; c < n; c += 256) {
          part += s.g[c] * s.d0[c];
        }
        gd = double(blockSumF(s, part, tid));
      }
      float scale = 1.f;
      for (int ls = 0; ls < 10; ++ls) {
        for (int i = tid; i < P; i += 256) {
          s.dfull[i] = s.th[i];
        }
        __syncthreads();
        for (int c = tid; c < n; c += 256) {
          s.dfull[fv.solveList[c]] -= scale * s.d0[c];
        }
        __syncthreads();
        const double eNew = blockError<kGen, kReuse, !kCsrLds>(rig, rv, pb, fv, s, s.dfull, b, tid, &stateError);
        if ((curError - eNew) >= (doLineSearch == 2 ? double(1e-4f * scale) * gd : double(scale * scaledError))) {
          break;
        }
        scale *= 0.5f;
      }
      for (int i = tid; i < P; i += 256) {
        s.th[i] = s.dfull[i];
      }
      stateValid = kReuse && !hasParamRows; // the last trial evaluated IS the new theta
    } else if (!notPd) {
      for (int c = tid; c < n; c += 256) {
        s.th[fv.solveList[c]] -= s.d0[c]; // skeleton_solver_function.cpp:158
      }
      stateValid = false;
    }
    break;
    }
    } // trust steps
    if (st.paramHistory != nullptr) { // iterationHistory_["parameters"].col(iteration_) = parameters_ (solver.cpp:101-106)
      float* ph = st.paramHistory + (size_t(b) * fp.maxIterations + it) * size_t(P);
      for (int i = tid; i < P; i += 256) {
        ph[i] = kMix ? float(m.th[i]) : s.th[i];
      }
    }
    // MMX_PRECISION_AUTO with a mixed-precision second pass (fp.autoAbort): an element one of this iteration's factorisations
    // marks (precision estimate above the bound) will be solved again from its initial parameters anyway -- it stops here,
    // its parameters are not written back.  The pivot ratio is a property of the problem class: a marked class pays ONE
    // iteration of the single-precision pass instead of all of them.
    // The precision estimate, iteration by iteration (round 6): what the rounding of g = J^T r contributes to theta in THIS
    // iteration is ~ eps x cond(iteration) x |r| / |J| -- the noise of g scales with the RESIDUAL.  Round 5 took the worst pivot
    // ratio of the whole solve at full weight, which is the first iteration's weight; an iteration whose residual has fallen to a
    // thousandth (the LM schedule's last ones, whose small lambda gives them the worst pivot ratios of the solve) contributes a
    // thirtieth of that.  With a fixed lambda the pivot ratio is the same in every iteration and the largest weight is the first
    // iteration's 1: the estimate -- and its calibration (mmx_device.hpp) -- are round 5's.  Wave 0 joins its sixteen diagonal
    // lanes' w of this iteration; thread 0 keeps max w and max w sqrt(e_it / e_0).
    if (wave == 0) {
      float w = pivotWorst;
      w = fmaxf(w, dppMoveF<0xB1>(w));
      w = fmaxf(w, dppMoveF<0x4E>(w));
      w = fmaxf(w, dppMoveF<0x141>(w));
      w = fmaxf(w, dppMoveF<0x140>(w));
      pivotWorst = 0.f; // (per iteration from here on)
      if (lane == 0) {
        float e0 = estAcc[2];
        if (e0 < 0.f) {
          e0 = estAcc[2] = float(curError);
        }
        estAcc[0] = fmaxf(estAcc[0], w);
        estAcc[1] = fmaxf(estAcc[1], w * (e0 > 0.f ? sqrtf(float(curError) / e0) : 1.f));
      }
    }
    if (tid == 0) {
      const double e = curError;
      if (st.errorHistory != nullptr) {
        st.errorHistory[size_t(b) * fp.maxIterations + it] = e;
      }
      itersDone = it + 1;
      if (badPivot && !kMix) { // (kMix: the factor is the preconditioner -- a floored pivot costs CG steps, and those are what is reported)
        s.flags[2] |= 2; // MMX_SOLVE_NOT_PD
      }
      if (floored) {
        s.flags[2] |= 4; // MMX_SOLVE_DAMPING_FLOORED
      }

      const bool converged = fabs(lastError - e) / (fabs(e) + double(FLT_MIN)) <= double(fp.threshold) * double(FLT_EPSILON);
      s.flags[0] = (it >= fp.minIterations && converged) ? 1 : 0;
      lastError = e;
      if (!kMix && !kTR && fp.autoAbort != 0 && st.precisionBound > 0.f) {
        // MMX_PRECISION_AUTO with a mixed-precision second pass: an element whose estimate SO FAR exceeds the bound will be solved
        // again from its initial parameters anyway -- it stops here, its parameters are not written back
        const float estNow = kPrecisionGain * FLT_EPSILON * estAcc[1] / kPivotFloorOrOne;
        if (!(estNow <= st.precisionBound)) {
          s.flags[0] = 2; // marked: leave
        }
      }
    }
    __syncthreads();
    MMX_CLK(10)
    if (s.flags[0] != 0) {
      break;
    }
  }

  // NaN/Inf guard of the batched driver (pymomentum/tensor_ik/tensor_ik.cpp:168-173): theta in
  // global memory still holds the initial parameters, so "revert" = do not write
  int bad = 0;
  for (int i = tid; i < P; i += 256) {
    if (!isfinite(kMix ? float(m.th[i]) : s.th[i])) {
      bad = 1;
    }
  }
  bad = __syncthreads_or(bad);
  const bool aborted = !kMix && !kTR && s.flags[0] == 2; // (fp.autoAbort: the element is MMX_PRECISION_AUTO's second pass's)
  float th2 = 0.f;
  if (!bad && !aborted) {
    for (int i = tid; i < P; i += 256) {
      const float v = kMix ? float(m.th[i]) : s.th[i];
      thg[i] = v;
      th2 += v * v;
    }
  }
  th2 = blockSumF(s, th2, tid);
  if (tid == 0) {
    // Estimated distance of theta from the same solve in double, relative to |theta|: kPrecisionGain * eps * max over the
    // iterations of sqrt(e_it / e_0) / (smallest pivot ratio d_jj / (H_jj + mu) of the iteration's factorisations) ~ eps x cond
    // x the residual's share -- the rounding of g = J^T r is amplified by |H^-1| ~ cond on its way into the step, whatever the
    // refinement through J does for the linear solve (mmx_device.hpp has the calibration).  [1] of the diagnostics stays the
    // smallest pivot ratio of the whole solve.
    const float worst = estAcc[0], wS = estAcc[1];
    const float ratio = worst > 0.f ? kPivotFloorOrOne / worst : 1.f;
    const float est = wS > 0.f ? kPrecisionGain * FLT_EPSILON * wS / kPivotFloorOrOne : kPrecisionGain * FLT_EPSILON;
    int stt = bad ? 1 : s.flags[2];
    if (kMix) { // the estimate is the single-precision solves' (informational here): marked only when the CG did not converge
      if (!bad && mixUnconverged) {
        stt |= 8;
      }
      stt |= 32; // MMX_SOLVE_MIXED
    } else if (!bad && st.precisionBound > 0.f && !(est <= st.precisionBound)) {
      stt |= 8; // MMX_SOLVE_PRECISION_SUSPECT
    }
    if (st.diag != nullptr) {
      float* dg = st.diag + 4 * size_t(b);
      dg[0] = est;
      dg[1] = ratio;
      dg[2] = kMix ? float(mixApplied) / float(itersDone > 0 ? itersDone : 1) : sqrtf(refineWorst);
      dg[3] = sqrtf(th2);
    }
    st.iterations[b] = itersDone;
    st.finalError[b] = curError;
    st.status[b] = stt;
    s.flags[3] = itersDone;
  }
  if (kMix && sel.map != nullptr) { // (list mode) the rows of the histories past this run's last iteration still hold the single-precision pass's
    __syncthreads();
    for (int i = s.flags[3] + tid; i < fp.maxIterations; i += 256) {
      if (st.errorHistory != nullptr) {
        st.errorHistory[size_t(b) * fp.maxIterations + i] = 0.0;
      }
      if (st.stepHistory != nullptr) {
        st.stepHistory[(size_t(b) * fp.maxIterations + i) * 2] = 0.0;
        st.stepHistory[(size_t(b) * fp.maxIterations + i) * 2 + 1] = 0.0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// treeNormalEquationsKernel: H = J^T J and g = J^T r of ONE Gauss-Newton iteration from the tree moments
// (phases A-G of fusedSolveKernel: same formulas, same slot tables, same matrix-core tile products), written to
// HBM for the explicit-Jacobian solver's Cholesky step -- for systems beyond the fused kernel's LDS budget
// (BASELINE configs[4]: n = 260, 153 tiles).  O(n^2) work per instance instead of the M n^2 of the product
// J^T J (normalEquationsMfmaKernel: 7.0 ms per 8192 instances at cfg5), and J is not read.  One workgroup per
// instance, the batch-shared integer tables straight from L2 (no LDS copies: one workgroup per CU at this
// size anyway).  Position / orientation constraints with batch-shared parents; anything else keeps the dense
// kernel (launchTreeNormalEquations returns false).
// Output: H[i * n + j] for i >= j (what choleskyStepGlobalKernel reads), no lambda; g[n].
// ---------------------------------------------------------------------------------------------
#if !defined(MMX_FUSED_GROUP) || MMX_FUSED_GROUP == 4 // (the tree kernels of the wide route: a translation unit of their own, compiled WITH the machine-level loop-invariant code motion the solve kernel's groups switch off -- momentum_amd/build.py)
// what treeNormalEquationsKernel hands to treeRefineKernel, per instance: js | up | ur | us
struct TreeStateLayout {
  size_t js, up, ur, us, total;
};
__host__ __device__ inline TreeStateLayout treeStateLayout(int J, int U) {
  TreeStateLayout l;
  l.js = 0;
  l.up = alignUp4(size_t(kJs) * J);
  l.ur = l.up + alignUp4(3 * size_t(U));
  l.us = l.ur + alignUp4(3 * size_t(U));
  l.total = l.us + alignUp4(size_t(U));
  return l;
}

struct TreeNeLds {
  float *th, *js, *up, *uy, *us, *own1, *own2, *sub1, *sub2, *umom, *jd, *srcT;
  double *fkA, *fkB; // buffers of the pointer-jumping FK; fkB lies over own1 / own2 (written in phase D, after FK)
  int* span; // span[slot] = tin | tout << 16
  int *subSize, *loadedPos; // copies of the tables the subtree sums walk (read once per inner step)
  int *posUnitStart, *posUnits; // ... and of the units-per-joint lists the own sums walk
  int* kRange; // [2 rowTiles] treeSumRanges
  double* red;
};

__host__ __device__ inline size_t treeNeLdsFloats(int J, int P, int U, int nsrc, TreeNeLds* out, float* base) {
  size_t off = 0;
  auto take = [&](size_t count) {
    const size_t o = off;
    off += alignUp4(count);
    return o;
  };
  const size_t oTh = take(P), oJs = take(size_t(kJs) * J), oFk = take(fkBufFloats(J));
  const size_t oUp = take(3 * size_t(U)), oUy = take(3 * size_t(U)), oUs = take(U);
  const size_t oOwn1 = take(size_t(kC1) * J), oOwn2 = take(size_t(kC2) * J);
  // one region, two lives: joint parameters (FK) and the per-unit moments (D), then the subtree sums
  const size_t life1 = alignUp4(7 * size_t(J)) > alignUp4(size_t(kUmom) * U) ? alignUp4(7 * size_t(J)) : alignUp4(size_t(kUmom) * U);
  const size_t life2 = alignUp4(size_t(kC1) * J) + alignUp4(size_t(kC2) * J);
  const size_t oR = take(life1 > life2 ? life1 : life2);
  const size_t oSrc = take(size_t(kSrcCh) * size_t(srcStrideFor(nsrc)));
  const size_t oSpan = take(nsrc);
  const size_t oSub = take(J), oLoaded = take(J), oPus = take(size_t(J) + 1), oPu = take(U), oKr = take(2 * ((size_t(J) + 15) / 16));
  const size_t oRed = take(32); // one double per wave (up to sixteen)
  if (out != nullptr) {
    out->span = reinterpret_cast<int*>(base + oSpan);
    out->kRange = reinterpret_cast<int*>(base + oKr);
    out->posUnitStart = reinterpret_cast<int*>(base + oPus), out->posUnits = reinterpret_cast<int*>(base + oPu);
    out->subSize = reinterpret_cast<int*>(base + oSub), out->loadedPos = reinterpret_cast<int*>(base + oLoaded);
    out->th = base + oTh, out->js = base + oJs;
    out->fkA = reinterpret_cast<double*>(base + oFk), out->fkB = reinterpret_cast<double*>(base + oOwn1); // (kC1 + kC2) J >= fkBufFloats(J) from three joints on
    out->up = base + oUp, out->uy = base + oUy, out->us = base + oUs;
    out->own1 = base + oOwn1, out->own2 = base + oOwn2;
    out->jd = base + oR, out->umom = base + oR;
    out->sub1 = base + oR, out->sub2 = base + oR + alignUp4(size_t(kC1) * J);
    out->srcT = base + oSrc;
    out->red = reinterpret_cast<double*>(base + oRed);
  }
  return off;
}

// The COMPACT carve (round 5): the same buffers placed by their lifetimes instead of side by side, so that two workgroups
// of the plain instantiation share a CU (BASELINE configs[4], J = U = 300: 79.7 KB instead of 140).  Phases and what lives:
//   FK      js | fkA | fkB (the joint parameters jd in its head: read before the first jump round writes there)
//   C       js | up uy us (over fkA)
//   D own   up uy us | umom | own2 (over js: the joint states have left for the hand-over buffer in HBM, phase E reads
//           them from there) | own1 (behind umom)
//   D sub   own1 own2 | sub2 sub1 (over up uy us / umom)
//   E, G    sub1 sub2 | srcT (over own2) ; own1 stays the scratch of the split term records
// Usable when the joint-state block covers own2 and srcT and sub1 ends before own1 begins (checked here: 0 = not usable).
__host__ __device__ inline size_t treeNeCompactLdsFloats(int J, int P, int U, int nsrc, TreeNeLds* out, float* base) {
  size_t off = 0;
  auto take = [&](size_t count) {
    const size_t o = off;
    off += alignUp4(count);
    return o;
  };
  const size_t oTh = take(P), oSpan = take(nsrc);
  const size_t oSub = take(J), oLoaded = take(J), oPus = take(size_t(J) + 1), oPu = take(U), oKr = take(2 * ((size_t(J) + 15) / 16));
  const size_t oRed = take(32);
  const size_t X = off; // the persistent part; the scratch plan follows
  const size_t a = alignUp4(size_t(kJs) * J), f = alignUp4(fkBufFloats(J)), u3 = alignUp4(3 * size_t(U)), u1 = alignUp4(U);
  const size_t m = alignUp4(size_t(kUmom) * U), o1 = alignUp4(size_t(kC1) * J), o2 = alignUp4(size_t(kC2) * J);
  const size_t st = size_t(kSrcCh) * size_t(srcStrideFor(nsrc));
  const size_t endU = a + 2 * u3 + u1, endM = endU + m;
  const size_t oOwn1 = endM > a + o2 + o1 ? endM : a + o2 + o1;
  if (o2 > a || st > a || alignUp4(7 * size_t(J)) > f) {
    return 0;
  }
  size_t S = a + 2 * f;
  S = oOwn1 + o1 > S ? oOwn1 + o1 : S;
  if (out != nullptr) {
    float* sc = base + X;
    out->th = base + oTh, out->span = reinterpret_cast<int*>(base + oSpan);
    out->subSize = reinterpret_cast<int*>(base + oSub), out->loadedPos = reinterpret_cast<int*>(base + oLoaded);
    out->posUnitStart = reinterpret_cast<int*>(base + oPus), out->posUnits = reinterpret_cast<int*>(base + oPu);
    out->kRange = reinterpret_cast<int*>(base + oKr), out->red = reinterpret_cast<double*>(base + oRed);
    out->js = sc;
    out->fkA = reinterpret_cast<double*>(sc + a), out->fkB = reinterpret_cast<double*>(sc + a + f), out->jd = sc + a + f;
    out->up = sc + a, out->uy = sc + a + u3, out->us = sc + a + 2 * u3;
    out->umom = sc + endU;
    out->own2 = sc, out->own1 = sc + oOwn1;
    out->sub2 = sc + a, out->sub1 = sc + a + o2;
    out->srcT = sc;
  }
  return X + S;
}

// LDS of the kExtraRows instantiation, carved behind the tables above (kept out of TreeNeLds: the plain kernel's carve
// stays the small all-in-registers struct it was)
struct TreeNeExtraLds {
  int* col; // [P] parameter -> solve column or -1 (parameter-space rows)
  float* pdiag; // [NP] diagonal contributions of the parameter-space rows
  float *gEv, *gRes, *gJ; // rows of the further joint error functions / ellipsoid limits (fusedSolveKernel kGen)
  int* unitPos; // [U] DFS position of each unit's joint (per-instance constraint parents)
};
__host__ __device__ inline size_t treeNeExtraLdsFloats(int P, int n, int GT, int genRows, int U, TreeNeExtraLds* out, float* base) {
  const size_t NP = (size_t(n) + 15) & ~size_t(15), rowsGp = (size_t(genRows) + 3) & ~size_t(3);
  const size_t oCol = 0, oPd = oCol + alignUp4(P), oGev = oPd + NP, oGres = oGev + alignUp4(size_t(kGenEv) * GT), oGj = oGres + rowsGp;
  if (out != nullptr) {
    out->col = reinterpret_cast<int*>(base + oCol), out->pdiag = base + oPd, out->gEv = base + oGev, out->gRes = base + oGres, out->gJ = base + oGj;
  }
  const size_t oUp = oGj + rowsGp * size_t(srcStrideFor(int(NP)));
  if (out != nullptr) {
    out->unitPos = reinterpret_cast<int*>(base + oUp);
  }
  return oUp + alignUp4(U);
}

// kExtraRows: the instantiation for problems with parameter-space rows (limits, model prior) and / or further joint error
// functions / ellipsoid limits; the plain one carries none of that code
// kWaves: wavefronts of the workgroup (4; 16 for the instantiation without extra rows: one workgroup per CU is all the
// LDS allows, so four waves are ONE per SIMD -- sixteen give every phase that deals work by thread or by wave four times
// the lanes and each SIMD three more waves to switch to; the term records stay with the first 256 threads)
// kCompact: the lifetime-packed LDS carve (treeNeCompactLdsFloats), two workgroups per CU; needs the hand-over buffer
// (`state`), from which phase E reads the joint states back
template <bool kExtraRows, int kWaves = 4, bool kCompact = false>
__global__ void __launch_bounds__(64 * kWaves, kCompact ? 2 : 1) treeNormalEquationsKernel(
    RigDev rig,
    ProblemDev pb,
    FusedDev fd,
    const float* __restrict__ theta, // [B][P]
    float* __restrict__ jtj, // [B][n*n], lower triangle written
    float* __restrict__ jtr, // [B][n]
    const int32_t* __restrict__ done,
    double* __restrict__ errOut, // [B] error at theta (SkeletonSolverFunctionT::getJacobian's return value), or null
    float* __restrict__ state, // [B][treeStateFloats] joint states and units for treeRefineKernel, or null
    long long* __restrict__ clk, // profiling aid (MMX_PHASE_CLOCKS): per-phase cycles of block 0, or null
    float* __restrict__ genState, // [B][treeGenStateFloats] J_g and its residual rows for treeRefineKernel, or null
    int tileMajor) { // 0: jtj = [n][n] row-major, lower triangle; 1: [tile (I,J) at I(I+1)/2 + J][col][row] (what the tiled factor reads) // profiling aid (MMX_PHASE_CLOCKS): per-phase cycles of block 0, or null
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(kWaves == 4 || ((kWaves == 8 || kWaves == 16) && !kExtraRows), "the helpers of the extra rows are written for 256 threads");
  constexpr int kT = 64 * kWaves;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (done != nullptr && done[b] != 0) {
    return;
  }
  selectInstanceRig(rig, b);
  selectInstanceWeights(pb, b);
  const int J = rig.J, P = rig.P, U = fd.U, n = fd.n, nsrc = fd.nsrc;
  const int NB = (n + 15) >> 4, NP = 16 * NB, T = NB * (NB + 1) / 2;
  static_assert(!kCompact || !kExtraRows, "the compact carve is the plain instantiation's");
  TreeNeLds t;
  const size_t baseFloats = kCompact ? treeNeCompactLdsFloats(J, P, U, nsrc, &t, smem) : treeNeLdsFloats(J, P, U, nsrc, &t, smem);
  TreeNeExtraLds x{};
  if (kExtraRows) {
    treeNeExtraLdsFloats(P, n, fd.GT, fd.genRows, U, &x, smem + baseFloats);
  }
  const bool hasParamRows = kExtraRows && pb.M > pb.rowsJoint; // limit / model-parameter rows present (uniform)
  const bool hasGen = kExtraRows && fd.GT > 0; // further joint error functions / ellipsoid limits: a dense block of rows J_g in LDS
  const int gst = srcStrideFor(NP), rowsGp = (fd.genRows + 3) & ~3;
  if (hasGen) {
    for (int i = tid; i < rowsGp * gst; i += 256) {
      x.gJ[i] = 0.f;
    }
    for (int i = tid; i < rowsGp; i += 256) {
      x.gRes[i] = 0.f;
    }
  }
  long long tclk = clock64();
#define MMX_TCLK(slot)                 \
  if (clk != nullptr && b == 0) {      \
    __syncthreads();                   \
    if (tid == 0) {                    \
      const long long now = clock64(); \
      clk[slot] += now - tclk;         \
      tclk = now;                      \
    }                                  \
  }
  // the fused kernel's helpers work on these views; here the tables stay where they are (L2)
  FusedLds s{};
  s.th = t.th, s.js = t.js, s.fkA = t.fkA, s.fkB = t.fkB, s.jd = t.jd;
  s.up = t.up, s.uy = t.uy, s.us = t.us, s.own1 = t.own1, s.own2 = t.own2, s.sub1 = t.sub1, s.sub2 = t.sub2, s.umom = t.umom;
  s.srcT = t.srcT, s.red = t.red;
  RigView rv;
  rv.J = J, rv.P = P, rv.R = rig.R, rv.numLevels = rig.numLevels, rv.jumpRounds = rig.jumpRounds;
  rv.parent = rig.parent, rv.preRot = rig.preRot, rv.offset = rig.offset;
  rv.ptOuter = rig.ptOuter, rv.ptInner = rig.ptInner, rv.ptValue = rig.ptValue, rv.ptOffsets = rig.ptOffsets;
  rv.hasOffsets = rig.ptOffsetsNonZero != 0;
  rv.levelOrder = rig.levelOrder, rv.levelStart = rig.levelStart;
  rv.rowRec = rig.ptRowRec, rv.numRowRec = rig.numRowRec; // (the transform's non-empty rows: one load per thread and walk)
  FusedView fv;
  fv.U = U, fv.Kp = fd.Kp, fv.subSize = t.subSize, fv.dfsJoint = fd.dfsJoint, fv.loadedPos = t.loadedPos, fv.numLoaded = fd.numLoaded;
  fv.colToSolve = nullptr, fv.unitPos = pb.unitTin, fv.posUnitStart = t.posUnitStart, fv.posUnits = t.posUnits, fv.solveList = fd.solveList;
  for (int i = tid; i < P; i += kT) {
    s.th[i] = theta[size_t(b) * P + i];
  }
  for (int i = tid; i < J; i += kT) {
    t.subSize[i] = fd.subSize[i];
    t.loadedPos[i] = i < fd.numLoaded ? fd.loadedPos[i] : 0;
  }
  for (int i = tid; i <= J; i += kT) {
    t.posUnitStart[i] = fd.posUnitStart[i];
  }
  for (int i = tid; i < U; i += kT) {
    t.posUnits[i] = fd.posUnits[i];
  }
  for (int e = tid; e < nsrc; e += kT) {
    t.span[e] = fd.srcs[e].tin | (fd.srcs[e].tout << 16);
  }
  if (kExtraRows) {
    for (int i = tid; i < P; i += 256) {
      x.col[i] = -1;
    }
  }
  __syncthreads();
  if (kExtraRows) {
    for (int c = tid; c < n; c += 256) {
      x.col[fd.solveList[c]] = c;
    }
    if (pb.instPosParent != nullptr || pb.instOriParent != nullptr) { // per-instance constraint parents: this element's unit lists
      fv.numLoaded = buildInstanceUnitTables(pb, fd, b, J, U, tid, x.unitPos, t.posUnitStart, t.posUnits, t.loadedPos);
      fv.unitPos = x.unitPos;
    }
  }
  treeSumRanges(t.subSize, t.loadedPos, fv.numLoaded, J, tid, t.kRange); // (barriers follow before the first tree sum)
  MMX_TCLK(0)
  // the thread's first unit: its payload is requested before the forward kinematics (an HBM round trip in the shadow of
  // phase B instead of in the open at the head of phase C)
  const UnitInput uin0 = loadUnitInput(pb, b, tid < U ? tid : U);
  // ---- A, B: forward kinematics with rotation axes
  blockFk<true, kT>(rv, s, s.th, tid, true);
  MMX_TCLK(1)
  // ---- C: units
  {
    const TreeStateLayout sl = treeStateLayout(J, U);
    float* stb = state != nullptr ? state + size_t(b) * sl.total : nullptr;
    double e = 0.0;
    for (int u = tid; u < U; u += kT) {
      const Unit un = evalUnitFrom(pb, u == tid ? uin0 : loadUnitInput(pb, b, u), s.js, u);
      s.up[3 * u] = un.v.x, s.up[3 * u + 1] = un.v.y, s.up[3 * u + 2] = un.v.z;
      const float sg2 = un.sigma * un.sigma;
      s.uy[3 * u] = sg2 * un.f.x, s.uy[3 * u + 1] = sg2 * un.f.y, s.uy[3 * u + 2] = sg2 * un.f.z;
      s.us[u] = un.sigma;
      e += double(un.werr);
      if (stb != nullptr) {
        float* o = stb + sl.ur + 3 * u;
        o[0] = un.sigma * un.f.x, o[1] = un.sigma * un.f.y, o[2] = un.sigma * un.f.z;
      }
    }
    if (hasParamRows) {
      e += paramRowsError<true>(rig, pb, P, s.th, b, tid);
    }
    if (hasGen) {
      e += generalRowsEvaluate(pb, s.js, b, U, tid, x.gEv, x.gRes);
    }
    e = waveReduceSum(e);
    if (lane == 0) {
      s.red[wave] = e;
    }
    __syncthreads();
    if (hasGen) { // J_g from the slots' table in global memory (slot numbering of phase F); its barrier: phase D's
      generalRowsGather(
          s.js, x.gEv, x.gJ, gst, fd.GT, n, tid,
          [&](int e2) {
            const ColumnSourceDev cs = fd.srcs[e2];
            return GenSlot{cs.tin, cs.tout, cs.joint, cs.dof, cs.parent, cs.weight};
          },
          [&](int c, int& e0, int& e1) { e0 = fd.slotBase + fd.srcStart[c], e1 = fd.slotBase + fd.srcStart[c + 1]; });
    }
    if (errOut != nullptr && tid == 0) {
      double eSum = (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]);
      for (int w0 = 4; w0 < kWaves; w0 += 4) {
        eSum += (s.red[w0] + s.red[w0 + 1]) + (s.red[w0 + 2] + s.red[w0 + 3]);
      }
      errOut[b] = eSum;
    }
    if (stb != nullptr) {
      for (int i = tid; i < kJs * J; i += kT) {
        stb[sl.js + i] = s.js[i];
      }
      for (int i = tid; i < 3 * U; i += kT) {
        stb[sl.up + i] = s.up[i];
      }
      for (int i = tid; i < U; i += kT) {
        stb[sl.us + i] = s.us[i];
      }
    }
  }
  MMX_TCLK(2)
  // ---- D: own sums (per-unit moments -> per-joint sums), then subtree sums (the moments' scratch is dead by then)
  ownSums<kT>(fv, s, s.umom, U, tid);
  __syncthreads();
  MMX_TCLK(3)
  treeSumT<kC1, true, kC1, 8>(fv.subSize, fv.loadedPos, fv.numLoaded, s.own1, s.sub1, J, wave, kWaves, lane, t.kRange);
  treeSumT<kC2Used, true, kC2, 8>(fv.subSize, fv.loadedPos, fv.numLoaded, s.own2, s.sub2, J, wave, kWaves, lane, t.kRange);
  __syncthreads();
  MMX_TCLK(4)
  // ---- E: per-slot tables (see fusedSolveKernel phase E)
  const int sst = srcStrideFor(nsrc);
  float* srcD = s.srcT;
  float* srcA = s.srcT + 7 * sst;
  float* srcG = s.srcT + 14 * sst;
  // (compact carve: the joint states left LDS after phase C -- own2 and srcT lie over them -- and come back from the
  // hand-over buffer this workgroup wrote in phase C: L2 hits, one round trip per slot, all its loads independent)
  const float* jsE = kCompact ? state + size_t(b) * treeStateLayout(J, U).total + treeStateLayout(J, U).js : s.js;
  for (int e = tid; e < nsrc; e += kT) {
    const ColumnSourceDev cs = fd.srcs[e];
    const float* a = jsE + kJs * cs.joint;
    const float* sb = s.sub2 + kC2 * cs.tin;
    const F3 ta{a[0], a[1], a[2]};
    const float m0 = sb[0];
    const F3 m1{sb[1], sb[2], sb[3]};
    F3 al, bv{0.f, 0.f, 0.f}, g0, ax;
    float bs = 0.f, tr;
    if (cs.dof < 3) {
      al = transAxisCol(jsE, cs.parent, cs.dof);
      g0 = m0 * al;
      ax = cross(m1, al);
      tr = dot(al, m1);
    } else if (cs.dof < 6) {
      const float* w = a + 8 + 3 * (cs.dof - 3);
      const F3 om{w[0], w[1], w[2]};
      al = F3{0.f, 0.f, 0.f} - cross(om, ta);
      bv = om;
      g0 = m0 * al + cross(om, m1);
      const float t2 = (sb[4] + sb[7] + sb[9]) + (sb[10] + sb[13] + sb[15]);
      const F3 Mo{
          (sb[4] + sb[10]) * om.x + (sb[5] + sb[11]) * om.y + (sb[6] + sb[12]) * om.z,
          (sb[5] + sb[11]) * om.x + (sb[7] + sb[13]) * om.y + (sb[8] + sb[14]) * om.z,
          (sb[6] + sb[12]) * om.x + (sb[8] + sb[14]) * om.y + (sb[9] + sb[15]) * om.z};
      ax = cross(m1, al) + (t2 * om - Mo);
      tr = dot(al, m1);
    } else {
      al = F3{0.f, 0.f, 0.f} - kLn2 * ta;
      bs = kLn2;
      g0 = m0 * al + kLn2 * m1;
      ax = cross(m1, al);
      tr = dot(al, m1) + kLn2 * (sb[4] + sb[7] + sb[9]);
    }
    const float w = cs.weight;
    float* d = srcD + e;
    d[0] = w * g0.x, d[sst] = w * g0.y, d[2 * sst] = w * g0.z;
    d[3 * sst] = w * ax.x, d[4 * sst] = w * ax.y, d[5 * sst] = w * ax.z;
    d[6 * sst] = w * tr;
    float* o = srcA + e;
    o[0] = w * al.x, o[sst] = w * al.y, o[2 * sst] = w * al.z;
    o[3 * sst] = w * bv.x, o[4 * sst] = w * bv.y, o[5 * sst] = w * bv.z;
    o[6 * sst] = w * bs;
    srcG[e] = w * sourceGradient(cs.joint, cs.dof, cs.parent, jsE, s.sub1 + kC1 * cs.tin);
  }
  __syncthreads();
  MMX_TCLK(5)
  // ---- F: g
  for (int c = tid; c < n; c += kT) {
    float acc = srcG[c];
    const int e1 = fd.slotBase + fd.srcStart[c + 1];
    for (int e = fd.slotBase + fd.srcStart[c]; e < e1; ++e) {
      acc += srcG[e];
    }
    if (hasParamRows) { // limit / model-parameter rows: evaluated on the fly from theta (fusedSolveKernel phase F)
      const ParamCol pc = paramRowsColumn(rig, pb, fd, s.th, nullptr, x.col, P, b, c, fd.solveList[c]);
      acc += pc.g;
      x.pdiag[c] = pc.h; // parked until the tiles of H are in place
    }
    if (hasGen) {
      for (int r = 0; r < fd.genRows; ++r) {
        acc += x.gJ[r * gst + c] * x.gRes[r];
      }
    }
    jtr[size_t(b) * n + c] = acc;
  }
  if (hasGen && genState != nullptr) { // hand-over to the refinement: J_g (rows x gst), then the residual rows
    float* gs = genState + size_t(b) * (size_t(rowsGp) * gst + rowsGp);
    for (int i = tid; i < rowsGp * gst; i += 256) {
      gs[i] = x.gJ[i];
    }
    for (int i = tid; i < rowsGp; i += 256) {
      gs[size_t(rowsGp) * gst + i] = x.gRes[i];
    }
  }
  // ---- G: the tiles of the lower triangle, two masked matrix-core products each, straight to HBM
  float* Hb = jtj + size_t(b) * size_t(n) * size_t(n);
  float* Ht = jtj + size_t(b) * size_t(T) * 256;
  {
    const int i = lane & 15, gq = lane >> 4;
    const int k1 = gq < 3 ? 4 + gq : 6;
    const float z = gq == 3 ? 0.f : 1.f;
    struct TileOps { // operands of a tile (fusedSolveKernel phase G): rows of block I on the A side, of block Jc on the B side
      float dI0, aI0, dJ0, aJ0, dI1, aI1, dJ1, aJ1;
      int spanC, spanR[4];
    };
    auto loadOps = [&](int I, int Jc) {
      TileOps o;
      const int ri = 16 * I + i, ci = 16 * Jc + i;
      o.dI0 = srcD[gq * sst + ri], o.aI0 = srcA[gq * sst + ri];
      o.dJ0 = srcD[gq * sst + ci], o.aJ0 = srcA[gq * sst + ci];
      o.dI1 = srcD[k1 * sst + ri], o.aI1 = srcA[k1 * sst + ri];
      o.dJ1 = srcD[k1 * sst + ci], o.aJ1 = srcA[k1 * sst + ci];
      o.spanC = t.span[ci];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        o.spanR[q] = t.span[16 * I + 4 * gq + q];
      }
      return o;
    };
    // the wave's tiles: entries wave, wave + kWaves, ... of the list of structurally non-zero tiles (tile-major hand-over:
    // the tiled factor reads exactly those, mmx::TileMasks) or of all T tiles (row-major: every entry is written)
    const int numT = tileMajor ? fd.numTiles : T;
    auto tileAt = [&](int ti, int& I, int& Jc) { // (wave-uniform)
      if (tileMajor) {
        const int code = fd.tileList[ti < numT ? ti : 0];
        I = code & 0xff, Jc = code >> 8;
      } else {
        tileDecode(ti < numT ? ti : 0, I, Jc);
      }
    };
    int I, Jc;
    tileAt(wave, I, Jc);
    TileOps cur = loadOps(I, Jc);
    for (int ti = wave; ti < numT; ti += kWaves) {
      const int tI = I, tJ = Jc, tt = tileIndex(tI, tJ);
      const bool more = ti + kWaves < numT; // wave-uniform
      if (more) {
        tileAt(ti + kWaves, I, Jc);
      }
      const TileOps nxt = loadOps(I, Jc); // the next tile's LDS reads fly while this one multiplies
      v4f Pm{0.f, 0.f, 0.f, 0.f}, Qm{0.f, 0.f, 0.f, 0.f};
      Pm = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.dI0, cur.aJ0, Pm, 0, 0, 0);
      Qm = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.aI0, cur.dJ0, Qm, 0, 0, 0);
      Pm = __builtin_amdgcn_mfma_f32_16x16x4f32(z * cur.dI1, cur.aJ1, Pm, 0, 0, 0);
      Qm = __builtin_amdgcn_mfma_f32_16x16x4f32(z * cur.aI1, cur.dJ1, Qm, 0, 0, 0);
      const int tinC = cur.spanC & 0xffff, toutC = cur.spanC >> 16;
      v4f Gm{0.f, 0.f, 0.f, 0.f}; // + J_g^T J_g: a rank-genRows update on the matrix cores, operands straight from J_g
      if (hasGen) {
        for (int k = 0; k < (rowsGp >> 2); ++k) {
          const float* rowp = x.gJ + (4 * k + gq) * gst;
          Gm = __builtin_amdgcn_mfma_f32_16x16x4f32(rowp[16 * tI + i], rowp[16 * tJ + i], Gm, 0, 0, 0);
        }
      }
      float hv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * tI + 4 * gq + q, col = 16 * tJ + i;
        const int tinR = cur.spanR[q] & 0xffff, toutR = cur.spanR[q] >> 16;
        const bool rowDeep = tinC <= tinR && tinR < toutC;
        const bool colDeep = tinR <= tinC && tinC < toutR;
        hv[q] = rowDeep ? Pm[q] : (colDeep ? Qm[q] : 0.f);
        if (kExtraRows) {
          hv[q] += Gm[q];
        }
        if (!tileMajor && row < n && col <= row) {
          Hb[size_t(row) * n + col] = hv[q];
        }
      }
      if (tileMajor) { // [col][row] inside the tile: the lane's four rows are one 16-byte store, the wave's a contiguous KB
        *reinterpret_cast<float4*>(Ht + size_t(tt) * 256 + i * 16 + 4 * gq) = float4{hv[0], hv[1], hv[2], hv[3]};
      }
      cur = nxt;
    }
  }
  MMX_TCLK(6)
  // ---- the pairs that involve an extra source of a shared parameter: term records, one thread per entry run
  if (fd.termRounds > 0) {
    __threadfence_block();
    __syncthreads();
    float h = 0.f;
    auto entryAddr = [&](uint32_t y) { // tile-region offset of the fused kernel -> (row, col) of H
      int I, Jc;
      tileDecode(int(y >> 8), I, Jc);
      const int r = (y >> 4) & 15, c = int(((((y >> 2) & 3) ^ (r >> 2)) & 3) << 2) | int(y & 3);
      return tileMajor ? Ht + size_t(y >> 8) * 256 + c * 16 + r : Hb + size_t(16 * I + r) * n + (16 * Jc + c);
    };
    // (the host deals the records to 256 threads and to 1024: the sixteen-wave instantiation takes the latter, the
    // others keep their first 256 threads on the former)
    const int rounds = kWaves == 16 ? fd.termRounds16 : fd.termRounds; // (a multiple of 8)
    for (int k0 = 0; k0 < rounds; k0 += 8) {
      uint2 recs[8]; // the trip's records are requested together (one round trip, not eight)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint4* rp = kWaves == 16 ? fd.gTerms16 + size_t(k0 + u) * 1024 + tid : fd.gTerms + size_t(k0 + u) * 256 + (tid & 255);
        recs[u] = *reinterpret_cast<const uint2*>(rp);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t x = recs[u].x;
        if ((kWaves != 8 || tid < 256) && (x & (1u << 26))) {
          const int deep = x & 0xfff, anc = (x >> 12) & 0xfff;
          float hj = 0.f;
#pragma unroll
          for (int ch = 0; ch < 7; ++ch) {
            hj += srcD[ch * sst + deep] * srcA[ch * sst + anc];
          }
          h = (x & (1u << 24)) ? hj : h + hj;
          if (x & (1u << 25)) {
            const uint32_t y = recs[u].y;
            if (y & (1u << 30)) {
              s.own1[y & 0xffff] = h; // partial cell of a split entry (the own sums are dead)
            } else {
              *entryAddr(y) += h; // (an atomic without return instead was measured slower: 25 k against 18 k cycles for the phase)
            }
          }
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < fd.numComb; i += kT) {
      const int dest = fd.comb[3 * i], first = fd.comb[3 * i + 1], cnt = fd.comb[3 * i + 2];
      float* hp = entryAddr(uint32_t(dest));
      float v = *hp;
      for (int c = 0; c < cnt; ++c) {
        v += s.own1[first + c];
      }
      *hp = v;
    }
  }
  if (hasParamRows) { // J^T J of the parameter-space rows: diagonal entries, then the shared off-diagonal ones
    __threadfence_block();
    __syncthreads();
    auto hEntry = [&](int row, int col) { // row >= col
      return tileMajor ? Ht + size_t(tileIndex(row >> 4, col >> 4)) * 256 + (col & 15) * 16 + (row & 15) : Hb + size_t(row) * n + col;
    };
    for (int c = tid; c < n; c += 256) {
      *hEntry(c, c) += x.pdiag[c];
    }
    if (pb.wLimit > 0.f) {
      const float tWeight = 1e+1f * pb.wLimit;
      for (int d = tid; d < fd.numPairDests; d += 256) {
        float accp = 0.f;
        const int k1 = fd.pairStart[d + 1];
        for (int k = fd.pairStart[d]; k < k1; ++k) {
          const LimitRow row = evalLimit(rig, pb.limits[fd.pairLim[k]], s.th, pb.enabledMask, tWeight);
          float ca = 0.f, cb = 0.f; // the row's entries in the two columns of this H entry
#pragma unroll
          for (int e = 0; e < kLimitEntries; ++e) {
            const int sc = row.idx[e] >= 0 ? x.col[row.idx[e]] : -1;
            ca += sc == fd.pairCols[2 * d] ? row.coef[e] : 0.f;
            cb += sc == fd.pairCols[2 * d + 1] ? row.coef[e] : 0.f;
          }
          accp += ca * cb;
        }
        *hEntry(fd.pairCols[2 * d], fd.pairCols[2 * d + 1]) += accp;
      }
    }
  }
  MMX_TCLK(7)
#undef MMX_TCLK
}

size_t treeNormalEquationsLdsBytes(int J, int P, int U, int nsrc, int n, int GT, int genRows) {
  return (treeNeLdsFloats(J, P, U, nsrc, nullptr, nullptr) + treeNeExtraLdsFloats(P, n, GT, genRows, U, nullptr, nullptr)) * sizeof(float);
}
size_t treeGenStateFloats(int n, int genRows) {
  const size_t rowsGp = (size_t(genRows) + 3) & ~size_t(3);
  return rowsGp * size_t(srcStrideFor((n + 15) & ~15)) + rowsGp;
}

hipError_t launchTreeNormalEquations(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    const float* theta,
    float* jtj,
    float* jtr,
    const int32_t* done,
    double* errOut,
    float* state,
    long long* clk,
    float* genState,
    bool tileMajor,
    hipStream_t stream) {
  const size_t lds = treeNormalEquationsLdsBytes(rig.J, rig.P, fd.U, fd.nsrc, fd.n, fd.GT, fd.genRows);
  if (lds > 160 * 1024 - 64) {
    return hipErrorInvalidValue;
  }
  const bool extra = pb.M > pb.rowsJoint || fd.GT > 0 || pb.instPosParent != nullptr || pb.instOriParent != nullptr;
  if (!extra && state != nullptr) {
    // two workgroups of eight waves per CU when the lifetime-packed carve fits half a CU's LDS (BASELINE configs[4]: 79.7 KB):
    // every phase of this kernel is a latency chain of one workgroup (VALU active 8.5 % of the wave cycles at one workgroup of
    // sixteen waves per CU, profiles/r04_pmc_cfg5.txt) -- a second, independent workgroup fills the waits
    const size_t compact = treeNeCompactLdsFloats(rig.J, rig.P, fd.U, fd.nsrc, nullptr, nullptr) * sizeof(float);
    if (compact > 0 && compact <= 80 * 1024 - 64) {
      static LdsLimitCache ldsLimitC;
      hipError_t rc = ldsLimitC.ensure(reinterpret_cast<const void*>(treeNormalEquationsKernel<false, 8, true>), compact);
      if (rc != hipSuccess) {
        return rc;
      }
      hipLaunchKernelGGL((treeNormalEquationsKernel<false, 8, true>), dim3(pb.B), dim3(512), compact, stream, rig, pb, fd, theta, jtj, jtr, done, errOut, state, clk, genState, tileMajor ? 1 : 0);
      return hipGetLastError();
    }
  }
  if (!extra) { // many waves per workgroup (one workgroup per CU either way: the LDS footprint decides)
    // sixteen waves = four per SIMD (93 VGPRs: no spill under the 128 of that occupancy).  cfg5, one box, solves/s:
    // four waves 1.476e5, eight 1.626e5, sixteen 1.704e5 (round 3)
    constexpr int kNeWaves = 16;
    static LdsLimitCache ldsLimit8;
    hipError_t rc = ldsLimit8.ensure(reinterpret_cast<const void*>(treeNormalEquationsKernel<false, kNeWaves>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
    hipLaunchKernelGGL((treeNormalEquationsKernel<false, kNeWaves>), dim3(pb.B), dim3(64 * kNeWaves), lds, stream, rig, pb, fd, theta, jtj, jtr, done, errOut, state, clk, genState, tileMajor ? 1 : 0);
    return hipGetLastError();
  }
  // parameter-space rows, further joint blocks or per-instance parents: four waves (their helpers stride by 256 threads)
  static LdsLimitCache ldsLimit;
  {
    hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(treeNormalEquationsKernel<true>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
  }
  hipLaunchKernelGGL(treeNormalEquationsKernel<true>, dim3(pb.B), dim3(256), lds, stream, rig, pb, fd, theta, jtj, jtr, done, errOut, state, clk, genState, tileMajor ? 1 : 0);
  return hipGetLastError();
}

// =============================================================================================
// The refinement residual of the wide explicit solve through the tree: rho = J^T (r - J d) - lambda d from the
// joint states and units treeNormalEquationsKernel left in `state` -- the tangent pass down the tree and the
// adjoint pass up (fusedSolveKernel phase J with its tables read from global memory).  No dense J anywhere:
// with this kernel the wide path neither writes nor reads one.  grid = B, block = 256.
// =============================================================================================
struct TreeRefLds {
  float *js, *up, *ur, *us, *jd, *tanOwn, *tanPre, *own1, *sub1, *d0, *th, *gJ, *gRes;
  int *col, *subSize, *loadedPos, *posUnitStart, *posUnits, *kRange, *unitPos;
};
__host__ __device__ inline size_t treeRefineLdsFloats(int J, int P, int U, int n, TreeRefLds* out, float* base, int genRows = 0) {
  size_t off = 0;
  auto take = [&](size_t count) {
    const size_t o = off;
    off += alignUp4(count);
    return o;
  };
  const size_t NP = (size_t(n) + 15) & ~size_t(15);
  const size_t oJs = take(size_t(kJs) * J), oUp = take(3 * size_t(U)), oUr = take(3 * size_t(U)), oUs = take(U);
  const size_t r1 = size_t(kC1) * (U > J ? U : J); // jd (7 J), then the per-unit contributions / the subtree sums
  const size_t oR1 = take(r1 > 7 * size_t(J) ? r1 : 7 * size_t(J));
  const size_t oR2 = take(size_t(kTan > kC1 ? kTan : kC1) * J); // tanOwn, then the own sums
  const size_t oPre = take(size_t(kTan) * J);
  const size_t oD = take(NP), oCol = take(P), oSub = take(J), oLoaded = take(J), oPus = take(size_t(J) + 1), oPu = take(U);
  const size_t oKr = take(2 * ((size_t(J) + 15) / 16)), oTh = take(P);
  const size_t rowsGp = (size_t(genRows) + 3) & ~size_t(3);
  const size_t oGj = take(rowsGp * size_t(srcStrideFor(int(NP)))), oGres = take(rowsGp), oUpos = take(U);
  if (out != nullptr) {
    out->gJ = base + oGj, out->gRes = base + oGres;
    out->unitPos = reinterpret_cast<int*>(base + oUpos);
    out->kRange = reinterpret_cast<int*>(base + oKr);
    out->th = base + oTh;
    out->posUnitStart = reinterpret_cast<int*>(base + oPus), out->posUnits = reinterpret_cast<int*>(base + oPu);
    out->subSize = reinterpret_cast<int*>(base + oSub), out->loadedPos = reinterpret_cast<int*>(base + oLoaded);
    out->js = base + oJs, out->up = base + oUp, out->ur = base + oUr, out->us = base + oUs;
    out->jd = base + oR1, out->sub1 = base + oR1, out->tanOwn = base + oR2, out->own1 = base + oR2, out->tanPre = base + oPre;
    out->d0 = base + oD;
    out->col = reinterpret_cast<int*>(base + oCol);
  }
  return off;
}

// kWaves: wavefronts of the workgroup -- 4 for problems with parameter rows / further joint blocks / per-instance parents
// (their helpers stride by 256 threads), 8 for the plain problem: the kernel holds ~71 KB of LDS on the 300-joint rig (two
// workgroups per CU), so four waves are ONE per SIMD and every tree pass a chain of exposed LDS round trips
template <int kWaves>
__global__ void __launch_bounds__(64 * kWaves, kWaves == 4 ? 2 : 1) treeRefineKernel(
    RigDev rig,
    ProblemDev pb,
    FusedDev fd,
    const float* __restrict__ theta, // [B][P] the parameters the normal equations were built at
    const float* __restrict__ state, // [B][treeStateFloats]
    const float* __restrict__ genState, // [B][treeGenStateFloats] J_g and its residual rows (problems with such rows), or null
    const float* __restrict__ dvec, // [B][NP] the step
    float* __restrict__ rhoVec, // [B][NP]
    const int32_t* __restrict__ refState, // [B] 0: this instance is waiting for a refinement round
    float lambdaAll,
    const float* __restrict__ lambdaPer) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int kT = 64 * kWaves;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if (refState[b] != 0) {
    return;
  }
  selectInstanceWeights(pb, b);
  const int J = rig.J, P = rig.P, U = fd.U, n = fd.n;
  const int NP = (n + 15) & ~15;
  const float lambda = lambdaPer != nullptr ? lambdaPer[b] : lambdaAll;
  TreeRefLds t;
  treeRefineLdsFloats(J, P, U, n, &t, smem, fd.genRows);
  const bool hasGen = fd.GT > 0 && genState != nullptr;
  const int gst = srcStrideFor(NP), rowsGp = (fd.genRows + 3) & ~3;
  FusedLds s{};
  s.js = t.js, s.up = t.up, s.ur = t.ur, s.us = t.us, s.own1 = t.own1, s.sub1 = t.sub1, s.jd = t.jd, s.tanOwn = t.tanOwn, s.tanPre = t.tanPre, s.d0 = t.d0;
  FusedView fv;
  fv.U = U, fv.Kp = fd.Kp, fv.subSize = t.subSize, fv.dfsJoint = fd.dfsJoint, fv.loadedPos = t.loadedPos, fv.numLoaded = fd.numLoaded;
  fv.colToSolve = t.col, fv.unitPos = pb.unitTin, fv.posUnitStart = t.posUnitStart, fv.posUnits = t.posUnits, fv.solveList = fd.solveList;
  {
    const TreeStateLayout sl = treeStateLayout(J, U);
    const float* stb = state + size_t(b) * sl.total;
    for (int i = tid; i < kJs * J; i += kT) {
      s.js[i] = stb[sl.js + i];
    }
    for (int i = tid; i < 3 * U; i += kT) {
      s.up[i] = stb[sl.up + i];
      s.ur[i] = stb[sl.ur + i];
    }
    for (int i = tid; i < U; i += kT) {
      s.us[i] = stb[sl.us + i];
    }
    for (int i = tid; i < NP; i += kT) {
      s.d0[i] = i < n ? dvec[size_t(b) * NP + i] : 0.f;
    }
    for (int i = tid; i < P; i += kT) {
      t.col[i] = -1;
      t.th[i] = theta[size_t(b) * P + i];
    }
    if (hasGen) {
      const float* gs = genState + size_t(b) * (size_t(rowsGp) * gst + rowsGp);
      for (int i = tid; i < rowsGp * gst; i += kT) {
        t.gJ[i] = gs[i];
      }
      for (int i = tid; i < rowsGp; i += kT) {
        t.gRes[i] = gs[size_t(rowsGp) * gst + i];
      }
    }
    for (int i = tid; i < J; i += kT) {
      t.subSize[i] = fd.subSize[i];
      t.loadedPos[i] = i < fd.numLoaded ? fd.loadedPos[i] : 0;
    }
    for (int i = tid; i <= J; i += kT) {
      t.posUnitStart[i] = fd.posUnitStart[i];
    }
    for (int i = tid; i < U; i += kT) {
      t.posUnits[i] = fd.posUnits[i];
    }
  }
  __syncthreads();
  if (pb.instPosParent != nullptr || pb.instOriParent != nullptr) { // per-instance constraint parents: this element's unit lists
    fv.numLoaded = buildInstanceUnitTables(pb, fd, b, J, U, tid, t.unitPos, t.posUnitStart, t.posUnits, t.loadedPos);
    fv.unitPos = t.unitPos;
  }
  treeSumRanges(t.subSize, t.loadedPos, fv.numLoaded, J, tid, t.kRange);
  for (int c = tid; c < n; c += kT) {
    t.col[fd.solveList[c]] = c;
  }
  if (hasGen) { // w_g = r_g - J_g d, in place (each row by one thread; consumed by the rho loop, several barriers later)
    for (int r = tid; r < fd.genRows; r += kT) {
      float a = t.gRes[r];
      for (int c = 0; c < n; ++c) {
        a -= t.gJ[r * gst + c] * s.d0[c];
      }
      t.gRes[r] = a;
    }
  }
  __syncthreads();
  // joint-parameter delta jd = transform * delta
  csrRowsPrefetched<kT>(
      rig.ptOuter, rig.ptInner, rig.ptValue, rig.R, tid,
      [&](int c) {
        const int cs = t.col[c];
        return cs >= 0 ? s.d0[cs] : 0.f;
      },
      [&](int r, float a) { s.jd[r] = a; });
  __syncthreads();
  // tangent pass: per joint (by DFS position) C = T - Om x t - ln2 sd t, W = Om, S = sd ...
  for (int k = tid; k < J; k += kT) {
    const int q = fd.dfsJoint[k];
    const float* ja = s.js + kJs * q;
    const float* d = s.jd + 7 * q;
    const F3 ta{ja[0], ja[1], ja[2]};
    F3 Tv{0.f, 0.f, 0.f};
    if (d[0] != 0.f || d[1] != 0.f || d[2] != 0.f) {
      const int par = rig.parent[q];
      Tv = d[0] * transAxisCol(s.js, par, 0) + d[1] * transAxisCol(s.js, par, 1) + d[2] * transAxisCol(s.js, par, 2);
    }
    const F3 Om = d[3] * F3{ja[8], ja[9], ja[10]} + d[4] * F3{ja[11], ja[12], ja[13]} + d[5] * F3{ja[14], ja[15], ja[16]};
    const F3 C = Tv - cross(Om, ta) - (kLn2 * d[6]) * ta;
    float* o = s.tanOwn + kTan * k;
    o[0] = C.x, o[1] = C.y, o[2] = C.z, o[3] = Om.x, o[4] = Om.y, o[5] = Om.z, o[6] = d[6], o[7] = 0.f;
  }
  __syncthreads();
  // ... summed over each joint's ancestor chain
  treeSumT<7, false, kTan, 8>(fv.subSize, fv.loadedPos, fv.numLoaded, s.tanOwn, s.tanPre, J, wave, kWaves, lane);
  __syncthreads();
  // w = r - J d, y = sigma w per unit, then the first-order own sums
  if (U <= J) {
    for (int u = tid; u < U; u += kT) {
      const float* pre = s.tanPre + kTan * fv.unitPos[u];
      const F3 p{s.up[3 * u], s.up[3 * u + 1], s.up[3 * u + 2]};
      const bool point = u < fv.Kp;
      F3 v = cross(F3{pre[3], pre[4], pre[5]}, p);
      if (point) {
        v = F3{pre[0], pre[1], pre[2]} + v + (kLn2 * pre[6]) * p;
      }
      const float sg = s.us[u];
      firstOrderMoments(s.sub1 + kC1 * u, p, sg * (s.ur[3 * u] - sg * v.x), sg * (s.ur[3 * u + 1] - sg * v.y), sg * (s.ur[3 * u + 2] - sg * v.z), point);
    }
    __syncthreads();
    gatherOwnSums<kC1, kC1, kT>(fv, s, s.sub1, tid);
  } else {
    for (int k = tid; k < J; k += kT) {
      const int e0 = fv.posUnitStart[k], e1 = fv.posUnitStart[k + 1];
      float a1[kC1] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (e1 > e0) {
        const float* pre = s.tanPre + kTan * k;
        for (int e = e0; e < e1; ++e) {
          const int u = fv.posUnits[e];
          const F3 p{s.up[3 * u], s.up[3 * u + 1], s.up[3 * u + 2]};
          const bool point = u < fv.Kp;
          F3 v = cross(F3{pre[3], pre[4], pre[5]}, p);
          if (point) {
            v = F3{pre[0], pre[1], pre[2]} + v + (kLn2 * pre[6]) * p;
          }
          const float sg = s.us[u];
          float o[kC1];
          firstOrderMoments(o, p, sg * (s.ur[3 * u] - sg * v.x), sg * (s.ur[3 * u + 1] - sg * v.y), sg * (s.ur[3 * u + 2] - sg * v.z), point);
#pragma unroll
          for (int c = 0; c < kC1; ++c) {
            a1[c] += o[c];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < kC1; ++c) {
        s.own1[kC1 * k + c] = a1[c];
      }
    }
  }
  __syncthreads();
  treeSumT<kC1, true, kC1, 8>(fv.subSize, fv.loadedPos, fv.numLoaded, s.own1, s.sub1, J, wave, kWaves, lane, t.kRange);
  __syncthreads();
  // J^T w per column: the primary source slot, then the extras (slot numbering of phase F)
  for (int c = tid; c < NP; c += kT) {
    float a = 0.f;
    if (c < n) {
      auto slotShare = [&](int e) {
        const ColumnSourceDev cs = fd.srcs[e];
        return cs.weight * sourceGradient(cs.joint, cs.dof, cs.parent, s.js, s.sub1 + kC1 * cs.tin);
      };
      a = slotShare(c);
      const int e1 = fd.slotBase + fd.srcStart[c + 1];
      for (int e = fd.slotBase + fd.srcStart[c]; e < e1; ++e) {
        a += slotShare(e);
      }
      if (pb.M > pb.rowsJoint) { // limit / model-parameter rows, residual r - J d (fusedSolveKernel phase J)
        a += paramRowsColumn(rig, pb, fd, t.th, s.d0, t.col, P, b, c, fd.solveList[c]).g;
      }
      if (hasGen) {
        for (int r = 0; r < fd.genRows; ++r) {
          a += t.gJ[r * gst + c] * t.gRes[r];
        }
      }
      a -= lambda * s.d0[c];
    }
    rhoVec[size_t(b) * NP + c] = a;
  }
}

size_t treeStateFloats(int J, int U) {
  return treeStateLayout(J, U).total;
}
size_t treeRefineLdsBytes(int J, int P, int U, int n, int genRows) {
  return treeRefineLdsFloats(J, P, U, n, nullptr, nullptr, genRows) * sizeof(float);
}

hipError_t launchTreeRefine(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    const float* theta,
    const float* state,
    const float* genState,
    const float* dvec,
    float* rhoVec,
    const int32_t* refState,
    float lambda,
    const float* lambdaPer,
    hipStream_t stream) {
  const size_t lds = treeRefineLdsBytes(rig.J, rig.P, fd.U, fd.n, fd.genRows);
  if (lds > 160 * 1024 - 64) {
    return hipErrorInvalidValue;
  }
  const bool extra = pb.M > pb.rowsJoint || fd.GT > 0 || pb.instPosParent != nullptr || pb.instOriParent != nullptr;
  if (!extra) { // (cfg5, one box, per launch: four waves 0.83 ms, eight 0.64 ms, sixteen 0.80 ms)
    constexpr int kRefWaves = 8;
    static LdsLimitCache ldsLimitWide;
    hipError_t rc = ldsLimitWide.ensure(reinterpret_cast<const void*>(treeRefineKernel<kRefWaves>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
    hipLaunchKernelGGL((treeRefineKernel<kRefWaves>), dim3(pb.B), dim3(64 * kRefWaves), lds, stream, rig, pb, fd, theta, state, genState, dvec, rhoVec, refState, lambda, lambdaPer);
    return hipGetLastError();
  }
  static LdsLimitCache ldsLimit;
  {
    hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(treeRefineKernel<4>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
  }
  hipLaunchKernelGGL(treeRefineKernel<4>, dim3(pb.B), dim3(256), lds, stream, rig, pb, fd, theta, state, genState, dvec, rhoVec, refState, lambda, lambdaPer);
  return hipGetLastError();
}
#endif

// ---------------------------------------------------------------------------------------------
#if !defined(MMX_FUSED_GROUP) || MMX_FUSED_GROUP == 0
size_t fusedCsrFloats(int J, int nnz) {
  auto a4 = [](size_t x) { return (x + 3) & ~size_t(3); };
  return a4(7 * size_t(J) + 1) + 2 * a4(nnz);
}
size_t fusedLdsBytes(int NB, int J, int P, int U, int nsrc, int n, int numCells, bool cellsBehindRho, int GT, int genRows, bool separateUy, size_t csrFloats, bool mix) {
  return fusedLayout(NB, J, P, U, nsrc, n, numCells, cellsBehindRho, GT, genRows, separateUy, csrFloats, mix).total * sizeof(float);
}
#endif

template <int NB, int MODE, bool kTR, bool kGen = false, int kRule = -1>
static hipError_t launchFusedMode(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    float* theta,
    const SolveStateDev& st,
    const FusedParams& fp,
    float* dbgH,
    float* dbgG,
    long long* dbgClk,
    void* argsBuf,
    hipStream_t stream) {
  constexpr bool kFourL = FusedFour<NB, kTR, kGen, kRule>::value; // (fusedSolveKernel's kFour)
  const size_t lds = fusedLdsBytes(NB, rig.J, rig.P, fd.U, fd.nsrc, fd.n, fd.numCells, kRule < 0, kGen ? fd.GT : 0, kGen ? fd.genRows : 0, kTR, kFourL ? 0 : fusedCsrFloats(rig.J, fd.nnz));
  if (lds > 160 * 1024) {
    return hipErrorInvalidValue;
  }
  // the lazy-argument form (kArgLazy): four-workgroup production instantiations, shared rig and weights, a buffer to stash into
  constexpr bool kLazyBuilt = kFourL && MODE == 0;
  const bool lazy = kLazyBuilt && argsBuf != nullptr && rig.instPreRot == nullptr && rig.instOffset == nullptr && pb.fnWeights == nullptr;
  if (lazy) {
    static LdsLimitCache ldsLimit; // (one per instantiation)
    hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(fusedSolveKernel<NB, MODE, kTR, kGen, kRule, kLazyBuilt>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
    FusedArgs* dst = static_cast<FusedArgs*>(argsBuf);
    hipLaunchKernelGGL(stashFusedArgsKernel, dim3(1), dim3(1), 0, stream, FusedArgs{rig, pb, fd, st, fp}, dst);
    hipLaunchKernelGGL((fusedSolveKernel<NB, MODE, kTR, kGen, kRule, kLazyBuilt>), dim3(pb.B), dim3(256), lds, stream, dst, RigDev{}, ProblemDev{}, FusedDev{}, theta, SolveStateDev{}, FusedParams{}, dbgH, dbgG, dbgClk, MixSelect{});
  } else {
    static LdsLimitCache ldsLimit;
    hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(fusedSolveKernel<NB, MODE, kTR, kGen, kRule, false>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
    hipLaunchKernelGGL((fusedSolveKernel<NB, MODE, kTR, kGen, kRule, false>), dim3(pb.B), dim3(256), lds, stream, static_cast<const FusedArgs*>(nullptr), rig, pb, fd, theta, st, fp, dbgH, dbgG, dbgClk, MixSelect{});
  }
  return hipGetLastError();
}

template <int NB>
static hipError_t launchFusedNB(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    float* theta,
    const SolveStateDev& st,
    const FusedParams& fp,
    float* dbgH,
    float* dbgG,
    long long* dbgClk,
    void* argsBuf,
    hipStream_t stream) {
  if (fd.GT > 0) { // further joint error functions / ellipsoid limits: the production and the parity-dump instantiations only
    if (fp.stepRule == 2) {
      return hipErrorInvalidValue;
    }
    if (dbgH != nullptr || dbgG != nullptr) {
      return launchFusedMode<NB, 1, false, true>(rig, pb, fd, theta, st, fp, dbgH, dbgG, nullptr, argsBuf, stream);
    }
    return launchFusedMode<NB, 0, false, true>(rig, pb, fd, theta, st, fp, nullptr, nullptr, nullptr, argsBuf, stream);
  }
  if (fp.stepRule == 2) { // MMX_STEP_TRUST_REGION: its own instantiation (no clocks / parity dump in it)
    return launchFusedMode<NB, 0, true>(rig, pb, fd, theta, st, fp, nullptr, nullptr, nullptr, argsBuf, stream);
  }
  if (dbgClk != nullptr) {
    return launchFusedMode<NB, 2, false>(rig, pb, fd, theta, st, fp, dbgH, dbgG, dbgClk, argsBuf, stream);
  }
  if (dbgH != nullptr || dbgG != nullptr) {
    return launchFusedMode<NB, 1, false>(rig, pb, fd, theta, st, fp, dbgH, dbgG, dbgClk, argsBuf, stream);
  }
  if (pb.M == pb.rowsJoint) { // no parameter-space rows: the instantiations per step rule (kRule)
    if (fp.stepRule == 0 && fp.doLineSearch == 0) {
      return launchFusedMode<NB, 0, false, false, 0>(rig, pb, fd, theta, st, fp, nullptr, nullptr, nullptr, argsBuf, stream);
    }
    if (fp.stepRule == 1) {
      return launchFusedMode<NB, 0, false, false, 1>(rig, pb, fd, theta, st, fp, nullptr, nullptr, nullptr, argsBuf, stream);
    }
  }
  return launchFusedMode<NB, 0, false>(rig, pb, fd, theta, st, fp, dbgH, dbgG, dbgClk, argsBuf, stream);
}

// the mixed-precision instantiation (generic rule; by-value arguments)
template <int NB>
static hipError_t launchFusedMixedNB(const RigDev& rig, const ProblemDev& pb, const FusedDev& fd, float* theta, const SolveStateDev& st, const FusedParams& fp, const MixSelect& sel, int blocks, long long* dbgClk, hipStream_t stream) {
  const size_t lds = fusedLdsBytes(NB, rig.J, rig.P, fd.U, fd.nsrc, fd.n, fd.numCells, true, 0, 0, false, 0, true);
  if (lds > 160 * 1024) {
    return hipErrorInvalidValue;
  }
  if (NB == 6 && dbgClk != nullptr) { // profiling aid (MMX_PHASE_CLOCKS): the clocked instantiation, BASELINE configs[1]'s block count only
    static LdsLimitCache ldsLimitClk;
    hipError_t rc = ldsLimitClk.ensure(reinterpret_cast<const void*>(fusedSolveKernel<NB == 6 ? 6 : 1, 2, false, false, -1, false, true>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
    hipLaunchKernelGGL((fusedSolveKernel<NB == 6 ? 6 : 1, 2, false, false, -1, false, true>), dim3(blocks), dim3(256), lds, stream, static_cast<const FusedArgs*>(nullptr), rig, pb, fd, theta, st, fp, nullptr, nullptr, dbgClk, sel);
    return hipGetLastError();
  }
  static LdsLimitCache ldsLimit;
  hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(fusedSolveKernel<NB, 0, false, false, -1, false, true>), lds);
  if (rc != hipSuccess) {
    return rc;
  }
  hipLaunchKernelGGL((fusedSolveKernel<NB, 0, false, false, -1, false, true>), dim3(blocks), dim3(256), lds, stream, static_cast<const FusedArgs*>(nullptr), rig, pb, fd, theta, st, fp, nullptr, nullptr, nullptr, sel);
  return hipGetLastError();
}

// The instantiations are split over four translation units (build.py compiles this file once per
// MMX_FUSED_GROUP, in parallel): group g provides launchFusedGroup<g>() for its block counts.
#ifndef MMX_FUSED_GROUP
#define MMX_FUSED_GROUP 0
#endif

#define MMX_FUSED_ARGS \
  const RigDev &rig, const ProblemDev &pb, const FusedDev &fd, float *theta, const SolveStateDev &st, const FusedParams &fp, \
      float *dbgH, float *dbgG, long long *dbgClk, void *argsBuf, hipStream_t stream
#define MMX_FUSED_PASS rig, pb, fd, theta, st, fp, dbgH, dbgG, dbgClk, argsBuf, stream

hipError_t launchFusedGroup0(int nb, MMX_FUSED_ARGS);
hipError_t launchFusedGroup1(int nb, MMX_FUSED_ARGS);
hipError_t launchFusedGroup2(int nb, MMX_FUSED_ARGS);
hipError_t launchFusedGroup3(int nb, MMX_FUSED_ARGS);
#define MMX_MIXED_ARGS \
  const RigDev &rig, const ProblemDev &pb, const FusedDev &fd, float *theta, const SolveStateDev &st, const FusedParams &fp, const MixSelect &sel, int blocks, long long *dbgClk, hipStream_t stream
hipError_t launchFusedMixedGroup5(int nb, MMX_MIXED_ARGS);
hipError_t launchFusedMixedGroup6(int nb, MMX_MIXED_ARGS);
#define MMX_MIXED_CASE(NB_) \
  case NB_:                 \
    return launchFusedMixedNB<NB_>(rig, pb, fd, theta, st, fp, sel, blocks, dbgClk, stream);

#define MMX_CASE(NB_) \
  case NB_:           \
    return launchFusedNB<NB_>(MMX_FUSED_PASS);

#if MMX_FUSED_GROUP == 0
hipError_t launchFusedGroup0(int nb, MMX_FUSED_ARGS) {
  switch (nb) {
    MMX_CASE(1) MMX_CASE(2) MMX_CASE(3) default : return hipErrorInvalidValue;
  }
}
#elif MMX_FUSED_GROUP == 1
hipError_t launchFusedGroup1(int nb, MMX_FUSED_ARGS) {
  switch (nb) {
    MMX_CASE(4) MMX_CASE(5) MMX_CASE(6) default : return hipErrorInvalidValue;
  }
}
#elif MMX_FUSED_GROUP == 2
hipError_t launchFusedGroup2(int nb, MMX_FUSED_ARGS) {
  switch (nb) {
    MMX_CASE(7) MMX_CASE(8) MMX_CASE(10) default : return hipErrorInvalidValue;
  }
}
#elif MMX_FUSED_GROUP == 9 // (compile probe, scripts/probes/fused_one.sh: ONE instantiation, for register / spill figures in half a minute)
#ifndef MMX_PROBE_RULE
#define MMX_PROBE_RULE 0
#endif
template __global__ void fusedSolveKernel<6, 0, false, false, MMX_PROBE_RULE, true>(
    const FusedArgs*, RigDev, ProblemDev, FusedDev, float*, SolveStateDev, FusedParams, float*, float*, long long*, MixSelect);
#elif MMX_FUSED_GROUP == 8 // (compile probe of the mixed-precision instantiation, six blocks)
template __global__ void fusedSolveKernel<6, 0, false, false, -1, false, true>(
    const FusedArgs*, RigDev, ProblemDev, FusedDev, float*, SolveStateDev, FusedParams, float*, float*, long long*, MixSelect);
#elif MMX_FUSED_GROUP == 3
hipError_t launchFusedGroup3(int nb, MMX_FUSED_ARGS) {
  switch (nb) {
    MMX_CASE(12) MMX_CASE(14) default : return hipErrorInvalidValue;
  }
}
#elif MMX_FUSED_GROUP == 5
hipError_t launchFusedMixedGroup5(int nb, MMX_MIXED_ARGS) {
  switch (nb) {
    MMX_MIXED_CASE(1) MMX_MIXED_CASE(2) MMX_MIXED_CASE(3) MMX_MIXED_CASE(4) default : return hipErrorInvalidValue;
  }
}
#elif MMX_FUSED_GROUP == 6
hipError_t launchFusedMixedGroup6(int nb, MMX_MIXED_ARGS) {
  switch (nb) {
    MMX_MIXED_CASE(5) MMX_MIXED_CASE(6) MMX_MIXED_CASE(7) MMX_MIXED_CASE(8) default : return hipErrorInvalidValue;
  }
}
#endif
#undef MMX_CASE
#undef MMX_MIXED_CASE

#if MMX_FUSED_GROUP == 0
int fusedBlocksFor(int n) {
  const int nb = (n + 15) / 16;
  const int avail[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14};
  for (int a : avail) {
    if (nb <= a) {
      return a;
    }
  }
  return -1;
}

size_t fusedArgsBytes() {
  return sizeof(FusedArgs);
}

// MMX_PRECISION_MIXED: up to eight blocks (128 solved parameters), at most 256 joints (the tangent pass keeps a joint per thread)
bool fusedMixedUsable(int J, int P, int U, int nsrc, int n, int numCells) {
  const int nb = fusedBlocksFor(n);
  return nb >= 1 && nb <= 8 && J <= 256 && fusedLdsBytes(nb, J, P, U, nsrc, n, numCells, true, 0, 0, false, 0, true) <= 160 * 1024;
}
hipError_t launchFusedMixed(MMX_MIXED_ARGS) {
  const int nb = fusedBlocksFor(fd.n);
  if (nb < 1 || nb > 8) {
    return hipErrorInvalidValue;
  }
  return nb <= 4 ? launchFusedMixedGroup5(nb, rig, pb, fd, theta, st, fp, sel, blocks, dbgClk, stream) : launchFusedMixedGroup6(nb, rig, pb, fd, theta, st, fp, sel, blocks, dbgClk, stream);
}

hipError_t launchFusedSolve(MMX_FUSED_ARGS) {
  const int nb = fusedBlocksFor(fd.n);
  if (nb < 0) {
    return hipErrorInvalidValue;
  }
  if (nb <= 3) {
    return launchFusedGroup0(nb, MMX_FUSED_PASS);
  }
  if (nb <= 6) {
    return launchFusedGroup1(nb, MMX_FUSED_PASS);
  }
  if (nb <= 10) {
    return launchFusedGroup2(nb, MMX_FUSED_PASS);
  }
  return launchFusedGroup3(nb, MMX_FUSED_PASS);
}
#endif

} // namespace mmx
