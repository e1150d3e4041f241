// mmx_f64.hip -- the double-precision instantiation of the solve: GaussNewtonSolverT<double> + SolverT<double>::solve
// (momentum/solver/gauss_newton_solver.cpp:315-316 instantiates both; pymomentum/tensor_ik/tensor_ik.cpp is
// templated on T) for every element of the batch, behind mmx_solve_f64.
//
// Written for exactness and brevity, not for the roofline: one workgroup per instance, forward kinematics by
// tree level in double, the dense J (solved columns only), H = J^T J + lambda I and its Cholesky factor in a
// global scratch of the problem (L2-resident per workgroup), everything the reference's double solver does in
// the reference's order of operations (normal equations from the explicit Jacobian, LL^T, two substitutions,
// theta -= delta, both backtracking rules, the LM schedule).  No refinement step is needed here: the factor is
// exact to double rounding.  Joint constants stay float like in the reference (JointT is float even for double
// solves, skeleton_state.cpp:89) and are widened per use.
//
// Scope: position and orientation constraints with their GeneralizedLoss, per-instance characters and constraint
// parents, enabled-parameter sets, per-element error-function weights, and the parameter-space rows -- LimitErrorFunctionT
// <double> for the limit types on model / joint parameters and ModelParametersErrorFunctionT<double> -- whose few
// non-zeros per row go straight into H and g (the products J^T J would form from them, in double).  The further joint
// error functions and ellipsoid limits are single-precision only (mmx_solve_f64 returns MMX_ERR_UNSUPPORTED for them).
#include "mmx_device.hpp"
#include "mmx_device_d.hpp"
#include "mmx_kernels.hpp"

#include <cfloat>

namespace mmx {

namespace {

// Every working array of these kernels lives in LDS; the pointers say so in their TYPE (address space 3), so that each access
// is a ds_read / ds_write wherever it sits -- in a helper that was not inlined, behind a struct that went to scratch.  (Round 4:
// with generic pointers the resident kernel had 422 flat loads for 306 LDS reads; a flat access takes the long way through
// the vector memory path and counts on both wait counters.)
typedef __attribute__((address_space(3))) double ldsd;
typedef __attribute__((address_space(3))) int ldsi;

constexpr int kDs = 17; // doubles per joint: world t(3) q(4) s(1) | rotation axes x, y, z (9)

struct F64Lds {
  ldsd* th; // [P]
  ldsd* trial; // [P]
  ldsd* jp; // [7 J]
  ldsd* js; // [kDs J]
  ldsd* uv; // [3 U] unit world vector
  ldsd* ur; // [M] scaled residual rows: 3 U of the position / orientation blocks, then the further joint error functions'
  ldsd* us; // [U] derivScale
  ldsi* utin; // [U]
  ldsd* g; // [n]
  ldsd* d; // [n]
  ldsd* red; // [8]
  ldsi* flags; // [4]
  ldsi* colOf; // [P] solve column of a model parameter, or -1
  ldsd* jl; // [n][rc + 1] a chunk of J's rows, column-major (normal equations)
  ldsd* gev; // [G][kGevD] the further joint error functions' evaluations (JointEvalD, rows scaled by sigma)
  // ---- the resident instantiation (kRes): the system never leaves LDS
  ldsd* H; // [n (n + 1) / 2] lower triangle of H, then of its factor, packed by columns (hpos)
  ldsd* invd; // [n] 1 / L(k,k)
  ldsd* w1; // [n] work vectors of the substitutions
  ldsd* w2; // [n]
  ldsi* srcTab; // the solved columns' sources packed into LDS ([n + 1] offsets, then three words per source), or null
};

// position of H(i, j), i >= j, in the packed lower triangle (column j holds rows j .. n-1)
__device__ __forceinline__ int hpos(int n, int i, int j) {
  return j * n - ((j * (j - 1)) >> 1) + (i - j);
}

// ParameterTransformT<double>::apply + SkeletonStateT<double>::set (parameter_transform.cpp:110-124,
// skeleton_state.cpp:87-121, joint_state.cpp:22-65): joint parameters one transform row per thread, then one
// tree level per barrier (parents before children), then the rotation axes of all joints at once.
__device__ void fkF64(const RigDev& rig, const F64Lds& s, const ldsd* th, int tid, bool withAxes) {
  for (int r = tid; r < rig.R; r += 256) {
    double acc = 0.0;
    const int k1 = rig.ptOuter[r + 1];
    for (int k = rig.ptOuter[r]; k < k1; ++k) {
      acc += double(rig.ptValue[k]) * th[rig.ptInner[k]];
    }
    s.jp[r] = acc + double(rig.ptOffsets[r]);
  }
  __syncthreads();
  // the theta-only part of JointStateT<double>::set for all joints at once (the six sin / cos of a joint are the bulk of
  // the arithmetic: inside the level sweep every tree level paid for them in turn): local (t, q, s) into the joint's
  // world slot, the partial rotations pre * Rz, pre * Rz * Ry into the slots of the y / x axes (8 ..15)
  for (int j = tid; j < rig.J; j += 256) {
    const ldsd* p = s.jp + 7 * j;
    const float* pre = rig.preRot + 4 * j;
    const float* off = rig.offset + 3 * j;
    double sx, cx, sy, cy, sz, cz;
    sincos(0.5 * p[3], &sx, &cx);
    sincos(0.5 * p[4], &sy, &cy);
    sincos(0.5 * p[5], &sz, &cz);
    const DQ q0{double(pre[0]), double(pre[1]), double(pre[2]), double(pre[3])};
    const DQ q1 = dqmul(q0, DQ{0.0, 0.0, sz, cz});
    const DQ q2 = dqmul(q1, DQ{0.0, sy, 0.0, cy});
    const DQ ql = dqmul(q2, DQ{sx, 0.0, 0.0, cx});
    ldsd* o = s.js + kDs * j;
    o[0] = double(off[0]) + p[0], o[1] = double(off[1]) + p[1], o[2] = double(off[2]) + p[2];
    o[3] = ql.x, o[4] = ql.y, o[5] = ql.z, o[6] = ql.w, o[7] = exp2(p[6]);
    o[8] = q1.x, o[9] = q1.y, o[10] = q1.z, o[11] = q1.w, o[12] = q2.x, o[13] = q2.y, o[14] = q2.z, o[15] = q2.w;
  }
  __syncthreads();
  for (int lvl = 0; lvl < rig.numLevels; ++lvl) { // parents before children (skeleton_state.cpp:100-121)
    const int i1 = rig.levelStart[lvl + 1];
    for (int i = rig.levelStart[lvl] + tid; i < i1; i += 256) {
      const int j = rig.levelOrder[i];
      const float* pre = rig.preRot + 4 * j;
      ldsd* o = s.js + kDs * j;
      const DQ q0{double(pre[0]), double(pre[1]), double(pre[2]), double(pre[3])};
      const DQ q1{o[8], o[9], o[10], o[11]}, q2{o[12], o[13], o[14], o[15]};
      DQ qp{0.0, 0.0, 0.0, 1.0};
      const int par = rig.parent[j];
      if (par >= 0) {
        const ldsd* w = s.js + kDs * par;
        const D3 tp{w[0], w[1], w[2]};
        qp = DQ{w[3], w[4], w[5], w[6]};
        const D3 t = tp + dqrot(qp, w[7] * D3{o[0], o[1], o[2]}); // transform.h:124-129
        const DQ q = dqmul(qp, DQ{o[3], o[4], o[5], o[6]});
        o[0] = t.x, o[1] = t.y, o[2] = t.z, o[3] = q.x, o[4] = q.y, o[5] = q.z, o[6] = q.w, o[7] = w[7] * o[7];
      }
      if (withAxes) { // rotationAxis.col(i) = (q_parent * q_partial) * e_i (joint_state.cpp:53-54)
        const D3 az = dqrot(dqmul(qp, q0), D3{0.0, 0.0, 1.0});
        const D3 ay = dqrot(dqmul(qp, q1), D3{0.0, 1.0, 0.0});
        const D3 ax = dqrot(dqmul(qp, q2), D3{1.0, 0.0, 0.0});
        o[8] = ax.x, o[9] = ax.y, o[10] = ax.z, o[11] = ay.x, o[12] = ay.y, o[13] = ay.z, o[14] = az.x, o[15] = az.y, o[16] = az.z;
      }
    }
    __syncthreads();
  }
}

// Position / Orientation evalFunction + the weighting of JointErrorFunctionT::getJacobian
// (position_error_function.cpp:15-27, orientation_error_function.cpp:15-40, joint_error_function-inl.h:197-213)
// for unit u of instance b; returns the unit's share of the error (w * loss(|f|^2), once per constraint).
__device__ double evalUnitF64(const ProblemDev& pb, const F64Lds& s, int b, int u, bool store) {
  const UnitInput in = loadUnitInput(pb, b, u);
  const ldsd* w = s.js + kDs * in.joint;
  const D3 t{w[0], w[1], w[2]};
  const DQ q{w[3], w[4], w[5], w[6]};
  const bool isPoint = u < pb.Kp;
  D3 v, f;
  double sqr, fw;
  bool first = true;
  const LossDev& ls = isPoint ? pb.lossPos : pb.lossOri;
  if (isPoint) {
    v = t + dqrot(q, w[7] * D3{double(in.a[0]), double(in.a[1]), double(in.a[2])});
    f = v - D3{double(in.t[0]), double(in.t[1]), double(in.t[2])};
    sqr = ddot(f, f);
    fw = double(pb.wPos);
  } else {
    const int uo = u - pb.Kp, k = uo - 3 * (uo / 3);
    // OrientationDataT<double>'s constructor normalises in double (orientation_error_function.h:33-35)
    const DQ qo = dqnormalized(DQ{double(in.a[0]), double(in.a[1]), double(in.a[2]), double(in.a[3])});
    const DQ qt = dqnormalized(DQ{double(in.t[0]), double(in.t[1]), double(in.t[2]), double(in.t[3])});
    v = dqrot(q, dqmatCol(qo, k));
    f = v - dqmatCol(qt, k);
    sqr = 0.0;
    for (int kk = 0; kk < 3; ++kk) { // the loss sees all nine rows of the constraint
      const D3 fo = dqrot(q, dqmatCol(qo, kk)) - dqmatCol(qt, kk);
      sqr += ddot(fo, fo);
    }
    first = k == 0;
    fw = double(pb.wOri);
  }
  double sigma = 0.0, werr = 0.0;
  if (in.cw != 0.f && fw > 0.0) {
    const double wgt = double(in.cw) * fw;
    werr = first ? wgt * dlossValue(ls, sqr) : 0.0;
    sigma = sqrt(wgt * dlossDeriv(ls, sqr));
  }
  if (store) {
    s.uv[3 * u] = v.x, s.uv[3 * u + 1] = v.y, s.uv[3 * u + 2] = v.z;
    s.ur[3 * u] = sigma * f.x, s.ur[3 * u + 1] = sigma * f.y, s.ur[3 * u + 2] = sigma * f.z;
    s.us[u] = sigma;
    s.utin[u] = in.tin;
  }
  return werr;
}

__device__ double blockSumF64(const F64Lds& s, double v, int tid) {
  for (int off = 32; off > 0; off >>= 1) {
    v += __shfl_xor(v, off, 64);
  }
  __syncthreads();
  if ((tid & 63) == 0) {
    s.red[tid >> 6] = v;
  }
  __syncthreads();
  return (s.red[0] + s.red[1]) + (s.red[2] + s.red[3]);
}

// this thread's share of the error of the parameter-space blocks at `th`: LimitErrorFunctionT<double> (the limit types
// on model / joint parameters, limit_error_function.cpp:992-1122) and ModelParametersErrorFunctionT<double>
// (model_parameters_error_function.cpp:44-131).  kJacobianRows: the value getJacobian returns (model rows with weight
// <= 0 are skipped, :113), else the one getError returns (:54-58).
template <bool kJacobianRows>
__device__ double paramRowsErrorF64(const RigDev& rig, const ProblemDev& pb, const ldsd* th, int b, int tid) {
  double e = 0.0;
  if (pb.NL > 0 && pb.wLimit > 0.f) {
    const double tWeight = double(1e+1f * pb.wLimit); // kLimitWeight * weight_ (a float product in both instantiations)
    for (int l = tid; l < pb.NL; l += 256) {
      e += evalLimit<double>(rig, pb.limits[l], (const double*)th, pb.enabledMask, tWeight).err; // (cold path: generic pointer)
    }
  }
  if (pb.hasModel && pb.wModel > 0.f) {
    const float* tp = pb.mpTarget + size_t(b) * rig.P;
    const float* tw = pb.mpWeights + size_t(b) * rig.P;
    double em = 0.0;
    for (int i = tid; i < rig.P; i += 256) {
      if (pb.enabledMask[i] != 0) {
        const double w = double(tw[i]);
        if (!kJacobianRows || w > 0.0) {
          const double pd = w * (th[i] - double(tp[i]));
          em += pd * pd;
        }
      }
    }
    e += em * double(pb.wModel) * 1e-1; // kMotionWeight (T(1e-1), model_parameters_error_function.h:61)
  }
  return e;
}

// SkeletonSolverFunctionT<double>::getError (skeleton_solver_function.cpp:64-83; rounded through float, :82)
// The further JointErrorFunctionT<double> specialisations (Plane / HalfPlane, AimDist / AimDir, FixedAxisDiff / Cos /
// Angle, Normal): evalFunction + the weighting of getJacobian in double -- evalJointConstraint of mmx_device.hpp with T =
// double (the constraint data stay float like the reference's tensors; joint_error_function-inl.h:197-226).
struct JointEvalD {
  D3 vp, vn;
  double dp[9], dn[9];
  double f[3];
  double sigma, werr;
  int nrows;
  bool hasPoint, hasDir;
};
constexpr int kGevD = 26; // doubles per constraint in LDS: vp(3) vn(3) sigma dp(9) sigma dn(9) | tin, row | flags, -

__device__ __forceinline__ D3 dnormalizedOrSame(D3 a) { // Eigen normalized(): unchanged when the norm is zero
  const double n2 = ddot(a, a);
  return n2 > 0.0 ? (1.0 / sqrt(n2)) * a : a;
}

__device__ JointEvalD evalJointConstraintF64(const JointBlockDev& k, const ldsd* js, int joint, size_t c) {
  JointEvalD o;
  o.vp = o.vn = D3{0.0, 0.0, 0.0};
  for (int i = 0; i < 9; ++i) {
    o.dp[i] = o.dn[i] = 0.0;
  }
  o.f[0] = o.f[1] = o.f[2] = 0.0;
  o.sigma = o.werr = 0.0;
  o.nrows = jointBlockFuncDim(k.type);
  o.hasPoint = k.type != MMX_JC_FIXED_AXIS_DIFF && k.type != MMX_JC_FIXED_AXIS_COS && k.type != MMX_JC_FIXED_AXIS_ANGLE;
  o.hasDir = k.type != MMX_JC_PLANE && k.type != MMX_JC_HALF_PLANE;
  const ldsd* w = js + kDs * joint;
  const D3 t{w[0], w[1], w[2]};
  const DQ q{w[3], w[4], w[5], w[6]};
  const double sc = w[7];
  auto vec = [&](const float* a) { return D3{double(a[3 * c]), double(a[3 * c + 1]), double(a[3 * c + 2])}; };
  auto setRow = [](double* m, D3 a, double f) { m[0] = f * a.x, m[1] = f * a.y, m[2] = f * a.z; };
  auto addOuter = [](double* m, D3 a, D3 b, double f) { // m += f * a b^T
    m[0] += f * a.x * b.x, m[1] += f * a.x * b.y, m[2] += f * a.x * b.z;
    m[3] += f * a.y * b.x, m[4] += f * a.y * b.y, m[5] += f * a.y * b.z;
    m[6] += f * a.z * b.x, m[7] += f * a.z * b.y, m[8] += f * a.z * b.z;
  };
  const D3 gl = vec(k.global);
  if (o.hasPoint) {
    o.vp = t + dqrot(q, sc * vec(k.localPoint)); // state.transform * point
  }
  if (o.hasDir) {
    o.vn = dqrot(q, dnormalizedOrSame(vec(k.localDir))); // state.rotation() * dir, normalised by the data ctor
  }
  switch (k.type) {
    case MMX_JC_PLANE:
    case MMX_JC_HALF_PLANE: { // plane_error_function.cpp:52-71
      const D3 nrm = dnormalizedOrSame(gl);
      double val = ddot(o.vp, nrm) - double(k.planeD[c]);
      const bool half = k.type == MMX_JC_HALF_PLANE;
      if (half && val > 0.0) {
        val = 0.0;
      }
      o.f[0] = val;
      if (!half || val < 0.0) {
        setRow(o.dp, nrm, 1.0);
      }
      break;
    }
    case MMX_JC_AIM_DIST: { // aim_error_function.cpp:15-36
      const D3 tgt = gl - o.vp;
      const double proj = ddot(o.vn, tgt);
      const D3 r = proj * o.vn - tgt;
      o.f[0] = r.x, o.f[1] = r.y, o.f[2] = r.z;
      o.dp[0] = o.dp[4] = o.dp[8] = 1.0;
      addOuter(o.dp, o.vn, o.vn, -1.0);
      addOuter(o.dn, o.vn, tgt, 1.0);
      o.dn[0] += proj, o.dn[4] += proj, o.dn[8] += proj;
      break;
    }
    case MMX_JC_AIM_DIR: { // aim_error_function.cpp:39-67
      const D3 tgt = gl - o.vp;
      const double nrm = sqrt(ddot(tgt, tgt));
      D3 dir{0.0, 0.0, 0.0};
      if (nrm > 1e-16) {
        dir = (1.0 / nrm) * tgt;
        addOuter(o.dp, dir, dir, -1.0 / nrm);
        o.dp[0] += 1.0 / nrm, o.dp[4] += 1.0 / nrm, o.dp[8] += 1.0 / nrm;
      }
      const D3 r = o.vn - dir;
      o.f[0] = r.x, o.f[1] = r.y, o.f[2] = r.z;
      o.dn[0] = o.dn[4] = o.dn[8] = 1.0;
      break;
    }
    case MMX_JC_FIXED_AXIS_DIFF: { // fixed_axis_error_function.cpp:15-26
      const D3 r = o.vn - dnormalizedOrSame(gl);
      o.f[0] = r.x, o.f[1] = r.y, o.f[2] = r.z;
      o.dn[0] = o.dn[4] = o.dn[8] = 1.0;
      break;
    }
    case MMX_JC_FIXED_AXIS_COS: { // :28-39
      const D3 ga = dnormalizedOrSame(gl);
      o.f[0] = 1.0 - ddot(o.vn, ga);
      setRow(o.dn, ga, -1.0);
      break;
    }
    case MMX_JC_FIXED_AXIS_ANGLE: { // :41-66
      const D3 ga = dnormalizedOrSame(gl);
      const double d = ddot(o.vn, ga);
      o.f[0] = acos(fmin(fmax(d, -1.0), 1.0));
      const double sine = sqrt(1.0 - d * d);
      if (sine > 1e-9) {
        setRow(o.dn, ga, -1.0 / sine);
      }
      break;
    }
    default: { // MMX_JC_NORMAL, normal_error_function.cpp:14-31
      const D3 dist = o.vp - gl;
      o.f[0] = ddot(o.vn, dist);
      setRow(o.dp, o.vn, 1.0);
      setRow(o.dn, dist, 1.0);
      break;
    }
  }
  const double cw = double(k.weight[c]);
  if (cw != 0.0 && k.fw > 0.f) { // :197-199 ; blocks with weight_ <= 0 are skipped (skeleton_solver_function.cpp:223-231)
    const double sqr = o.f[0] * o.f[0] + o.f[1] * o.f[1] + o.f[2] * o.f[2];
    const double wgt = cw * double(k.fw);
    o.werr = wgt * dlossValue(k.loss, sqr); // :207
    o.sigma = sqrt(wgt * dlossDeriv(k.loss, sqr)); // :208
  }
  return o;
}

// computeEllipsoidError / the point and weight of computeEllipsoidJacobian in double (evalEllipsoid of mmx_device.hpp;
// limit_error_function.cpp:173-195,702-737)
struct EllipsoidEvalD {
  D3 position, diff;
  double jwgt, werr;
};
__device__ EllipsoidEvalD evalEllipsoidF64(const EllipsoidDev& ct, const ldsd* js, float wLimit) {
  const ldsd* wp = js + kDs * ct.parent;
  const ldsd* we = js + kDs * ct.ellipsoidParent;
  const D3 te{we[0], we[1], we[2]};
  const DQ qe{we[3], we[4], we[5], we[6]};
  auto affine = [](const float* a, D3 p) { // 3 x 4 row-major
    return D3{
        double(a[0]) * p.x + double(a[1]) * p.y + double(a[2]) * p.z + double(a[3]),
        double(a[4]) * p.x + double(a[5]) * p.y + double(a[6]) * p.z + double(a[7]),
        double(a[8]) * p.x + double(a[9]) * p.y + double(a[10]) * p.z + double(a[11])};
  };
  EllipsoidEvalD o;
  o.position = D3{wp[0], wp[1], wp[2]} + dqrot(DQ{wp[3], wp[4], wp[5], wp[6]}, wp[7] * D3{double(ct.offset[0]), double(ct.offset[1]), double(ct.offset[2])});
  const D3 local = (1.0 / we[7]) * dqrot(DQ{-qe.x, -qe.y, -qe.z, qe.w}, o.position - te); // transform.inverse() * position
  const D3 nrm = dnormalizedOrSame(affine(ct.ellipsoidInv, local));
  const D3 proj = affine(ct.ellipsoid, nrm);
  o.diff = o.position - (te + dqrot(qe, we[7] * proj));
  o.jwgt = o.werr = 0.0;
  if (wLimit > 0.f) { // a block with weight_ <= 0 is skipped; its rows stay zero
    const double w = (10.0 * double(wLimit)) * double(1e-4f) * double(ct.weight); // kLimitWeight * weight_ * kPositionWeight * limit.weight
    o.werr = w * ddot(o.diff, o.diff);
    o.jwgt = sqrt(w);
  }
  return o;
}

// this thread's share of the further joint error functions' error at the state in s.js
__device__ double jointBlocksErrorF64(const ProblemDev& pb, const F64Lds& s, int b, int tid) {
  double e = 0.0;
  for (int g = tid; g < pb.G; g += 256) {
    const JointBlockDev k = jointBlockOf(pb, b, pb.genBlock[g]);
    e += evalJointConstraintF64(k, s.js, pb.genJoint[g], size_t(b) * size_t(k.count) + size_t(g - k.first)).werr;
  }
  for (int q = tid; q < pb.NE; q += 256) { // LimitType::Ellipsoid entries of the limit block
    e += evalEllipsoidF64(pb.ellipsoids[q], s.js, pb.wLimit).werr;
  }
  return e;
}

__device__ double errorF64(const RigDev& rig, const ProblemDev& pb, const F64Lds& s, const ldsd* th, int b, int tid) {
  fkF64(rig, s, th, tid, false);
  double e = 0.0;
  for (int u = tid; u < pb.U; u += 256) {
    e += evalUnitF64(pb, s, b, u, false);
  }
  e += paramRowsErrorF64<false>(rig, pb, th, b, tid);
  e += jointBlocksErrorF64(pb, s, b, tid);
  return double(float(blockSumF64(s, e, tid)));
}

// d(unit vector) / d(joint parameter (joint, dof)) (joint_error_function-inl.h:248-291, joint_state.cpp:68-82)
__device__ __forceinline__ D3 sourceDerivativeF64(const ColumnSourceDev& c, const ldsd* js, D3 v, int utin, bool isPoint, bool& applies) {
  const bool anc = c.tin <= utin && utin < c.tout;
  const ldsd* a = js + kDs * c.joint;
  if (c.dof >= 3 && c.dof < 6) {
    const ldsd* ax = a + 8 + 3 * (c.dof - 3);
    applies = anc;
    return dcross(D3{ax[0], ax[1], ax[2]}, isPoint ? v - D3{a[0], a[1], a[2]} : v);
  }
  applies = anc && isPoint;
  if (c.dof < 3) {
    if (c.parent < 0) {
      return D3{c.dof == 0 ? 1.0 : 0.0, c.dof == 1 ? 1.0 : 0.0, c.dof == 2 ? 1.0 : 0.0};
    }
    const ldsd* p = js + kDs * c.parent;
    return p[7] * dqmatCol(DQ{p[3], p[4], p[5], p[6]}, c.dof);
  }
  return 0.693147180559945309417232121458176568 * (v - D3{a[0], a[1], a[2]});
}

// ---- the resident instantiation's three hot routines, each compiled on its own (not inlined: inside the kernel, whose
// register file is full of the state of everything else, they were allocated around ~140 spilled registers)
// g += J^T r, H += J^T J for the rows of J in jl (column-major, ldj doubles per column, rows padded with zeros to a multiple
// of four).  H on the matrix cores: a 16 x 16 tile of the lower triangle takes one v_mfma_f64_16x16x4_f64 per four rows --
// lane l feeds A[l % 16][l / 16] = J[row 4 s + l / 16][column 16 I + l % 16], likewise B for block column Jc, and register r
// of lane l comes back as C[4 r + l / 16][l % 16] (measured: scripts/probes/mfma_f64_layout.hip; NOT the single-precision
// 16x16x4 layout) -- and is added to the packed triangle once per chunk.  gfx950 issues the f64 matrix instruction at
// 64 cycles per wave-instruction, HALF the rate of the f32 16x16x4 one (scripts/probes/mfma_f64_rate.hip).
// The rows of J of the position / orientation units u0 .. u0 + nu - 1 into jl (column-major, ldj doubles per column, rows
// padded with zeros to a multiple of four): a thread per entry (column, unit).  The column's sources come from the table
// staged in LDS (srcTab; F64Lds) -- from L2 (solveList -> colStart -> colSources: three dependent round trips per entry,
// one more per further source) when it did not fit -- and the ancestor test comes first: five of six (source, unit)
// pairs of the 72-joint rig fail it and cost two LDS reads.  Compiled on its own like the routines below (inlined into
// the kernel the same loop ran on registers spilled to scratch).
__device__ __noinline__ void residentAssembleUnits(
    const ldsi* srcTab, const ColumnSourceDev* __restrict__ colSources, const int32_t* __restrict__ colStart, const int32_t* __restrict__ solveList,
    const ldsd* js, const ldsd* uv, const ldsd* us, const ldsi* utin, ldsd* jl, int ldj, int n, int u0, int nu, int Kp, int tid) {
  for (int item = tid; item < n * nu; item += 256) {
    const int c = item / nu, u = u0 + (item - c * nu);
    const int ut = utin[u];
    const bool isPoint = u < Kp;
    const D3 v{uv[3 * u], uv[3 * u + 1], uv[3 * u + 2]};
    const double sc = us[u];
    D3 acc{0.0, 0.0, 0.0};
    auto add = [&](const ColumnSourceDev& cs) { // jac.col(p) += derivScale * dfdv * jc * value (joint_error_function-inl.h:254-289)
      bool applies;
      const D3 gq = sourceDerivativeF64(cs, js, v, ut, isPoint, applies);
      if (applies) {
        const double w = double(cs.weight);
        acc.x += (sc * gq.x) * w, acc.y += (sc * gq.y) * w, acc.z += (sc * gq.z) * w;
      }
    };
    if (srcTab != nullptr) {
      const ldsi* rec = srcTab + n + 1;
      const int k1 = srcTab[c + 1];
      for (int k = srcTab[c]; k < k1; ++k) {
        const int w1 = rec[3 * k + 1];
        const int tin = w1 & 0xffff, tout = int(unsigned(w1) >> 16);
        if (!(tin <= ut && ut < tout)) {
          continue; // the source's joint is not an ancestor of the unit's joint
        }
        const int w0 = rec[3 * k];
        add(ColumnSourceDev{w0 & 0xfff, (w0 >> 12) & 7, tin, tout, (w0 >> 15) - 1, __int_as_float(rec[3 * k + 2])});
      }
    } else {
      const int p = solveList[c];
      const int e1 = colStart[p + 1];
      for (int k = colStart[p]; k < e1; ++k) {
        const ColumnSourceDev cs = colSources[k];
        if (cs.tin <= ut && ut < cs.tout) {
          add(cs);
        }
      }
    }
    ldsd* o = jl + c * ldj + 3 * (u - u0);
    o[0] = acc.x, o[1] = acc.y, o[2] = acc.z;
  }
  const int pad = (4 - (3 * nu) % 4) % 4; // rows up to a multiple of four: zeros (the matrix cores take four at a time)
  for (int idx = tid; idx < n * pad; idx += 256) {
    const int c = idx / pad;
    jl[c * ldj + 3 * nu + (idx - c * pad)] = 0.0;
  }
}
// The same rows from the host's assembly list (F64AssemblyList): the chunk is zeroed, then a thread takes whole entries
// (column, unit) that HAVE an applicable source -- perfectly balanced, no ancestor test, two entries' list words requested
// together.  Needs the packed source table (srcTab).
__device__ __noinline__ void residentAssembleUnitsList(
    const ldsi* srcTab, const uint2* __restrict__ groups, const int32_t* __restrict__ extra, int g0, int g1, const ldsd* js, const ldsd* uv,
    const ldsd* us, const ldsi* utin, ldsd* jl, int ldj, int n, int u0, int nu, int Kp, int tid) {
  const int rows4 = (3 * nu + 3) & ~3;
  for (int c = tid >> 2; c < n; c += 64) { // four threads per column (no division by the runtime row count)
    ldsd* col = jl + c * ldj;
    for (int r = tid & 3; r < rows4; r += 4) {
      col[r] = 0.0;
    }
  }
  __syncthreads();
  const ldsi* rec = srcTab + n + 1;
  for (int base = g0; base < g1; base += 512) {
    uint2 w[2];
    bool valid[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int g = base + 256 * e + tid;
      valid[e] = g < g1;
      w[e] = groups[valid[e] ? g : g0];
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (!valid[e]) {
        continue;
      }
      const int c = int(w[e].x & 0xfffu), ul = int((w[e].x >> 12) & 0x3fu), count = int(w[e].x >> 18);
      const int u = u0 + ul;
      const int ut = utin[u];
      const bool isPoint = u < Kp;
      const D3 v{uv[3 * u], uv[3 * u + 1], uv[3 * u + 2]};
      const double sc = us[u];
      D3 acc{0.0, 0.0, 0.0};
      for (int si = 0; si < count; ++si) {
        const int k = count == 1 ? int(w[e].y) : extra[int(w[e].y) + si];
        const int w0 = rec[3 * k], w1 = rec[3 * k + 1];
        const ColumnSourceDev cs{w0 & 0xfff, (w0 >> 12) & 7, w1 & 0xffff, int(unsigned(w1) >> 16), (w0 >> 15) - 1, __int_as_float(rec[3 * k + 2])};
        bool applies;
        const D3 gq = sourceDerivativeF64(cs, js, v, ut, isPoint, applies);
        if (applies) { // jac.col(p) += derivScale * dfdv * jc * value (joint_error_function-inl.h:254-289)
          const double wt = double(cs.weight);
          acc.x += (sc * gq.x) * wt, acc.y += (sc * gq.y) * wt, acc.z += (sc * gq.z) * wt;
        }
      }
      ldsd* o = jl + c * ldj + 3 * ul;
      o[0] = acc.x, o[1] = acc.y, o[2] = acc.z;
    }
  }
}
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __noinline__ void residentAccumulate(const ldsd* jl, int ldj, const ldsd* ur, int rows, ldsd* g, ldsd* H, int n, int tid) {
  const int rows4 = (rows + 3) & ~3; // (the pad rows of jl are zero; ur is read past `rows` only against those zeros)
  for (int c = tid; c < n; c += 256) {
    const ldsd* col = jl + c * ldj;
    double acc = g[c];
    for (int r = 0; r < rows4; r += 4) { // four rows per trip: eight reads in flight
      const double j0 = col[r], j1 = col[r + 1], j2 = col[r + 2], j3 = col[r + 3];
      const double r0 = ur[r], r1 = r + 1 < rows ? ur[r + 1] : 0.0, r2 = r + 2 < rows ? ur[r + 2] : 0.0, r3 = r + 3 < rows ? ur[r + 3] : 0.0;
      acc += j0 * r0;
      acc += j1 * r1;
      acc += j2 * r2;
      acc += j3 * r3;
    }
    g[c] = acc;
  }
  const int NB = (n + 15) >> 4, T = NB * (NB + 1) / 2, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, k = lane >> 4, steps = rows4 >> 2;
  // the wave's tiles (row-major order of the lower triangle, round robin), two at a time: two independent accumulator
  // chains, and per trip the operands of four steps (= sixteen rows) of both tiles are in flight together
  auto tileOf = [](int t, int& I, int& Jc) {
    I = int((sqrtf(8.f * float(t) + 1.f) - 1.f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= t) {
      ++I;
    }
    while (I * (I + 1) / 2 > t) {
      --I;
    }
    Jc = t - I * (I + 1) / 2;
  };
  for (int t0 = wave; t0 < T; t0 += 8) {
    const int t1 = t0 + 4;
    const bool two = t1 < T;
    int I0, J0, I1, J1;
    tileOf(t0, I0, J0);
    tileOf(two ? t1 : t0, I1, J1);
    const ldsd* pa0 = jl + (16 * I0 + i < n ? 16 * I0 + i : n - 1) * ldj + k;
    const ldsd* pb0 = jl + (16 * J0 + i < n ? 16 * J0 + i : n - 1) * ldj + k;
    const ldsd* pa1 = jl + (16 * I1 + i < n ? 16 * I1 + i : n - 1) * ldj + k;
    const ldsd* pb1 = jl + (16 * J1 + i < n ? 16 * J1 + i : n - 1) * ldj + k;
    v4d c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
    for (int s0 = 0; s0 < steps; s0 += 4) {
      double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int st = s0 + q < steps ? s0 + q : steps - 1;
        a0[q] = pa0[4 * st], b0[q] = pb0[4 * st], a1[q] = pa1[4 * st], b1[q] = pb1[4 * st];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (s0 + q < steps) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[q], b0[q], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[q], b1[q], c1, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row0 = 16 * I0 + 4 * r + k, col0 = 16 * J0 + i;
      if (row0 < n && col0 <= row0) {
        H[hpos(n, row0, col0)] += c0[r];
      }
      const int row1 = 16 * I1 + 4 * r + k, col1 = 16 * J1 + i;
      if (two && row1 < n && col1 <= row1) {
        H[hpos(n, row1, col1)] += c1[r];
      }
    }
  }
}
// The whole sweep over the position / orientation units for systems of at most 4 kTW tiles (n <= 96 with kTW = 6): chunk by
// chunk residentAssembleUnits, g += J^T r, and H's tiles accumulated IN REGISTERS across the chunks (a wave owns tiles
// t = wave, wave + 4, ...; eight registers each) -- added to the packed triangle once per iteration instead of once per
// chunk (the read-modify-write of the tiles and the tile bookkeeping were a quarter of the accumulation's 30 k cycles per
// chunk; the v_mfma_f64 themselves issue at 64 cycles per instruction, half the f32 rate: scripts/probes/mfma_f64_rate.hip).
template <int kTW>
__device__ __noinline__ void residentUnitsNormalEquations(
    const ldsi* srcTab, const ColumnSourceDev* __restrict__ colSources, const int32_t* __restrict__ colStart, const int32_t* __restrict__ solveList,
    const ldsd* js, const ldsd* uv, const ldsd* us, const ldsi* utin, const ldsd* ur, ldsd* jl, int ldj, int n, int U, int uc, int Kp, ldsd* g,
    ldsd* H, int tid, F64AssemblyList list) {
  const int NB = (n + 15) >> 4, T = NB * (NB + 1) / 2, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, k = lane >> 4;
  int offA[kTW], offB[kTW]; // LDS offsets of the lane's operands of tile q (doubles)
  uint32_t tileBlocks[kTW]; // 1 << I | 1 << J of tile q
  v4d acc[kTW];
#pragma unroll
  for (int q = 0; q < kTW; ++q) {
    const int t = wave + 4 * q < T ? wave + 4 * q : 0;
    int I = int((sqrtf(8.f * float(t) + 1.f) - 1.f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= t) {
      ++I;
    }
    while (I * (I + 1) / 2 > t) {
      --I;
    }
    const int Jc = t - I * (I + 1) / 2;
    offA[q] = (16 * I + i < n ? 16 * I + i : n - 1) * ldj + k;
    offB[q] = (16 * Jc + i < n ? 16 * Jc + i : n - 1) * ldj + k;
    tileBlocks[q] = (1u << I) | (1u << Jc);
    acc[q] = v4d{0.0, 0.0, 0.0, 0.0};
  }
  const int numChunks = (U + uc - 1) / uc;
  for (int u0 = 0; u0 < U; u0 += uc) {
    const int nu = U - u0 < uc ? U - u0 : uc;
    __syncthreads(); // (the previous chunk has been consumed)
    uint32_t blockMask = 0xffffffffu; // column blocks with an entry in this chunk (all, without the list)
    if (list.groups != nullptr && srcTab != nullptr) {
      const int ch = u0 / uc;
      blockMask = uint32_t(list.chunkStart[numChunks + 1 + ch]);
      residentAssembleUnitsList(srcTab, list.groups, list.extra, list.chunkStart[ch], list.chunkStart[ch + 1], js, uv, us, utin, jl, ldj, n, u0, nu, Kp, tid);
    } else {
      residentAssembleUnits(srcTab, colSources, colStart, solveList, js, uv, us, utin, jl, ldj, n, u0, nu, Kp, tid);
    }
    __syncthreads();
    const int rows = 3 * nu, rows4 = (rows + 3) & ~3, steps = rows4 >> 2;
    const ldsd* urc = ur + 3 * u0;
    // g on the two waves with the fewer tiles (wave 0 owns the 4 q-th tiles: one more whenever T is not a multiple of four)
    for (int c = tid - 128; c >= 0 && c < n; c += 128) {
      if ((blockMask >> (c >> 4) & 1u) == 0u) {
        continue; // (the column is zero in this chunk)
      }
      const ldsd* col = jl + c * ldj;
      double a = g[c];
      for (int r = 0; r < rows4; r += 4) { // four rows per trip: eight reads in flight (the pad rows of jl are zero)
        const double j0 = col[r], j1 = col[r + 1], j2 = col[r + 2], j3 = col[r + 3];
        const double r0 = urc[r], r1 = r + 1 < rows ? urc[r + 1] : 0.0, r2 = r + 2 < rows ? urc[r + 2] : 0.0, r3 = r + 3 < rows ? urc[r + 3] : 0.0;
        a += j0 * r0;
        a += j1 * r1;
        a += j2 * r2;
        a += j3 * r3;
      }
      g[c] = a;
    }
#pragma unroll
    for (int q = 0; q < kTW; q += 2) { // two tiles at a time: two independent accumulator chains
      if (wave + 4 * q >= T) {
        break;
      }
      // a tile whose row block or column block has no entry in this chunk gets nothing from it: J^T J over zero columns
      const bool need0 = (tileBlocks[q] & ~blockMask) == 0u;
      const bool need1 = q + 1 < kTW && wave + 4 * (q + 1) < T && (tileBlocks[q + 1 < kTW ? q + 1 : q] & ~blockMask) == 0u;
      if (!need0 && !need1) {
        continue;
      }
      const bool two = need1;
      const ldsd *pa0 = jl + offA[q], *pb0 = jl + offB[q];
      const ldsd *pa1 = jl + offA[q + 1 < kTW ? q + 1 : q], *pb1 = jl + offB[q + 1 < kTW ? q + 1 : q];
      v4d c0 = acc[q], c1 = acc[q + 1 < kTW ? q + 1 : q];
      for (int s0 = 0; s0 < steps; s0 += 4) {
        double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int st = s0 + e < steps ? s0 + e : steps - 1;
          a0[e] = pa0[4 * st], b0[e] = pb0[4 * st], a1[e] = pa1[4 * st], b1[e] = pb1[4 * st];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (s0 + e < steps) {
            if (need0) {
              c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], b0[e], c0, 0, 0, 0);
            }
            if (two) {
              c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], b1[e], c1, 0, 0, 0);
            }
          }
        }
      }
      acc[q] = c0;
      if (q + 1 < kTW) {
        acc[q + 1] = c1;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kTW; ++q) {
    const int t = wave + 4 * q;
    if (t < T) {
      int I = int((sqrtf(8.f * float(t) + 1.f) - 1.f) * 0.5f);
      while ((I + 1) * (I + 2) / 2 <= t) {
        ++I;
      }
      while (I * (I + 1) / 2 > t) {
        --I;
      }
      const int Jc = t - I * (I + 1) / 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * I + 4 * r + k, col = 16 * Jc + i;
        if (row < n && col <= row) {
          H[hpos(n, row, col)] += acc[q][r];
        }
      }
    }
  }
}
// a double of another lane (uniform lane index): two v_readlane_b32
__device__ __forceinline__ double readLaneD(double v, int srcLane) {
  const long long bits = __double_as_longlong(v);
  const unsigned lo = unsigned(__builtin_amdgcn_readlane(int(bits), srcLane));
  const unsigned hi = unsigned(__builtin_amdgcn_readlane(int(bits >> 32), srcLane));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// 1 / sqrt(v): the hardware's estimate and two Newton steps (each squares the error)
__device__ __forceinline__ double rsqrtNewton(double v) {
  double y = __builtin_amdgcn_rsq(v);
  y = y * (1.5 - 0.5 * v * y * y);
  y = y * (1.5 - 0.5 * v * y * y);
  return y;
}
// Right-looking blocked Cholesky of the packed lower triangle, SIXTEEN columns per panel (the shape of the single-precision
// tiledPanelFactor, mmx_kernels.hip): a lane holds one row of the panel in registers -- lanes 0-15 of every wave the
// diagonal block (redundantly, so that the pivots travel by v_readlane inside a wave), lanes 16-63 the 4 x 48 rows below
// it -- and the sixteen column steps run without a barrier; the rank-16 trailing update is four v_mfma_f64_16x16x4_f64 per
// 16 x 16 tile.  Three barriers per panel (n / 16 panels) where the four-column form took three per four columns: the
// factor of a 96-parameter system went from 173 k to ~45 k cycles.  Column-packed storage makes "a lane = a row" reads
// consecutive in LDS.  Rows covered per panel: 16 + 192 (the resident form is taken up to n = 208).  True when a pivot was
// not positive (nothing of that panel is written).
// With rhs (and the work vectors w1, w2) the forward substitution L y = rhs rides along, y in w2: the rows of a panel are
// in registers when its block of y is known, so taking the block out of the right-hand sides below costs no further read
// of the factor and no barrier of its own (the separate forward sweep was 6 x 3.6 k cycles on a 96-parameter system).
__device__ __noinline__ bool residentFactor(ldsd* H, ldsd* invd, int n, int tid, const ldsd* rhs = nullptr, ldsd* w1 = nullptr, ldsd* w2 = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  const bool diagLane = lane < 16;
  const int NB = (n + 15) >> 4;
  if (rhs != nullptr) {
    for (int c = tid; c < n; c += 256) {
      w1[c] = rhs[c];
    }
    __syncthreads();
  }
  for (int k0 = 0; k0 < n; k0 += 16) {
    const int row = diagLane ? k0 + lane : k0 + 16 + 48 * wave + (lane - 16);
    const bool active = row < n;
    double bi = (rhs != nullptr && active) ? w1[row] : 0.0;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int col = k0 + c;
      a[c] = (active && col < n && col <= row) ? H[hpos(n, row, col)] : (row == col ? 1.0 : 0.0); // (rows / columns beyond n: identity)
    }
    double invMine = 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const double djj = readLaneD(a[j], j);
      bad = bad || !(djj > 0.0);
      const double y = rsqrtNewton(djj);
      a[j] *= y; // the diagonal lane: l_jj = v / sqrt(v)
      if (lane == j) {
        invMine = y;
      }
#pragma unroll
      for (int c = j + 1; c < 16; ++c) {
        a[c] -= a[j] * readLaneD(a[j], c); // L(k0 + c, k0 + j) sits in diagonal lane c
      }
    }
    if (bad) { // (every wave has factored the same diagonal block: uniform)
      return true;
    }
    if (rhs != nullptr && wave == 0) { // L_kk y_k = s_k with the block's rows in registers (lanes 0-15 matter)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double yj = readLaneD(bi, j) * readLaneD(invMine, j);
        bi = (lane == j) ? yj : ((diagLane && lane > j) ? bi - a[j] * yj : bi); // (lanes 16-63 keep their rows' right-hand sides)
      }
      if (diagLane && active) {
        w2[row] = bi;
      }
    }
    __syncthreads(); // the panel has been read by everybody (and y_k is visible)
    if (rhs != nullptr && !diagLane && active) { // the block out of this row's right-hand side
      double v = bi;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        v -= a[c] * w2[k0 + c];
      }
      w1[row] = v;
    }
    if (diagLane) {
      if (wave == 0 && active) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if (c <= lane) {
            H[hpos(n, row, k0 + c)] = a[c];
          }
        }
        invd[row] = invMine;
      }
    } else if (active) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        H[hpos(n, row, k0 + c)] = a[c]; // (row >= k0 + 16: every column of the panel exists)
      }
    }
    __syncthreads();
    // trailing update H(i, j) -= sum_c L(i, k0 + c) L(j, k0 + c), i >= j >= k0 + 16
    const int base = k0 + 16;
    if (base < n) {
      const int li = lane & 15, lk = lane >> 4, I0 = base >> 4;
      int t = 0;
      for (int I = I0; I < NB; ++I) {
        for (int Jc = I0; Jc <= I; ++Jc, ++t) {
          if ((t & 3) != wave) {
            continue;
          }
          const int ra = 16 * I + li, rb = 16 * Jc + li;
          double av[4], bv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = k0 + 4 * q + lk;
            av[q] = ra < n ? H[hpos(n, ra, col)] : 0.0;
            bv[q] = rb < n ? H[hpos(n, rb, col)] : 0.0;
          }
          v4d c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], c, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rr = 16 * I + 4 * r + lk, cc = 16 * Jc + li;
            if (rr < n && cc <= rr) {
              H[hpos(n, rr, cc)] -= c[r];
            }
          }
        }
      }
      __syncthreads();
    }
  }
  return false;
}
__device__ __noinline__ void residentBackward(const ldsd* H, const ldsd* invd, ldsd* w2, ldsd* x, int n, int tid);
// L y = rhs, L^T x = y on the packed factor, sixteen unknowns per barrier pair: wave 0 solves the diagonal block with its
// rows (columns) in registers and the unknowns travelling by v_readlane, then every thread takes the block out of its own
// row.  w1 / w2: work vectors.  n <= 256 (one row per thread).
__device__ __noinline__ void residentSolve(const ldsd* H, const ldsd* invd, ldsd* w1, ldsd* w2, const ldsd* rhs, ldsd* x, int n, int tid) {
  const int lane = tid & 63, wave = tid >> 6, lrow = lane & 15;
  const int NB = (n + 15) >> 4;
  for (int c = tid; c < n; c += 256) {
    w1[c] = rhs[c];
  }
  for (int k = 0; k < NB; ++k) { // L y = rhs: w1 is consumed, y lands in w2
    const int k0 = 16 * k;
    __syncthreads();
    if (wave == 0) {
      const int row = k0 + lrow;
      double dg[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        dg[c] = (row < n && c < lrow) ? H[hpos(n, row, k0 + c)] : 0.0;
      }
      double bi = row < n ? w1[row] : 0.0;
      const double iv = row < n ? invd[row] : 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double yj = readLaneD(bi, j) * readLaneD(iv, j);
        bi = (lrow == j) ? yj : bi - dg[j] * yj; // (dg[j] = 0 for the rows above j)
      }
      if (lane < 16 && row < n) {
        w2[row] = bi;
      }
    }
    __syncthreads();
    const int i = tid;
    if (i >= k0 + 16 && i < n) {
      double v = w1[i];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        v -= H[hpos(n, i, k0 + c)] * w2[k0 + c];
      }
      w1[i] = v;
    }
  }
  residentBackward(H, invd, w2, x, n, tid);
}
// L^T x = y on the packed factor (y in w2, consumed), sixteen unknowns per barrier pair
__device__ __noinline__ void residentBackward(const ldsd* H, const ldsd* invd, ldsd* w2, ldsd* x, int n, int tid) {
  const int lane = tid & 63, wave = tid >> 6, lrow = lane & 15;
  const int NB = (n + 15) >> 4;
  for (int k = NB - 1; k >= 0; --k) {
    const int k0 = 16 * k;
    __syncthreads();
    if (wave == 0) {
      const int col = k0 + lrow;
      double dg[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        dg[c] = (col < n && k0 + c < n && c > lrow) ? H[hpos(n, k0 + c, col)] : 0.0; // L(k0 + c, col)
      }
      double bi = col < n ? w2[col] : 0.0;
      const double iv = col < n ? invd[col] : 0.0;
#pragma unroll
      for (int j = 15; j >= 0; --j) {
        const double xj = readLaneD(bi, j) * readLaneD(iv, j);
        bi = (lrow == j) ? xj : bi - dg[j] * xj; // (dg[j] = 0 for the columns right of j)
      }
      if (lane < 16 && col < n) {
        x[col] = bi;
      }
    }
    __syncthreads();
    const int i = tid;
    if (i < k0) {
      double v = w2[i];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if (k0 + c < n) {
          v -= H[hpos(n, k0 + c, i)] * x[k0 + c];
        }
      }
      w2[i] = v;
    }
  }
  __syncthreads();
}

// kRes: the resident instantiation (round 4).  The system lives in LDS for the whole solve: J is assembled a chunk of
// constraints at a time straight into LDS and consumed there (no dense J anywhere), H is the packed lower triangle in
// LDS, factored by a right-looking blocked Cholesky (four columns per barrier pair) and solved by blocked substitutions
// (one barrier per four unknowns), all in double.  Same arithmetic per entry as the scratch form below it (sums over the
// rows in a different grouping: parity with the oracle's double run stays at the 1e-10 the tests hold).  Taken when the
// packed H and a chunk of at least twelve rows fit a workgroup's LDS (n <= ~180 solved parameters); the scratch form
// (J and H in a global scratch of the problem) is kept for the systems beyond.
template <bool kRes>
__global__ void __launch_bounds__(256, kRes ? 2 : 1) solveF64Kernel(
    const RigDev rigArg, // (read-only: the per-element copies are taken below the element-list gate -- a modified by-value argument
    const ProblemDev pbArg, // is copied to scratch at entry, which every workgroup beyond *sel.count paid before it left)
    const int32_t* __restrict__ solveList, // [n] parameters of the dense system (enabled, structurally non-zero)
    int n,
    double* __restrict__ theta, // [B][P] in/out
    SolveStateDev st,
    FusedParams fp,
    double* __restrict__ Jg, // [B][n][M] scratch: the dense Jacobian (solved columns, column-major)
    double* __restrict__ Hg, // [B][n][n] scratch: H, then its Cholesky factor (lower triangle, column-major)
    double* __restrict__ Hg2, // [B][n][n] scratch of MMX_STEP_TRUST_REGION: J^T J without damping, kept while the damping changes (else null)
    int rc, // rows of J staged in LDS at a time
    F64AssemblyList alist, // the resident form's assembly list, or null pointers
    F64Select sel) { // MMX_PRECISION_AUTO: the elements to solve (null map: all of them)
  extern __shared__ __attribute__((aligned(16))) double dmem[];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  if (sel.map != nullptr) { // (uniform)
    if (b >= *sel.count) {
      return;
    }
    b = sel.map[b];
  }
  RigDev rig = rigArg;
  ProblemDev pb = pbArg;
  selectInstanceRig(rig, b);
  selectInstanceWeights(pb, b);
  const int J = rig.J, P = rig.P, U = pb.U, G = pb.G, M = pb.rowsJoint; // (rowsJoint = 3 U + rows of the further joint error functions + 3 NE)
  F64Lds s;
  {
    ldsd* p = (ldsd*)dmem;
    auto take = [&](size_t c) {
      ldsd* r = p;
      p += (c + 1) & ~size_t(1);
      return r;
    };
    s.th = take(P), s.trial = take(P), s.jp = take(7 * size_t(J)), s.js = take(size_t(kDs) * J);
    s.uv = take(3 * size_t(U)), s.ur = take(size_t(M)), s.us = take(U);
    s.g = take(n), s.d = take(n), s.red = take(8);
    s.utin = reinterpret_cast<ldsi*>(take((U + 1) / 2 + 1));
    s.flags = reinterpret_cast<ldsi*>(take(2));
    s.colOf = reinterpret_cast<ldsi*>(take((size_t(P) + 1) / 2));
    s.jl = take(size_t(n) * size_t(rc + 1));
    s.gev = take(size_t(kGevD) * size_t(G + pb.NE));
    s.H = s.invd = s.w1 = s.w2 = nullptr;
    s.srcTab = nullptr;
    if (kRes) {
      s.H = take(size_t(n) * size_t(n + 1) / 2);
      s.invd = take(n), s.w1 = take(n), s.w2 = take(n);
    }
  }
  if (kRes && J < 4096 && size_t(7) * size_t(J) <= size_t(n) * size_t(rc + 1)) {
    // The joint parameters (FK's scratch) move into the chunk buffer, which is dead whenever FK runs; their own 7 J
    // doubles hold the solved columns' sources for the whole solve, packed to three words each -- when they fit
    // (n + 1 + 3 nsrc <= 14 J words; else the assembly reads them from L2 as before).  residentAssembleUnits reads them
    // once per entry of J: from L2 that was three dependent round trips per entry.
    ldsi* tab = reinterpret_cast<ldsi*>(s.jp);
    for (int c = tid; c < n; c += 256) {
      const int p = solveList[c];
      tab[c + 1] = pb.colStart[p + 1] - pb.colStart[p];
    }
    if (tid == 0) {
      tab[0] = 0;
    }
    __syncthreads();
    if (tid == 0) {
      for (int c = 0; c < n; ++c) {
        tab[c + 1] += tab[c];
      }
    }
    __syncthreads();
    const int nsrc = tab[n];
    if (n + 1 + 3 * nsrc <= 14 * J) { // (uniform)
      for (int c = tid; c < n; c += 256) {
        const int p = solveList[c];
        const int k0 = pb.colStart[p], cnt = tab[c + 1] - tab[c];
        ldsi* o = tab + n + 1 + 3 * tab[c];
        for (int e = 0; e < cnt; ++e) {
          const ColumnSourceDev cs = pb.colSources[k0 + e];
          o[3 * e] = cs.joint | cs.dof << 12 | (cs.parent + 1) << 15;
          o[3 * e + 1] = cs.tin | cs.tout << 16;
          o[3 * e + 2] = __float_as_int(cs.weight);
        }
      }
      s.srcTab = tab;
      s.jp = s.jl;
    }
    __syncthreads();
  }
  // H(i, j), i >= j: the packed LDS triangle (kRes) or the column-major global scratch
  auto addToH = [&](int i, int j, double v) {
    if (kRes) {
      s.H[hpos(n, i, j)] += v;
    } else {
      Hg[size_t(b) * size_t(n) * size_t(n) + size_t(j) * n + i] += v;
    }
  };
  double* thg = theta + size_t(b) * P;
  double* Jb = Jg + size_t(b) * size_t(n) * size_t(M);
  double* Hb = Hg + size_t(b) * size_t(n) * size_t(n);
  double* H0 = Hg2 != nullptr ? Hg2 + size_t(b) * size_t(n) * size_t(n) : nullptr;
  const bool trust = fp.stepRule == MMX_STEP_TRUST_REGION; // TrustRegionQRT<double>::doIteration (trust_region_qr.cpp:52-270)
  double trRadius = double(fp.trustRadius); // initializeSolver (:38-41); lives across the iterations
  for (int i = tid; i < P; i += 256) {
    s.th[i] = sel.thetaInit != nullptr ? double(sel.thetaInit[size_t(b) * P + i]) : thg[i];
  }
  if (tid == 0) {
    s.flags[0] = 0, s.flags[1] = 0, s.flags[2] = 0;
  }
  for (int i = tid; i < P; i += 256) {
    s.colOf[i] = -1;
  }
  __syncthreads();
  for (int c = tid; c < n; c += 256) {
    s.colOf[solveList[c]] = c;
  }
  const bool hasParamRows = pb.M > pb.rowsJoint; // limit / model-parameter rows present (uniform)
  __syncthreads();
#ifdef MMX_EXP_F64CLK
  long long clkAcc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // (8, 9: assembly / accumulation of the J chunks, parts of 2)
  long long clkT = clock64();
#define F64CLK(slot) { __syncthreads(); const long long now_ = clock64(); clkAcc[slot] += now_ - clkT; clkT = now_; }
#else
#define F64CLK(slot)
#endif
  double lastError = DBL_MAX, curError = DBL_MAX; // solver.cpp:84-85
  double lambda = double(fp.lambda);
  int itersDone = 0;
  for (int it = 0; it < fp.maxIterations; ++it) {
    // ---- SkeletonSolverFunctionT::getJacobian: state, residual, Jacobian (skeleton_solver_function.cpp:200-261)
    F64CLK(7)
    fkF64(rig, s, s.th, tid, true);
    F64CLK(0)
    double e = 0.0;
    for (int u = tid; u < U; u += 256) {
      e += evalUnitF64(pb, s, b, u, true);
    }
    if (hasParamRows) {
      e += paramRowsErrorF64<true>(rig, pb, s.th, b, tid);
    }
    for (int g = tid; g < G; g += 256) { // the further joint error functions: residual rows, evaluation record for the Jacobian
      const JointBlockDev k = jointBlockOf(pb, b, pb.genBlock[g]);
      const int i = g - k.first;
      const JointEvalD o = evalJointConstraintF64(k, s.js, pb.genJoint[g], size_t(b) * size_t(k.count) + size_t(i));
      const int row = k.rowStart + o.nrows * i;
      e += o.werr;
      for (int q = 0; q < o.nrows; ++q) {
        s.ur[row + q] = o.sigma * o.f[q];
      }
      const double sg = fabs(o.sigma) <= 1e-9 ? 0.0 : o.sigma; // early termination (joint_error_function-inl.h:216): the rows stay zero
      ldsd* w = s.gev + kGevD * g;
      w[0] = o.vp.x, w[1] = o.vp.y, w[2] = o.vp.z, w[3] = o.vn.x, w[4] = o.vn.y, w[5] = o.vn.z;
      for (int q = 0; q < 9; ++q) {
        w[6 + q] = sg * o.dp[q];
        w[15 + q] = sg * o.dn[q];
      }
      ldsi* wi = reinterpret_cast<ldsi*>(w + 24);
      wi[0] = pb.genTin[g], wi[1] = row, wi[2] = o.nrows | (o.hasPoint ? 16 : 0) | (o.hasDir ? 32 : 0), wi[3] = -1;
    }
    for (int q = tid; q < pb.NE; q += 256) { // ellipsoid limits: a point constraint whose walk stops at ellipsoidParent
      const EllipsoidDev ct = pb.ellipsoids[q];
      const EllipsoidEvalD o = evalEllipsoidF64(ct, s.js, pb.wLimit);
      const int row = pb.rowsJoint - 3 * pb.NE + 3 * q;
      e += o.werr;
      s.ur[row] = o.diff.x * o.jwgt, s.ur[row + 1] = o.diff.y * o.jwgt, s.ur[row + 2] = o.diff.z * o.jwgt;
      ldsd* w = s.gev + kGevD * (G + q);
      w[0] = o.position.x, w[1] = o.position.y, w[2] = o.position.z, w[3] = w[4] = w[5] = 0.0;
      for (int k = 0; k < 9; ++k) {
        w[6 + k] = (k == 0 || k == 4 || k == 8) ? o.jwgt : 0.0;
        w[15 + k] = 0.0;
      }
      ldsi* wi = reinterpret_cast<ldsi*>(w + 24);
      wi[0] = ct.tinParent, wi[1] = row, wi[2] = 3 | 16, wi[3] = ct.tinStop;
    }
    curError = blockSumF64(s, e, tid); // (not rounded: the value getJacobian returns)
    F64CLK(1)
    if (trust) {
      lambda = 0.0; // H is assembled without damping; the trust region adds its own per factorisation
    }
    if (kRes) {
      // ---- the resident form: H = lambda I, g = 0; then per chunk of constraints their rows of J straight into LDS
      // (jl, column-major, ldj doubles per column) and from there into g and H
      const int ldj = rc + 1;
      for (int idx = tid; idx < n * (n + 1) / 2; idx += 256) {
        s.H[idx] = 0.0;
      }
      for (int c = tid; c < n; c += 256) {
        s.g[c] = 0.0;
      }
      __syncthreads();
      for (int c = tid; c < n; c += 256) {
        s.H[hpos(n, c, c)] = lambda;
      }
      // rows [r0, r0 + rows) of J are in jl: g += J^T r, H += J^T J (a thread owns 4 x 4 blocks of the lower triangle)
      auto accumulate = [&](int r0, int rows) { residentAccumulate(s.jl, ldj, s.ur + r0, rows, s.g, s.H, n, tid); };
      // position / orientation units, uc at a time
      const int uc = rc / 3;
      {
        const int NBt = (n + 15) >> 4;
        if (NBt * (NBt + 1) / 2 <= 24) { // (n <= 96: the tiles of H stay in registers across the chunks)
          residentUnitsNormalEquations<6>(s.srcTab, pb.colSources, pb.colStart, solveList, s.js, s.uv, s.us, s.utin, s.ur, s.jl, ldj, n, U, uc, pb.Kp, s.g, s.H, tid, alist);
          F64CLK(9)
        } else {
          for (int u0 = 0; u0 < U; u0 += uc) {
            const int nu = U - u0 < uc ? U - u0 : uc;
            __syncthreads(); // (the previous chunk has been consumed)
            if (alist.groups != nullptr && s.srcTab != nullptr) {
              const int ch = u0 / uc;
              residentAssembleUnitsList(s.srcTab, alist.groups, alist.extra, alist.chunkStart[ch], alist.chunkStart[ch + 1], s.js, s.uv, s.us, s.utin, s.jl, ldj, n, u0, nu, pb.Kp, tid);
            } else {
              residentAssembleUnits(s.srcTab, pb.colSources, pb.colStart, solveList, s.js, s.uv, s.us, s.utin, s.jl, ldj, n, u0, nu, pb.Kp, tid);
            }
            __syncthreads();
            F64CLK(8)
            accumulate(3 * u0, 3 * nu);
            F64CLK(9)
          }
        }
      }
      // the further joint error functions / ellipsoid limits: consecutive constraints while their rows fit a chunk
      const int GT = G + pb.NE;
      for (int g0 = 0; g0 < GT;) {
        const int rowFirst = reinterpret_cast<const ldsi*>(s.gev + kGevD * g0 + 24)[1];
        int g1 = g0, rowEnd = rowFirst;
        while (g1 < GT) {
          const ldsi* wi = reinterpret_cast<const ldsi*>(s.gev + kGevD * g1 + 24);
          if (wi[1] + (wi[2] & 15) - rowFirst > rc) {
            break;
          }
          rowEnd = wi[1] + (wi[2] & 15);
          ++g1;
        }
        const int ng = g1 - g0;
        __syncthreads();
        for (int item = tid; item < n * ng; item += 256) {
          const int c = item / ng, g = g0 + (item - c * ng);
          const int p = solveList[c];
          const ldsd* w = s.gev + kGevD * g;
          const ldsi* wi = reinterpret_cast<const ldsi*>(w + 24);
          const int tin = wi[0], row = wi[1], nrows = wi[2] & 15, tinStop = wi[3];
          const bool hasPoint = (wi[2] & 16) != 0, hasDir = (wi[2] & 32) != 0;
          const D3 vp{w[0], w[1], w[2]}, vn{w[3], w[4], w[5]};
          double acc[3] = {0.0, 0.0, 0.0};
          const int e1 = pb.colStart[p + 1];
          for (int k = pb.colStart[p]; k < e1; ++k) {
            const ColumnSourceDev cs = pb.colSources[k];
            if (!(cs.tin <= tin && tin < cs.tout)) {
              continue; // the source's joint is not an ancestor of the constraint's joint
            }
            if (tinStop >= 0 && cs.tin <= tinStop && tinStop < cs.tout) {
              continue; // ellipsoid limit: the walk stopped before this joint
            }
            bool ap;
            D3 gp{0.0, 0.0, 0.0}, gn{0.0, 0.0, 0.0};
            if (hasPoint) {
              gp = sourceDerivativeF64(cs, s.js, vp, tin, true, ap);
              if (!ap) {
                gp = D3{0.0, 0.0, 0.0};
              }
            }
            if (hasDir) {
              gn = sourceDerivativeF64(cs, s.js, vn, tin, false, ap);
              if (!ap) {
                gn = D3{0.0, 0.0, 0.0};
              }
            }
            const double wt = double(cs.weight);
            for (int q = 0; q < 3; ++q) {
              const double jc = (w[6 + 3 * q] * gp.x + w[7 + 3 * q] * gp.y + w[8 + 3 * q] * gp.z) +
                  (w[15 + 3 * q] * gn.x + w[16 + 3 * q] * gn.y + w[17 + 3 * q] * gn.z);
              acc[q] += jc * wt;
            }
          }
          ldsd* o = s.jl + c * ldj + (row - rowFirst);
          for (int q = 0; q < nrows; ++q) {
            o[q] = acc[q];
          }
        }
        for (int idx = tid; idx < n * ((4 - (rowEnd - rowFirst) % 4) % 4); idx += 256) {
          const int pad = (4 - (rowEnd - rowFirst) % 4) % 4, c = idx / pad;
          s.jl[c * ldj + (rowEnd - rowFirst) + (idx - c * pad)] = 0.0;
        }
        __syncthreads();
        accumulate(rowFirst, rowEnd - rowFirst);
        g0 = g1;
      }
      __syncthreads();
    } else {
    for (int item = tid; item < n * U; item += 256) {
      const int c = item / U, u = item - c * U;
      const int p = solveList[c];
      const D3 v{s.uv[3 * u], s.uv[3 * u + 1], s.uv[3 * u + 2]};
      D3 acc{0.0, 0.0, 0.0};
      const int e1 = pb.colStart[p + 1];
      for (int k = pb.colStart[p]; k < e1; ++k) {
        const ColumnSourceDev cs = pb.colSources[k];
        bool applies;
        const D3 gq = sourceDerivativeF64(cs, s.js, v, s.utin[u], u < pb.Kp, applies);
        if (applies) { // jac.col(p) += derivScale * dfdv * jc * value (joint_error_function-inl.h:254-289)
          const double w = double(cs.weight);
          acc.x += (s.us[u] * gq.x) * w, acc.y += (s.us[u] * gq.y) * w, acc.z += (s.us[u] * gq.z) * w;
        }
      }
      double* o = Jb + size_t(c) * M + 3 * u;
      o[0] = acc.x, o[1] = acc.y, o[2] = acc.z;
    }
    const int GT = G + pb.NE;
    for (int item = tid; item < n * GT; item += 256) { // rows of the further joint error functions / ellipsoid limits (jointBlocksKernel in double)
      const int c = item / GT, g = item - c * GT;
      const int p = solveList[c];
      const ldsd* w = s.gev + kGevD * g;
      const ldsi* wi = reinterpret_cast<const ldsi*>(w + 24);
      const int tin = wi[0], row = wi[1], nrows = wi[2] & 15, tinStop = wi[3];
      const bool hasPoint = (wi[2] & 16) != 0, hasDir = (wi[2] & 32) != 0;
      const D3 vp{w[0], w[1], w[2]}, vn{w[3], w[4], w[5]};
      double acc[3] = {0.0, 0.0, 0.0};
      const int e1 = pb.colStart[p + 1];
      for (int k = pb.colStart[p]; k < e1; ++k) {
        const ColumnSourceDev cs = pb.colSources[k];
        if (!(cs.tin <= tin && tin < cs.tout)) {
          continue; // the source's joint is not an ancestor of the constraint's joint
        }
        if (tinStop >= 0 && cs.tin <= tinStop && tinStop < cs.tout) {
          continue; // ellipsoid limit: the walk stopped before this joint
        }
        bool ap;
        D3 gp{0.0, 0.0, 0.0}, gn{0.0, 0.0, 0.0};
        if (hasPoint) {
          gp = sourceDerivativeF64(cs, s.js, vp, tin, true, ap);
          if (!ap) {
            gp = D3{0.0, 0.0, 0.0};
          }
        }
        if (hasDir) {
          gn = sourceDerivativeF64(cs, s.js, vn, tin, false, ap);
          if (!ap) {
            gn = D3{0.0, 0.0, 0.0};
          }
        }
        const double wt = double(cs.weight);
        for (int q = 0; q < 3; ++q) {
          const double jc = (w[6 + 3 * q] * gp.x + w[7 + 3 * q] * gp.y + w[8 + 3 * q] * gp.z) +
              (w[15 + 3 * q] * gn.x + w[16 + 3 * q] * gn.y + w[17 + 3 * q] * gn.z);
          acc[q] += jc * wt;
        }
      }
      double* o = Jb + size_t(c) * M + row;
      for (int q = 0; q < nrows; ++q) {
        o[q] = acc[q];
      }
    }
    __threadfence_block();
    __syncthreads();
    // ---- H.triangularView<Lower>() += J^T J ; Jtr += J^T r ; diagonal += regularization (gauss_newton_solver.cpp:215-216,248)
    // J is staged through LDS in chunks of `rc` rows (read from the scratch once per iteration instead of once
    // per entry of H: 4096 x 10 x 14 MB of re-reads made the first version HBM-bound); a thread owns 4 x 4
    // blocks of the lower triangle and adds a chunk's contribution to the scratch block by block.
    {
      const int nb4 = (n + 3) >> 2, nblk = nb4 * (nb4 + 1) / 2, ldj = rc + 1;
      for (int c = tid; c < n; c += 256) {
        s.g[c] = 0.0;
      }
      for (int r0 = 0; r0 < M; r0 += rc) {
        const int rows = M - r0 < rc ? M - r0 : rc;
        __syncthreads();
        for (int idx = tid; idx < n * rows; idx += 256) {
          const int c = idx / rows, r = idx - c * rows;
          s.jl[c * ldj + r] = Jb[size_t(c) * M + r0 + r];
        }
        __syncthreads();
        for (int c = tid; c < n; c += 256) {
          double acc = s.g[c];
          for (int r = 0; r < rows; ++r) {
            acc += s.jl[c * ldj + r] * s.ur[r0 + r];
          }
          s.g[c] = acc;
        }
        for (int q = tid; q < nblk; q += 256) {
          int bi = int((sqrt(8.0 * double(q) + 1.0) - 1.0) * 0.5); // q = bi (bi + 1) / 2 + bj, bj <= bi
          while ((bi + 1) * (bi + 2) / 2 <= q) {
            ++bi;
          }
          while (bi * (bi + 1) / 2 > q) {
            --bi;
          }
          const int bj = q - bi * (bi + 1) / 2;
          double acc[4][4] = {};
          const ldsd* ci[4];
          const ldsd* cj[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { // (columns beyond n read column n - 1: their entries are never stored)
            ci[k] = s.jl + (4 * bi + k < n ? 4 * bi + k : n - 1) * ldj;
            cj[k] = s.jl + (4 * bj + k < n ? 4 * bj + k : n - 1) * ldj;
          }
          for (int r = 0; r < rows; ++r) {
            double av[4], bv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              av[k] = ci[k][r], bv[k] = cj[k][r];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
#pragma unroll
              for (int y = 0; y < 4; ++y) {
                acc[x][y] += av[x] * bv[y];
              }
            }
          }
#pragma unroll
          for (int x = 0; x < 4; ++x) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
              const int i = 4 * bi + x, j = 4 * bj + y;
              if (i < n && j <= i) {
                double* h = Hb + size_t(j) * n + i;
                *h = (r0 == 0 ? (i == j ? lambda : 0.0) : *h) + acc[x][y];
              }
            }
          }
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    }
    F64CLK(2)
    if (!kRes && M == 0) { // no joint-constraint rows at all (parameter-space rows only): H starts as lambda I
      for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx % n, j = idx / n;
        if (j <= i) {
          Hb[size_t(j) * n + i] = i == j ? lambda : 0.0;
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    if (hasParamRows) {
      // the rows of the parameter-space blocks have at most kLimitEntries non-zeros: what J^T J and J^T r would take
      // from them goes straight into H and g.  Limits one after the other by one thread (several may share an entry of
      // H: a fixed order keeps the solve deterministic), the model prior's diagonal rows over the threads.
      if (tid == 0 && pb.NL > 0 && pb.wLimit > 0.f) {
        const double tWeight = double(1e+1f * pb.wLimit);
        for (int l = 0; l < pb.NL; ++l) {
          const LimitRowT<double> row = evalLimit<double>(rig, pb.limits[l], (const double*)s.th, pb.enabledMask, tWeight);
          for (int x = 0; x < kLimitEntries; ++x) {
            const int cx = row.idx[x] >= 0 ? s.colOf[row.idx[x]] : -1;
            if (cx < 0) {
              continue;
            }
            s.g[cx] += row.coef[x] * row.r;
            for (int y = 0; y <= x; ++y) {
              const int cy = row.idx[y] >= 0 ? s.colOf[row.idx[y]] : -1;
              if (cy < 0) {
                continue;
              }
              const int hi = cx > cy ? cx : cy, lo = cx > cy ? cy : cx;
              addToH(hi, lo, row.coef[x] * row.coef[y]); // (a row's parameters are distinct: limitScatterRow merges)
            }
          }
        }
      }
      __threadfence_block();
      __syncthreads();
      if (pb.hasModel && pb.wModel > 0.f) {
        const float* tp = pb.mpTarget + size_t(b) * P;
        const float* tw = pb.mpWeights + size_t(b) * P;
        const double sW = double(sqrtf(pb.wModel * 1e-1f)); // sWeight: a float in both instantiations (:109)
        for (int c = tid; c < n; c += 256) {
          const int p = solveList[c];
          const double w = double(tw[p]);
          if (pb.enabledMask[p] != 0 && w > 0.0) {
            const double jw = sW * w, r = (w * (s.th[p] - double(tp[p]))) * sW;
            addToH(c, c, jw * jw);
            s.g[c] += jw * r;
          }
        }
      }
      __threadfence_block();
      __syncthreads();
    }
    // ---- llt_.compute(H) (Eigen::LLT, lower): right-looking, one column per step; a non-positive pivot is
    // recorded (the reference never checks LLT::info(), gauss_newton_solver.cpp:251)
    bool notPd = false;
    // the resident forms (kRes).  Rows are dealt one per thread (n <= 256).
    auto factorRes = [&]() { notPd = residentFactor(s.H, s.invd, n, tid); };
    // x = (L L^T)^-1 rhs by blocked substitutions: every thread solves the four unknowns of a block for itself, the
    // thread of a later row takes them out of its right-hand side; one barrier per block (x and rhs may be the same array)
    auto solveRes = [&](const ldsd* rhs, ldsd* x) { residentSolve(s.H, s.invd, s.w1, s.w2, rhs, x, n, tid); };
    auto factorScratch = [&]() {
    notPd = false;
    for (int k = 0; k < n; ++k) {
      const double dkk = Hb[size_t(k) * n + k];
      if (!(dkk > 0.0)) {
        notPd = true; // every thread reads the same value
        break;
      }
      const double lkk = sqrt(dkk);
      __syncthreads();
      for (int i = k + tid; i < n; i += 256) {
        Hb[size_t(k) * n + i] = i == k ? lkk : Hb[size_t(k) * n + i] / lkk;
      }
      __threadfence_block();
      __syncthreads();
      const int rem = n - k - 1;
      for (int item = tid; item < rem * (rem + 1) / 2; item += 256) {
        int ii = int((sqrt(8.0 * double(item) + 1.0) - 1.0) * 0.5);
        while ((ii + 1) * (ii + 2) / 2 <= item) {
          ++ii;
        }
        while (ii * (ii + 1) / 2 > item) {
          --ii;
        }
        const int jj = item - ii * (ii + 1) / 2;
        const int i = k + 1 + ii, j = k + 1 + jj;
        Hb[size_t(j) * n + i] -= Hb[size_t(k) * n + i] * Hb[size_t(k) * n + j];
      }
      __threadfence_block();
      __syncthreads();
    }
    };
    // ---- x = llt_.solve(rhs): wave 0, lanes over the already known entries (x and rhs may be the same array)
    auto solveScratch = [&](const ldsd* rhs, ldsd* x) {
      for (int c = tid; c < n; c += 256) {
        x[c] = rhs[c];
      }
      __syncthreads();
      if (tid < 64) {
        for (int k = 0; k < n; ++k) { // L y = g
          double part = 0.0;
          for (int j = tid; j < k; j += 64) {
            part += Hb[size_t(j) * n + k] * x[j];
          }
          for (int off = 32; off > 0; off >>= 1) {
            part += __shfl_xor(part, off, 64);
          }
          if (tid == 0) {
            x[k] = (x[k] - part) / Hb[size_t(k) * n + k];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        for (int k = n - 1; k >= 0; --k) { // L^T x = y
          double part = 0.0;
          for (int j = k + 1 + tid; j < n; j += 64) {
            part += Hb[size_t(k) * n + j] * x[j];
          }
          for (int off = 32; off > 0; off >>= 1) {
            part += __shfl_xor(part, off, 64);
          }
          if (tid == 0) {
            x[k] = (x[k] - part) / Hb[size_t(k) * n + k];
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      __syncthreads();
    };
    auto factorH = [&]() {
      if (kRes) {
        factorRes();
      } else {
        factorScratch();
      }
    };
    auto solveLLt = [&](const ldsd* rhs, ldsd* x) {
      if (kRes) {
        solveRes(rhs, x);
      } else {
        solveScratch(rhs, x);
      }
    };
    F64CLK(3)
    if (!trust) {
      if (kRes) { // factor with the forward substitution riding along, then the backward sweep
        notPd = residentFactor(s.H, s.invd, n, tid, s.g, s.w1, s.w2);
        F64CLK(4)
        if (!notPd) {
          residentBackward(s.H, s.invd, s.w2, s.d, n, tid);
        }
      } else {
        factorH();
        F64CLK(4)
        if (!notPd) {
          solveLLt(s.g, s.d);
        }
      }
      F64CLK(5)
    }
    // ---- updateParameters (gauss_newton_solver.cpp:283-313; subset_gauss_newton_solver.cpp:117-142) / LM schedule
    auto makeTrial = [&](double scale) {
      for (int i = tid; i < P; i += 256) {
        s.trial[i] = s.th[i];
      }
      __syncthreads();
      for (int c = tid; c < n; c += 256) {
        s.trial[solveList[c]] -= scale * s.d[c];
      }
      __syncthreads();
    };
    auto acceptTrial = [&]() {
      for (int i = tid; i < P; i += 256) {
        s.th[i] = s.trial[i];
      }
      __syncthreads();
    };
    if (trust) {
      // ---- TrustRegionQRT<double>::doIteration (trust_region_qr.cpp:52-270) on the normal equations: the reference
      // seeds R with lambda = 1e-10 ON its diagonal (:86-87, online_householder_qr.cpp:133-140) and appends
      // sqrt(lambda_new - lambda) I rows when the damping grows (:215-224), so R^T R = J^T J + (1e-20 + lambda - 1e-10) I;
      // here H0 = J^T J (+ the parameter-space rows) is kept and every value of the damping is one LL^T of H0 + mu I.
      if (kRes) {
        for (int idx = tid; idx < n * (n + 1) / 2; idx += 256) {
          H0[idx] = s.H[idx];
        }
      } else {
        for (int idx = tid; idx < n * n; idx += 256) { // (the lower triangle is what was assembled)
          H0[idx] = Hb[idx];
        }
      }
      __threadfence_block();
      __syncthreads();
      double lam = 1e-10;
      bool haveStep = false; // s.d solves the system of the current damping
      for (int trustStep = 0; trustStep < 10; ++trustStep) { // :157
        double mu = 0.0, dn2 = 0.0, dg = 0.0;
        bool noStep = false;
        int pdRetries = 0;
        for (int newton = 0;;) { // one pass per value of the damping (:180-231)
          mu = 1e-20 + (lam - 1e-10);
          if (!haveStep) {
            if (kRes) {
              for (int idx = tid; idx < n * (n + 1) / 2; idx += 256) {
                s.H[idx] = H0[idx];
              }
              __syncthreads();
              for (int c = tid; c < n; c += 256) {
                s.H[hpos(n, c, c)] += mu;
              }
            } else {
              for (int idx = tid; idx < n * n; idx += 256) {
                const int i = idx % n, j = idx / n;
                if (j <= i) {
                  Hb[idx] = H0[idx] + (i == j ? mu : 0.0);
                }
              }
            }
            __threadfence_block();
            __syncthreads();
            factorH();
            if (notPd) { // (cannot happen in the reference's QR) more damping, a bounded number of times
              lam = fmax(4.0 * lam, 1e-6);
              if (++pdRetries > 16) {
                noStep = true;
                break;
              }
              continue;
            }
            solveLLt(s.g, s.d);
            haveStep = true;
          }
          double p0 = 0.0, p1 = 0.0;
          for (int c = tid; c < n; c += 256) {
            p0 += s.d[c] * s.d[c];
            p1 += s.d[c] * s.g[c];
          }
          dn2 = blockSumF64(s, p0, tid);
          dg = blockSumF64(s, p1, tid);
          if (newton == 0 && 2.0 * dg < double(FLT_EPSILON) * (1.0 + curError)) { // :164 (gradientSub_ = 2 J^T r): not worth a step
            noStep = true;
            break;
          }
          if (newton < 3 && sqrt(dn2) >= 1.05 * trRadius) { // :180-181
            // Newton step on lambda (Nocedal & Wright eq. 4.44, :191-204): p_l = -(step), |q_l|^2 = p_l^T (R^T R)^-1 p_l
            solveLLt(s.d, s.trial);
            double pq = 0.0;
            for (int c = tid; c < n; c += 256) {
              pq += s.d[c] * s.trial[c];
            }
            const double q2 = blockSumF64(s, pq, tid);
            if (q2 >= double(FLT_EPSILON)) { // :198
              const double pn = sqrt(dn2);
              const double deltaLambda = (dn2 / q2) * ((pn - trRadius) / trRadius);
              if (deltaLambda > 0.0) { // :207: lambda only ever grows
                lam += deltaLambda;
                ++newton;
                haveStep = false;
                continue;
              }
            }
          }
          break;
        }
        if (noStep) {
          break;
        }
        // trial step, gain ratio against the quadratic model e - 2 g.p + p^T (J^T J + 1e-20 I) p, where
        // p^T J^T J p = g.p - mu |p|^2 because (J^T J + mu I) p = g  (:240-247)
        makeTrial(1.0);
        const double eNew = errorF64(rig, pb, s, s.trial, b, tid);
        const double predicted = dg + (mu - 1e-20) * dn2; // e - model
        const double rho = (curError - eNew) / predicted;
        if (rho < 0.25) { // :256-262
          trRadius = 0.25 * trRadius;
        } else if (rho > 0.75 && lam > 0.0) {
          trRadius = fmin(2.0 * trRadius, 10.0);
        }
        if (rho > 0.0) { // :265
          acceptTrial();
          break;
        }
      }
      notPd = false; // (the rule's own retries dealt with it)
    } else if (!notPd && fp.stepRule == 1) {
      double part = 0.0;
      for (int c = tid; c < n; c += 256) {
        part += s.d[c] * s.g[c] + lambda * s.d[c] * s.d[c];
      }
      const double predicted = blockSumF64(s, part, tid);
      makeTrial(1.0);
      const double eNew = errorF64(rig, pb, s, s.trial, b, tid);
      const double rho = predicted > 0.0 ? (curError - eNew) / predicted : -1.0;
      if (st.stepHistory != nullptr && tid == 0) {
        double* sh = st.stepHistory + (size_t(b) * fp.maxIterations + it) * 2;
        sh[0] = lambda;
        sh[1] = rho;
      }
      if (rho > 0.0) {
        acceptTrial();
      }
      if (!(rho >= 0.25)) {
        lambda = fmin(lambda * double(fp.lmUp), double(fp.lmLambdaMax));
      } else if (rho > 0.75) {
        lambda = fmax(lambda * double(fp.lmDown), double(fp.lmLambdaMin));
      }
    } else if (notPd && fp.stepRule == 1) {
      if (st.stepHistory != nullptr && tid == 0) {
        double* sh = st.stepHistory + (size_t(b) * fp.maxIterations + it) * 2;
        sh[0] = lambda;
        sh[1] = -1.0;
      }
      lambda = fmin(lambda * double(fp.lmUp), double(fp.lmLambdaMax));
    } else if (!notPd && fp.doLineSearch == 2) {
      double part = 0.0;
      for (int c = tid; c < n; c += 256) {
        part += s.g[c] * s.d[c];
      }
      const double gd = blockSumF64(s, part, tid);
      float alpha = 1.0f; // the reference keeps c_1 / tau / alpha in float (:118-142)
      for (int ls = 0; ls < 10; ++ls) {
        makeTrial(double(alpha));
        const double eNew = errorF64(rig, pb, s, s.trial, b, tid);
        if ((curError - eNew) >= double(1e-4f * alpha) * gd) {
          break;
        }
        alpha *= 0.5f;
      }
      acceptTrial();
    } else if (!notPd && fp.doLineSearch == 1) {
      const double scaledError = 1e-3 * curError;
      double scale = 1.0;
      for (int ls = 0; ls < 10; ++ls) {
        makeTrial(scale);
        const double eNew = errorF64(rig, pb, s, s.trial, b, tid);
        if ((curError - eNew) >= scale * scaledError) {
          break;
        }
        scale *= 0.5;
      }
      acceptTrial();
    } else if (!notPd) {
      for (int c = tid; c < n; c += 256) {
        s.th[solveList[c]] -= s.d[c]; // skeleton_solver_function.cpp:158
      }
    }
    if (tid == 0) { // solver.cpp:92-119
      if (st.errorHistory != nullptr) {
        st.errorHistory[size_t(b) * fp.maxIterations + it] = curError;
      }
      itersDone = it + 1;
      if (notPd) {
        s.flags[2] = 2;
      }
      const bool converged = fabs(lastError - curError) / (fabs(curError) + double(FLT_MIN)) <= double(fp.threshold) * double(FLT_EPSILON);
      s.flags[0] = (it >= fp.minIterations && converged) ? 1 : 0;
      lastError = curError;
    }
    __syncthreads();
    if (s.flags[0] != 0) {
      break;
    }
  }
#ifdef MMX_EXP_F64CLK
  F64CLK(6)
  if (b == 0 && tid == 0 && st.errorHistory != nullptr && fp.maxIterations >= 10) {
    for (int i = 0; i < 10; ++i) {
      st.errorHistory[i] = double(clkAcc[i]);
    }
  }
#endif
  // NaN / Inf: revert to the initial parameters (tensor_ik.cpp:168-173) = do not write
  int bad = 0;
  for (int i = tid; i < P; i += 256) {
    if (!isfinite(s.th[i])) {
      bad = 1;
    }
  }
  if (tid == 0) {
    s.flags[3] = itersDone; // (thread 0 keeps the count)
  }
  bad = __syncthreads_or(bad);
  if (sel.thetaInit != nullptr) { // float parameters in and out (mmx_solve's MMX_PRECISION_F64 / AUTO); a non-finite answer: the initial ones
    for (int i = tid; i < P; i += 256) {
      sel.thetaOut[size_t(b) * P + i] = bad ? sel.thetaInit[size_t(b) * P + i] : float(s.th[i]);
    }
    if (sel.map != nullptr) { // the rows of the histories past this run's last iteration still hold the single-precision run's
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
  } else if (!bad) {
    for (int i = tid; i < P; i += 256) {
      thg[i] = s.th[i];
    }
  }
  if (tid == 0) {
    st.iterations[b] = itersDone;
    st.finalError[b] = curError;
    // (escalated: the double run's status + why the element was taken + MMX_SOLVE_ESCALATED_F64)
    st.status[b] = (bad ? 1 : s.flags[2]) | (sel.map != nullptr ? ((st.status[b] & 8) | 16) : 0);
  }
}

// MMX_PRECISION_AUTO: the elements the last pass marked (status & mask, and all of `require` set), compacted into map[0 .. *count - 1].
// One thread per element, one atomic per wave (*count zeroed by the launcher): the ORDER of the list depends on the waves' arrival
// and is not reproducible -- it only decides which workgroup of the next pass solves which element; every element's result is
// its own.  (Round 5's single-workgroup scan kept index order and took 0.4 ms per 65 536 elements.)
__global__ void __launch_bounds__(256) selectSuspectKernel(const int32_t* __restrict__ status, int B, int32_t mask, int32_t require, int32_t* __restrict__ map, int32_t* __restrict__ count) {
  const int b = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  const bool take = b < B && (status[b] & mask) != 0 && (status[b] & require) == require;
  const unsigned long long m = __ballot(take);
  if (m == 0ull) {
    return;
  }
  int base = 0;
  if (lane == 0) {
    base = atomicAdd(count, __popcll(m));
  }
  base = __shfl(base, 0, 64);
  if (take) {
    map[base + __popcll(m & ((1ull << lane) - 1ull))] = b;
  }
}

} // namespace

static size_t solveF64BaseDoubles(int J, int P, int U, int n, int G, int genRows) {
  auto e = [](size_t c) { return (c + 1) & ~size_t(1); };
  return 2 * e(P) + e(7 * size_t(J)) + e(size_t(kDs) * J) + e(3 * size_t(U)) + e(3 * size_t(U) + size_t(genRows)) + e(U) + 2 * e(n) + e(8) +
      e((U + 1) / 2 + 1) + e(2) + e((size_t(P) + 1) / 2) + e(size_t(kGevD) * size_t(G));
}
// rows of J staged per chunk: as many as fit next to the fixed part (at most 64, at least 4), leaving room for two
// workgroups per CU when the system is small
static int solveF64ChunkRows(int J, int P, int U, int n, int G, int genRows) {
  const size_t base = solveF64BaseDoubles(J, P, U, n, G, genRows) * sizeof(double);
  const size_t budget = base < 60 * 1024 ? 78 * 1024 : 158 * 1024;
  if (base + size_t(n > 0 ? n : 1) * 5 * sizeof(double) > budget) {
    return 0;
  }
  size_t rows = (budget - base) / (size_t(n > 0 ? n : 1) * sizeof(double)) - 1;
  return int(rows > 64 ? 64 : rows);
}
size_t solveF64LdsBytes(int J, int P, int U, int n, int G, int genRows) {
  const int rc = solveF64ChunkRows(J, P, U, n, G, genRows);
  if (rc < 4) {
    return size_t(1) << 30; // does not fit
  }
  return (solveF64BaseDoubles(J, P, U, n, G, genRows) + ((size_t(n) * size_t(rc + 1) + 1) & ~size_t(1))) * sizeof(double);
}

// the resident instantiation: rows of J per chunk (a multiple of three, >= 12) next to the packed H, or 0 when it does
// not fit (then the scratch form runs).  Two workgroups per CU while that leaves a chunk of at least twelve rows.
int solveF64ResidentChunkRows(int J, int P, int U, int n, int G, int genRows) {
  if (n <= 0 || n > 208) { // (residentFactor's panel: 16 + 4 x 48 rows)
    return 0;
  }
  auto e = [](size_t c) { return (c + 1) & ~size_t(1); };
  const size_t base = (solveF64BaseDoubles(J, P, U, n, G, genRows) + e(size_t(n) * size_t(n + 1) / 2) + 3 * e(n)) * sizeof(double);
  for (size_t budget : {size_t(79) * 1024, size_t(158) * 1024}) {
    if (base + size_t(n) * 13 * sizeof(double) + 16 > budget) {
      continue;
    }
    size_t rows = (budget - base - 16) / (size_t(n) * sizeof(double)) - 1;
    rows = rows > 48 ? 48 : rows;
    rows -= rows % 12; // whole units (3 rows) and whole steps of the matrix cores (4 rows)
    if (rows >= 12) {
      return int(rows);
    }
  }
  return 0;
}
bool solveF64IsResident(int J, int P, int U, int n, int G, int genRows) {
  return solveF64ResidentChunkRows(J, P, U, n, G, genRows) >= 12;
}

hipError_t launchSolveF64(
    const RigDev& rig,
    const ProblemDev& pb,
    const int32_t* solveList,
    int n,
    double* theta,
    const SolveStateDev& st,
    const FusedParams& fp,
    double* Jg,
    double* Hg,
    double* Hg2,
    hipStream_t stream,
    const F64AssemblyList& list,
    const F64Select& select) {
  const int genRows = pb.rowsJoint - 3 * pb.U;
  const int rcRes = solveF64ResidentChunkRows(rig.J, rig.P, pb.U, n, pb.G + pb.NE, genRows);
  const F64AssemblyList none{nullptr, nullptr, nullptr, 0};
  if (rcRes >= 12) {
    auto e = [](size_t c) { return (c + 1) & ~size_t(1); };
    const size_t lds = (solveF64BaseDoubles(rig.J, rig.P, pb.U, n, pb.G + pb.NE, genRows) + e(size_t(n) * size_t(rcRes + 1)) +
                        e(size_t(n) * size_t(n + 1) / 2) + 3 * e(n)) * sizeof(double);
    static LdsLimitCache ldsLimitRes;
    hipError_t rc = ldsLimitRes.ensure(reinterpret_cast<const void*>(solveF64Kernel<true>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
    // (the list is only good for the chunking it was built for; per-instance constraint parents have none)
    const bool useList = list.groups != nullptr && list.unitsPerChunk == rcRes / 3 && pb.instPosParent == nullptr && pb.instOriParent == nullptr;
    hipLaunchKernelGGL(solveF64Kernel<true>, dim3(pb.B), dim3(256), lds, stream, rig, pb, solveList, n, theta, st, fp, nullptr, nullptr, Hg2, rcRes, useList ? list : none, select);
    return hipGetLastError();
  }
  const size_t lds = solveF64LdsBytes(rig.J, rig.P, pb.U, n, pb.G + pb.NE, genRows);
  if (lds > 160 * 1024) {
    return hipErrorInvalidValue;
  }
  static LdsLimitCache ldsLimit;
  {
    hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(solveF64Kernel<false>), lds);
    if (rc != hipSuccess) {
      return rc;
    }
  }
  hipLaunchKernelGGL(
      solveF64Kernel<false>, dim3(pb.B), dim3(256), lds, stream, rig, pb, solveList, n, theta, st, fp, Jg, Hg, Hg2, solveF64ChunkRows(rig.J, rig.P, pb.U, n, pb.G + pb.NE, genRows), none, select);
  return hipGetLastError();
}

hipError_t launchSelectSuspect(const int32_t* status, int B, int32_t mask, int32_t* map, int32_t* count, hipStream_t stream, int32_t require) {
  hipError_t rc = zeroAsync(count, sizeof(int32_t), stream); // (a kernel, not a memset node: mmx_kernels.hpp)
  if (rc != hipSuccess) {
    return rc;
  }
  hipLaunchKernelGGL(selectSuspectKernel, dim3((B + 255) / 256), dim3(256), 0, stream, status, B, mask, require, map, count);
  return hipGetLastError();
}

} // namespace mmx
