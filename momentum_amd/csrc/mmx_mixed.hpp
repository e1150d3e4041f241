// mmx_mixed.hpp -- the double-precision passes of the MIXED-precision one-launch solve (fusedSolveKernel<..., kMix = true>,
// mmx_gn_options::precision == MMX_PRECISION_MIXED).
//
// What limits an all-fp32 normal-equation solve is not its factor but the rounding of g = J^T r (forward kinematics -> residual
// rows -> adjoint pass), amplified by cond(J^T J + lambda I) on its way into theta (profiles/r04_fk_noise.txt,
// r05_precision_estimate.txt).  Those passes are O(J + U) work per iteration; H = J^T J and its Cholesky factor are the
// O(n^2) / O(n^3) part.  The mixed route therefore keeps theta, the joint states, the units, g and the linear solve's RESIDUAL in
// double and leaves H, the factor and the triangular solves in single precision, where they act as the PRECONDITIONER of a
// conjugate-gradient iteration on (J^T S^2 J + lambda I) d = g whose operator is applied in double through the tree (tangent
// pass down, adjoint pass up -- J is never formed): classic mixed-precision iterative refinement, accurate while
// cond x eps_f32 < 1.  The reference instantiates the whole path in double (momentum/solver/gauss_newton_solver.cpp:315-316,
// SolverT<double>); this route reproduces that instantiation's answers to ~1e-7 at close to the single-precision rate.
//
// Formulas: tests/tree_algebra_np.py (jt_times, j_times), the double twins of mmx_tree.hpp / mmx_fused.hip phase J.
#pragma once

#include "mmx_device.hpp"
#include "mmx_device_d.hpp"

namespace mmx {

constexpr int kJsD = 17; // doubles per joint: world t(3) q(4) s(1) | rotation axes x, y, z (9)  (= kJs)
constexpr int kTanD = 8; // tangent-pass doubles per joint: C(3) W(3) S(1) | jump target (int bits)
// damping floor of the mixed route's single-precision factor (fraction of the mean diagonal of J^T J; kFactorDamping of
// mmx_device.hpp for the single-precision routes).  The factor is only the preconditioner here: A/B variants mixf6 / mixf7
constexpr float kMixFactorDamping = 1e-6f;
constexpr double kLn2D = 0.693147180559945309417232121458176568; // momentum/math/constants.h:30,40

// the double arrays of one instance in LDS
struct MixLds {
  double* th; // [P] theta
  double* js; // [kJsD J] joint states
  double* up; // [3 U] unit world vectors
  double* uf; // [3 U] unit residuals f (unscaled)
  double* us; // [U] sigma = sqrt(w loss')
  double* g; // [NP] J^T r
  double* x; // [NP] the step (CG iterate)
  double* r; // [NP] CG residual
  double* p; // [NP] CG direction
  double* q; // [NP] A p
  // scratch, each over a dead predecessor (in the single-precision arena: the slot tables of phases E-G are not alive when
  // these are): X: joint parameters (FK) | joint-parameter step (tangent) | own sums | per-slot gradients ; Y: ancestor
  // prefixes of the tangent pass | subtree sums | the trial parameters of a line search / LM step
  double* X; // [max(7 J, nsrc)]
  double* Y; // [max(8 J, P)]
  int16_t* lo; // [J] by DFS position: first index into loadedPos that lies inside the joint's subtree
  int16_t* hi; // [J] ... one past the last
};
__host__ __device__ inline size_t mixXDoubles(int J, int nsrc) {
  return size_t(7 * J > nsrc ? 7 * J : nsrc);
}
__host__ __device__ inline size_t mixYDoubles(int J, int P) {
  return size_t(kTanD * J > P ? kTanD * J : P);
}
__host__ __device__ inline size_t mixPersistentDoubles(int J, int P, int U, int NP) {
  return size_t(P) + size_t(kJsD) * J + 7 * size_t(U) + 5 * size_t(NP);
}

// ---------------------------------------------------------------------------------------------
// forward kinematics in double: ParameterTransformT<double>::apply + SkeletonStateT<double>::set
// (parameter_transform.cpp:110-124, joint_state.cpp:22-65, skeleton_state.cpp:87-121; pointer-jumping composition as in
// the single-precision kernels, whose partial products are double already)
// ---------------------------------------------------------------------------------------------
// sin and cos of a joint half-angle in double without the library's large-argument path (its Payne-Hanek tables live in
// scratch memory: 464 bytes per lane of the kernel).  Two-term Cody-Waite reduction by pi/2 -- n * pio2_1 is exact for |n| < 2^20 --
// then fdlibm's kernel polynomials on [-pi/4, pi/4] (k_sin.c / k_cos.c, < 1 ulp; 1.1e-16 absolute against libm over |x| < 9e4).
// Beyond |x| = 1e5 (no pose parameter is; a diverged run can be) the argument is first folded by 2 pi in plain double.
__device__ __forceinline__ void mixSinCos(double x, double* sOut, double* cOut) {
  if (!(fabs(x) < 1e5)) { // (a diverged run: bounded values of the right period, not the library's last-bit accuracy; NaN / Inf stay NaN)
    x = fma(-rint(x * 1.59154943091895335769e-01), 6.28318530717958647693e+00, x);
  }
  const double fn = rint(x * 6.36619772367581382433e-01); // 2 / pi
  double r = fma(-fn, 1.57079632673412561417e+00, x); // first 33 bits of pi / 2
  r = fma(-fn, 6.07710050650619224932e-11, r); // pi / 2 - pio2_1
  const int q = int(fn) & 3;
  const double z = r * r;
  const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double sn = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
  const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cs = w + (((1.0 - w) - hz) + z * pc);
  const double sv = (q & 1) ? cs : sn, cv = (q & 1) ? sn : cs;
  *sOut = (q & 2) ? -sv : sv;
  *cOut = ((q + 1) & 2) ? -cv : cv;
}
__device__ __forceinline__ void fkLocalFromParamsD(const double* p, const float* pre, const float* off, double* o, double* oq) {
  double sx, cx, sy, cy, sz, cz;
  mixSinCos(0.5 * p[3], &sx, &cx);
  mixSinCos(0.5 * p[4], &sy, &cy);
  mixSinCos(0.5 * p[5], &sz, &cz);
  const DQ q0{double(pre[0]), double(pre[1]), double(pre[2]), double(pre[3])};
  const DQ q1 = dqmul(q0, DQ{0.0, 0.0, sz, cz});
  const DQ q2 = dqmul(q1, DQ{0.0, sy, 0.0, cy});
  const DQ ql = dqmul(q2, DQ{sx, 0.0, 0.0, cx});
  o[0] = double(off[0]) + p[0], o[1] = double(off[1]) + p[1], o[2] = double(off[2]) + p[2];
  o[3] = ql.x, o[4] = ql.y, o[5] = ql.z, o[6] = ql.w;
  o[7] = exp2(p[6]);
  oq[0] = q1.x, oq[1] = q1.y, oq[2] = q1.z, oq[3] = q1.w;
  oq[4] = q2.x, oq[5] = q2.y, oq[6] = q2.z, oq[7] = q2.w;
}
__device__ __forceinline__ void fkStoreLocalDD(double* bufA, int Jp, int j, const double* o, int parentPlus1) {
  FkXf x;
  x.tx = o[0], x.ty = o[1], x.tz = o[2], x.qx = o[3], x.qy = o[4], x.qz = o[5], x.qw = o[6];
  x.s = o[7], x.jl = parentPlus1;
  fkStoreD(bufA, Jp, j, x);
}
// rotationAxis.col(i) = (q_parent * q_partial) * e_i (joint_state.cpp:53-54); slots 8..15 hold q1 = pre Rz, q2 = pre Rz Ry
__device__ __forceinline__ void fkAxesInPlaceD(const float* pre, int j, int par, double* js) {
  DQ qp{0.0, 0.0, 0.0, 1.0};
  if (par >= 0) {
    const double* p = js + kJsD * par;
    qp = DQ{p[3], p[4], p[5], p[6]};
  }
  double* o = js + kJsD * j;
  const D3 az = dqrot(dqmul(qp, DQ{double(pre[0]), double(pre[1]), double(pre[2]), double(pre[3])}), D3{0.0, 0.0, 1.0});
  const D3 ay = dqrot(dqmul(qp, DQ{o[8], o[9], o[10], o[11]}), D3{0.0, 1.0, 0.0});
  const D3 ax = dqrot(dqmul(qp, DQ{o[12], o[13], o[14], o[15]}), D3{1.0, 0.0, 0.0});
  o[8] = ax.x, o[9] = ax.y, o[10] = ax.z;
  o[11] = ay.x, o[12] = ay.y, o[13] = ay.z;
  o[14] = az.x, o[15] = az.y, o[16] = az.z;
}
// translationAxis column d of a joint = column d of parent.toLinear() (joint_state.cpp:36-42)
__device__ __forceinline__ D3 transAxisColD(const double* js, int parent, int d) {
  if (parent < 0) {
    return D3{d == 0 ? 1.0 : 0.0, d == 1 ? 1.0 : 0.0, d == 2 ? 1.0 : 0.0};
  }
  const double* p = js + kJsD * parent;
  return p[7] * dqmatCol(DQ{p[3], p[4], p[5], p[6]}, d);
}

// Position / Orientation evalFunction + the weighting of JointErrorFunctionT<double>::getJacobian
// (position_error_function.cpp:15-27, orientation_error_function.cpp:15-40, joint_error_function-inl.h:197-213)
struct UnitD {
  D3 v, f;
  double sigma, werr;
};
__device__ __forceinline__ UnitD evalUnitD(const ProblemDev& pb, const UnitInput& in, const double* js, int u) {
  UnitD o;
  o.v = o.f = D3{0.0, 0.0, 0.0};
  o.sigma = o.werr = 0.0;
  if (u >= pb.U) {
    return o;
  }
  const double* w = js + kJsD * in.joint;
  const D3 t{w[0], w[1], w[2]};
  const DQ q{w[3], w[4], w[5], w[6]};
  const bool isPoint = u < pb.Kp;
  double sqr, fw;
  bool first = true;
  // (both members read up front and SELECTED BY VALUE: a reference chosen between two members -- or loads of them in the two
  // branches below, which the optimiser merges into one load of a chosen address -- is a dynamic offset into the by-value
  // descriptor, and the whole 344-byte ProblemDev then lives in scratch)
  const LossDev lsP = pb.lossPos, lsO = pb.lossOri;
  const float fwP = pb.wPos, fwO = pb.wOri;
  const LossDev ls{isPoint ? lsP.type : lsO.type, isPoint ? lsP.alpha : lsO.alpha, isPoint ? lsP.invC2 : lsO.invC2, isPoint ? lsP.c : lsO.c};
  if (isPoint) {
    o.v = t + dqrot(q, w[7] * D3{double(in.a[0]), double(in.a[1]), double(in.a[2])});
    o.f = o.v - D3{double(in.t[0]), double(in.t[1]), double(in.t[2])};
    sqr = ddot(o.f, o.f);
    fw = double(fwP);
  } else {
    const int uo = u - pb.Kp, k = uo - 3 * (uo / 3);
    // OrientationDataT<double>'s constructor normalises in double (orientation_error_function.h:33-35)
    const DQ qo = dqnormalized(DQ{double(in.a[0]), double(in.a[1]), double(in.a[2]), double(in.a[3])});
    const DQ qt = dqnormalized(DQ{double(in.t[0]), double(in.t[1]), double(in.t[2]), double(in.t[3])});
    o.v = dqrot(q, dqmatCol(qo, k));
    o.f = o.v - dqmatCol(qt, k);
    sqr = ddot(o.f, o.f);
    if (ls.type != 0) { // a robust loss sees all nine rows of the constraint
      first = k == 0;
      for (int kk = 0; kk < 3; ++kk) {
        if (kk != k) {
          const D3 fo = dqrot(q, dqmatCol(qo, kk)) - dqmatCol(qt, kk);
          sqr += ddot(fo, fo);
        }
      }
    }
    fw = double(fwO);
  }
  if (in.cw != 0.f && fw > 0.0) {
    const double wgt = double(in.cw) * fw;
    if (ls.type == 0) { // L2: error and scale split per unit (the loss is linear in |f|^2)
      const double ic = 1.0 / (double(ls.c) * double(ls.c));
      o.werr = wgt * (sqr * ic);
      o.sigma = sqrt(wgt * ic);
    } else {
      o.werr = first ? wgt * dlossValue(ls, sqr) : 0.0;
      o.sigma = sqrt(wgt * dlossDeriv(ls, sqr));
    }
  }
  return o;
}

// J^T y component of one (joint, dof) from the first-order subtree sums F(3) N(3) D (mmx_tree.hpp sourceGradient in double)
__device__ __forceinline__ double sourceGradientD(int joint, int dof, int parent, const double* js, const double* sb) {
  const double* a = js + kJsD * joint;
  const D3 ta{a[0], a[1], a[2]};
  const D3 Fv{sb[0], sb[1], sb[2]};
  if (dof < 3) {
    return ddot(transAxisColD(js, parent, dof), Fv);
  }
  if (dof < 6) {
    const double* ax = a + 8 + 3 * (dof - 3);
    const D3 Nv{sb[3], sb[4], sb[5]};
    return ddot(D3{ax[0], ax[1], ax[2]}, Nv - dcross(ta, Fv));
  }
  return kLn2D * (sb[6] - ddot(ta, Fv));
}

__device__ __forceinline__ double blockSumD(double* red, double v, int tid) {
  v = waveReduceSum(v);
  if ((tid & 63) == 0) {
    red[tid >> 6] = v;
  }
  __syncthreads();
  const double tot = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return tot;
}
// two sums with one pair of barriers
__device__ __forceinline__ void blockSum2D(double* red, double& a, double& b, int tid) {
  a = waveReduceSum(a);
  b = waveReduceSum(b);
  if ((tid & 63) == 0) {
    red[tid >> 6] = a;
    red[4 + (tid >> 6)] = b;
  }
  __syncthreads();
  a = (red[0] + red[1]) + (red[2] + red[3]);
  b = (red[4] + red[5]) + (red[6] + red[7]);
  __syncthreads();
}

} // namespace mmx
