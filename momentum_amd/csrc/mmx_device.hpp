// mmx_device.hpp -- device-side building blocks of the batched-IK kernels (gfx950, wave64).
//
// Data layout in LDS for ONE skeleton instance (all fp32):
//   jp[10*J]  per joint: tx,ty,tz, sin(rx/2),cos(rx/2), sin(ry/2),cos(ry/2), sin(rz/2),cos(rz/2), exp2(sc)
//   js[kJs*J] per joint: world t(3) q(4, xyzw) s(1) | rotationAxis columns x,y,z (3x3)
// Math follows momentum's JointStateT::set (momentum/character/joint_state.cpp:22-65),
// TransformT::operator* (momentum/math/transform.h:124-129) and the Position / Orientation
// evalFunction + ancestor walk (momentum/character_solver/position_error_function.cpp:15-27,
// orientation_error_function.cpp:15-40, joint_error_function-inl.h:179-297), re-organised so that a
// 64-lane wavefront works on one instance: lanes = joints during FK, lanes = constraint vectors
// ("units", 3 Jacobian rows each) during assembly, and every Jacobian column is GATHERED from the
// column's (joint, dof, weight) source list instead of scattered by an ancestor walk.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/mmx.h"

namespace mmx {

constexpr int kJs = 17; // floats per joint in js[]: t(3) q(4) s | axes (9); odd stride (lanes = joints read a field without LDS bank
                      // conflicts).  21 until round 5: the four pad floats per joint were what stood between the one-launch solve and a
                      // fourth workgroup per CU (288 of the 180 floats it is under the 40 KB mark now)
// Pivot threshold of every single-precision Cholesky in this library: when column j's pivot
// d_jj = (H_jj + lambda) - sum_k l_jk^2 comes out at or below kPivotFloor * (H_jj + lambda), column j is DROPPED from this
// iteration's step -- 1 / l_jj := 0, so l_ij = 0 below it and parameter j's step is exactly 0, as if the column were
// linearly dependent on the ones before it; every other parameter solves the reduced system.
// In exact arithmetic d_jj >= lambda > 0; in fp32 the subtraction leaves a rounding error of a few 1e-7 H_jj, so when J
// is rank deficient (or nearly) and lambda is small (the reference's own IK test runs lambda = 1e-7,
// inverse_kinematics_test.cpp:114) the computed pivot is noise of either sign.  The reference's double instantiation never
// sees that; its float instantiation hands Eigen's aborted factor to solve() unchecked (gauss_newton_solver.cpp:251).
// Measured on the GPU (round 3): FLOORING such pivots instead keeps noise / floor ~ O(1) steps in the directions J does
// not determine -- a scale parameter that takes one turns the next forward pass into 2^90 -- while dropping them gives a
// basic least-squares step: the objective converges like the reference's (its 3-joint known-answer test passes at its
// float tolerances), only the undetermined components of theta differ from the double solver's minimum-norm ones.
// Pivots above the threshold -- every problem whose H + lambda I is numerically positive definite -- are untouched bit
// for bit.  2^-18 = 64 ulp: a kept pivot amplifies the rounding of its column by at most ~1 / 64.
constexpr float kPivotFloor = 3.814697265625e-6f; // 2^-18
// Damping floor of every single-precision FACTOR in this library: what is factored is J^T J + max(lambda, kFactorDamping *
// mean diag(J^T J)) I.  1e-5 ~ n eps: the level at which the pivots of an fp32 Cholesky of a rank-deficient J^T J are
// rounding noise -- above it the factorisation completes whatever the column order (with the columns in elimination
// order, mmx_host_tables.hpp, the column-drop rule above would otherwise remove the ROOT's columns instead of the
// leaves': a basic step of far worse quality), below it nothing changes bit for bit (lambda = 0.05 on every BASELINE
// configuration: 1e-2 ... 1e-3 of the mean diagonal).  The STEP is still the one for the caller's lambda wherever J
// determines it: the refinement measures its residual with the true lambda through J and moves the step there; in the
// directions J does not determine the step is the more damped one (the reference's QR returns the minimum-norm-like
// step there, the objective decreases the same).  The trust-region solver used the same device from the start.
constexpr float kFactorDamping = 1e-5f;
// Precision estimate of the single-precision solves (mmx_problem_solve_diagnostics; MMX_SOLVE_PRECISION_SUSPECT):
//   est = kPrecisionGain * eps / min over every factorisation of the solve of the pivot ratio d_jj / (H_jj + lambda)
// -- eps x an estimate of cond(J^T J + lambda I): what the rounding of g = J^T r is amplified by on its way into theta,
// whatever the refinement through J does for the step.  Calibrated against the oracle's double run on 1024 instances per
// row of (BASELINE configs[0] chain / configs[1] humanoid / the all-joints variant) x lambda {5e-2, 1e-2, 1e-3, 1e-5} x
// {no line search, the driver's} x {one-launch, wide route} (scripts/diag_precision.py, profiles/r05_precision_estimate.txt):
// the smallest pivot ratio is a property of the PROBLEM CLASS (rig, constraint set, lambda) -- it varies by 10 % inside a
// batch and by decades between the rows --, and the share of instances further than 1e-5 from the double run follows it:
//   1 / ratio   3.1e2 (configs[1], 0.05)  1.0e3  1.5e3 (configs[1], 0.01)  | 7.3e3 (configs[0], 0.05)  1.5e4 (configs[1], 1e-3)  3.6e4   >= 1.7e5
//   above 1e-5  0 (worst 1.0e-6)          0      0 (worst 1.9e-6)          | 1-2 %                     1-4 %                      2-4 %   all
// The gain puts the bound's default (1e-5) at 1 / ratio = 2000 (eps = FLT_EPSILON = 1.19e-7), between the last row without
// and the first row with instances above it: est ~ the 98th percentile of the relative distance.  Which instances of a marked class end above the
// bound is not predictable from the conditioning (a discrete line-search decision, an overshooting step): the whole class
// is marked, which is what MMX_PRECISION_AUTO needs -- {above the bound} is a subset of {marked}.
// Round 6: an iteration's pivot ratio enters with the weight sqrt(e_it / e_0) (the noise of g = J^T r scales with the residual):
// with a fixed lambda the ratio is the same in every iteration and the first one's weight, 1, decides -- the figures above are
// unchanged to the digit (re-measured: 3.65e-5 / 1.82e-4 / 8.5e-4 on configs[0], 1.53e-6 / 7.59e-6 / 7.57e-5 / 9.2e-4 on
// configs[1]) --, while the LM schedule's last iterations (lambda 0.05 -> 1e-4, residual down by 1e3 and more) no longer mark
// configs[2], whose answers hold the bound (8192 of 8192 marked -> 0; est 1.5e-6).
constexpr float kPrecisionGain = 0.042f;
constexpr float kPivotFloorOrOne = kPivotFloor > 0.f ? kPivotFloor : 1.f;

constexpr float kLn2 = 0.693147180559945309417232121458176568f; // momentum/math/constants.h:30,40

struct F3 {
  float x, y, z;
};
struct Q4 { // Eigen storage order
  float x, y, z, w;
};

__device__ __forceinline__ F3 f3(float x, float y, float z) {
  return F3{x, y, z};
}
__device__ __forceinline__ F3 operator+(F3 a, F3 b) {
  return F3{a.x + b.x, a.y + b.y, a.z + b.z};
}
__device__ __forceinline__ F3 operator-(F3 a, F3 b) {
  return F3{a.x - b.x, a.y - b.y, a.z - b.z};
}
__device__ __forceinline__ F3 operator*(float s, F3 a) {
  return F3{s * a.x, s * a.y, s * a.z};
}
__device__ __forceinline__ F3 cross(F3 a, F3 b) {
  return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float dot(F3 a, F3 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z;
}
// Eigen quaternion product
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return Q4{
      a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
      a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
      a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
      a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// Eigen _transformVector: v + w*uv + qv x uv, uv = 2 (qv x v)
__device__ __forceinline__ F3 qrot(Q4 q, F3 v) {
  const F3 qv{q.x, q.y, q.z};
  F3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}
// column c of Eigen toRotationMatrix
__device__ __forceinline__ F3 qmatCol(Q4 q, int c) {
  const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
  const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  if (c == 0) {
    return F3{1.f - (tyy + tzz), txy + twz, txz - twy};
  }
  if (c == 1) {
    return F3{txy - twz, 1.f - (txx + tzz), tyz + twx};
  }
  return F3{txz + twy, tyz - twx, 1.f - (txx + tyy)};
}
__device__ __forceinline__ Q4 qnormalized(Q4 q) { // Eigen normalized(): q / sqrt(|q|^2)
  const float n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  if (n2 > 0.f) {
    const float n = sqrtf(n2);
    return Q4{q.x / n, q.y / n, q.z / n, q.w / n};
  }
  return q;
}

// ---------------------------------------------------------------------------------------------
// device views of the rig / problem tables (all pointers are device memory)
// ---------------------------------------------------------------------------------------------
struct ColumnSourceDev { // mirrors mmx::ColumnSource (mmx_host_tables.hpp), 24 bytes
  int32_t joint, dof, tin, tout, parent;
  float weight;
};

struct JacRecDev { // mirrors mmx::JacRec, 32 bytes
  int32_t joint, dof, col, tin, tout, parent;
  float weight;
  int32_t valid;
};

struct RigDev {
  int32_t J, P, R, numLevels;
  const int32_t* parent; // [J]
  const float* preRot; // [J][4]
  const float* offset; // [J][3]
  const int32_t* ptOuter; // [R+1]
  const int32_t* ptInner; // [nnz]
  const float* ptValue; // [nnz]
  const float* ptOffsets; // [R]
  int32_t ptOffsetsNonZero; // 0: every entry of ptOffsets is zero (the usual case: the solve kernels then skip the load)
  const int32_t* levelOrder; // [J]
  const int32_t* levelStart; // [numLevels+1]
  // two (parameter index, value bits) pairs per joint-parameter row, index -1 = unused slot; null
  // when some row of the transform has more than two entries (then the CSR arrays are walked)
  const int4* ptEll; // [R]
  // the NON-EMPTY rows of the transform, ascending (round 5; the one-launch solve's four-workgroup instantiations, whose LDS
  // has no room for the CSR): record t = { row | rows until the next record's row << 16 (the last record: until R), first
  // column | entries << 16, first value (float bits), offset of the first entry in ptInner / ptValue } -- ONE 16-byte load
  // per thread and walk; thread t writes its row and the (empty) rows up to the next record's, thread 0 also rows 0 .. row_0 - 1
  const int4* ptRowRec; // [numRowRec] or null (no non-empty row, or R >= 65536)
  int32_t numRowRec;
  const int32_t* jumpParent; // [J] (parent + 1) << 16 | (parent + 1): parent and initial jump target
  int32_t jumpRounds; // ceil(log2(numLevels)): pointer-jumping rounds that finish every joint
  // per-instance characters of the same topology (mmx_problem_set_instance_rig): [B][J][4] / [B][J][3] or null
  const float* instPreRot;
  const float* instOffset;
};

// element b's view of the rig: the per-instance constants replace the shared ones (a kernel calls this
// once on its by-value copy of the descriptor)
// A pointer known to point into global memory.  Pointers read out of a descriptor struct in memory (the one-launch solve's lazily
// loaded arguments) are generic to the compiler and their loads flat; through an integer the address space sticks (a plain cast
// pair is folded away): + 0.5-1 % on the one-launch solve's lines (r05_exp_fused.txt).  A/B variant noglobalptr: identity.
template <class T>
__device__ __forceinline__ T* asGlobal(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)p;
}

__device__ __forceinline__ void selectInstanceRig(RigDev& rig, int b) {
  if (rig.instPreRot != nullptr) {
    rig.preRot = rig.instPreRot + size_t(b) * 4 * size_t(rig.J);
  }
  if (rig.instOffset != nullptr) {
    rig.offset = rig.instOffset + size_t(b) * 3 * size_t(rig.J);
  }
}

// GeneralizedLossT(alpha, c) (momentum/math/generalized_loss.h:46-101, .cpp:20-155)
struct LossDev {
  int32_t type; // 0 L2, 1 L1 / pseudo-Huber, 2 Cauchy, 3 Welsch, 4 Barron's general form
  float alpha, invC2;
  float c; // the scale as given (1 when none was): GeneralizedLossT<double> forms 1 / c^2 in double (mmx_f64.hip)
};
__device__ __forceinline__ float lossValue(const LossDev& l, float s) {
  const float q = s * l.invC2;
  switch (l.type) {
    case 0:
      return q;
    case 1:
      return sqrtf(q + 1.f) - 1.f;
    case 2:
      return logf(0.5f * q + 1.f);
    case 3:
      return 1.f - expf(-0.5f * q);
    default:
      return (powf(q / fabsf(l.alpha - 2.f) + 1.f, 0.5f * l.alpha) - 1.f) * fabsf(l.alpha - 2.f) / l.alpha;
  }
}
__device__ __forceinline__ float lossDeriv(const LossDev& l, float s) {
  const float q = s * l.invC2;
  switch (l.type) {
    case 0:
      return l.invC2;
    case 1:
      return 0.5f * l.invC2 / sqrtf(q + 1.f);
    case 2:
      return l.invC2 / (l.invC2 * s + 2.f);
    case 3:
      return 0.5f * l.invC2 * expf(-0.5f * q);
    default:
      return 0.5f * l.invC2 * powf(q / fabsf(l.alpha - 2.f) + 1.f, 0.5f * l.alpha - 1.f);
  }
}

// == mmx_parameter_limit (include/mmx.h)
struct LimitDev {
  int32_t type, index0, index1;
  float weight;
  float v[4];
};

// == mmx_joint_constraint_block (include/mmx.h) for the device: one further JointErrorFunctionT
// specialisation (Plane / Aim / FixedAxis / Normal) with `count` constraints per instance
struct JointBlockDev {
  int32_t type, count;
  int32_t first; // index of the block's first constraint in the flattened list (genJoint, genTin, genBlock)
  int32_t rowStart; // first row of the block in J / r
  const float* localPoint; // [B][count][3] or null
  const float* localDir; // [B][count][3] or null
  const float* global; // [B][count][3]
  const float* planeD; // [B][count] or null
  const float* weight; // [B][count]
  float fw; // SkeletonErrorFunction::weight_
  LossDev loss;
};

// == mmx_ellipsoid_limit (include/mmx.h) + the DFS indices the kernels test ancestry with
struct EllipsoidDev {
  float ellipsoid[12], ellipsoidInv[12], offset[3], weight;
  int32_t ellipsoidParent, parent;
  int32_t tinParent; // tin[parent]
  int32_t tinStop; // tin[ellipsoidParent] when that joint is an ancestor-or-self of `parent` (the walk stops there), else -1
};

struct ProblemDev {
  int32_t B, Kp, Ko, U, M, n; // U = Kp + 3 Ko constraint vectors, M = 3 U rows, n = #enabled
  const int32_t* unitJoint; // [U] parent joint of the unit's constraint
  const int32_t* unitTin; // [U] tin[unitJoint]
  const int32_t* colStart; // [P+1]
  const ColumnSourceDev* colSources;
  const int32_t* enabledList; // [n]
  const JacRecDev* jacRecs; // [numJacRecs] single-source columns grouped by joint (multiple of 4)
  const int32_t* multiCols; // [numMultiCols]
  const int32_t* zeroCols; // [numZeroCols]
  int32_t numJacRecs, numMultiCols, numZeroCols;
  const float* posOffset; // [B][Kp][3]
  const float* posTarget; // [B][Kp][3]
  const float* posWeight; // [B][Kp]
  const float* oriOffset; // [B][Ko][4]
  const float* oriTarget; // [B][Ko][4]
  const float* oriWeight; // [B][Ko]
  float wPos, wOri; // SkeletonErrorFunction::weight_ of the two blocks
  LossDev lossPos, lossOri; // JointErrorFunctionT::loss_ of the two blocks
  // ---- parameter-space blocks (rows rowsJoint .. M-1): LimitErrorFunctionT on model parameters,
  // ModelParametersErrorFunctionT.  M = rowsJoint + NL + (hasModel ? P : 0).
  int32_t rowsJoint; // 3 U + rows of the further joint-constraint blocks + 3 NE: first parameter-space row
  // ---- further joint-constraint blocks (rows 3 U .. rowsJoint-1), explicit-Jacobian path only
  int32_t numBlocks, G; // blocks, constraints of all blocks
  const JointBlockDev* blocks; // [numBlocks]
  const int32_t* genJoint; // [G] parent joint
  const int32_t* genTin; // [G] tin[genJoint]
  const int32_t* genBlock; // [G] block of the constraint
  // ---- LimitType::Ellipsoid entries of the limit block: rows 3 U + block rows .. rowsJoint-1, three each
  int32_t NE;
  const EllipsoidDev* ellipsoids; // [NE]
  int32_t NL; // limits (one row each)
  int32_t hasModel; // model-parameter block present (P rows, used rows compacted to the top)
  const LimitDev* limits; // [NL]
  float wLimit, wModel; // weight_ of the two blocks
  const float* mpTarget; // [B][P]
  const float* mpWeights; // [B][P]
  const uint8_t* enabledMask; // [P]
  // per-instance constraint parents (mmx_problem_set_instance_parents): [B][Kp] / [B][Ko] or null, and
  // the joint -> DFS position table they are looked up in
  const int32_t* instPosParent;
  const int32_t* instOriParent;
  const int32_t* jointTin; // [J]
  // per-element error-function weights (mmx_constraint_data::function_weights): [B][fnCols] or null
  const float* fnWeights;
  int32_t fnCols;
};

// element b's view of the problem: its error-function weights folded into the block weights of the kernel's by-value
// copy of the descriptor (a kernel calls this once, like selectInstanceRig); weight_ = scalar x per-element entry
__device__ __forceinline__ void selectInstanceWeights(ProblemDev& pb, int b) {
  if (pb.fnWeights != nullptr) {
    const float* w = pb.fnWeights + size_t(b) * size_t(pb.fnCols);
    pb.wPos *= pb.fnCols > 0 ? w[0] : 1.f;
    pb.wOri *= pb.fnCols > 1 ? w[1] : 1.f;
    pb.wLimit *= pb.fnCols > 2 ? w[2] : 1.f;
    pb.wModel *= pb.fnCols > 3 ? w[3] : 1.f;
  }
}
// joint block i as element b sees it (column 4 + i of the per-element weights)
__device__ __forceinline__ JointBlockDev jointBlockOf(const ProblemDev& pb, int b, int i) {
  JointBlockDev k = pb.blocks[i];
  if (pb.fnWeights != nullptr && 4 + i < pb.fnCols) {
    k.fw *= pb.fnWeights[size_t(b) * size_t(pb.fnCols) + size_t(4 + i)];
  }
  return k;
}

// ---------------------------------------------------------------------------------------------
// Forward kinematics split into its parallel and its serial part: every joint's local transform
// and partial rotations need only theta (all joints at once), the world transforms compose
// parent * local (pointer jumping), and the rotation axes again need only the parent's world
// rotation (all joints at once).  Arithmetic per joint is the reference's (joint_state.cpp:22-65).
// ---------------------------------------------------------------------------------------------
// local transform (t, q, s: the layout of a world transform) + partial rotations q1 = pre*Qz,
// q2 = pre*Qz*Qy from the 7 joint parameters (joint_state.cpp:44-62)
__device__ __forceinline__ void
fkLocalFromParams(const float* jpv, const float* pre, const float* off, float* o, float* oq) {
  float sx, cx, sy, cy, sz, cz;
  sincosf(0.5f * jpv[3], &sx, &cx);
  sincosf(0.5f * jpv[4], &sy, &cy);
  sincosf(0.5f * jpv[5], &sz, &cz);
  const Q4 q0{pre[0], pre[1], pre[2], pre[3]};
  const Q4 q1 = qmul(q0, Q4{0.f, 0.f, sz, cz});
  const Q4 q2 = qmul(q1, Q4{0.f, sy, 0.f, cy});
  const Q4 ql = qmul(q2, Q4{sx, 0.f, 0.f, cx});
  o[0] = off[0] + jpv[0], o[1] = off[1] + jpv[1], o[2] = off[2] + jpv[2];
  o[3] = ql.x, o[4] = ql.y, o[5] = ql.z, o[6] = ql.w;
  o[7] = exp2f(jpv[6]);
  oq[0] = q1.x, oq[1] = q1.y, oq[2] = q1.z, oq[3] = q1.w;
  oq[4] = q2.x, oq[5] = q2.y, oq[6] = q2.z, oq[7] = q2.w;
}

template <class RigT>
__device__ __forceinline__ void
fkLocalSplit(const RigT& rig, int j, const float* __restrict__ theta, float* o, float* oq) {
  float jpv[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) {
    const int r = 7 * j + d;
    float acc = 0.f;
    const int k1 = rig.ptOuter[r + 1];
    for (int k = rig.ptOuter[r]; k < k1; ++k) {
      acc += rig.ptValue[k] * theta[rig.ptInner[k]];
    }
    jpv[d] = acc + rig.ptOffsets[r];
  }
  fkLocalFromParams(jpv, rig.preRot + 4 * j, rig.offset + 3 * j, o, oq);
}
template <class RigT>
__device__ __forceinline__ void fkLocalTo(const RigT& rig, int j, const float* __restrict__ theta, float* o) {
  fkLocalSplit(rig, j, theta, o, o + 8);
}

// World transforms of all joints by double-buffered pointer jumping (see fkJacobianKernel for the single-buffer
// single-precision form): `rounds` = ceil(log2(depth)) rounds, one workgroup barrier each, T_j <- T_a * T_j, a_j <- a_a.
//
// The partial products are carried in DOUBLE (round 4).  Pointer jumping re-associates SkeletonStateT::set's
// parent-before-child product; in single precision every joint's world transform then carries its own rounding history
// (each round rounds a different partial product), so the errors of a joint and of its parent are INDEPENDENT, where
// the reference's sequential product hands the parent's error on to the child (child = fl(parent * local)).  Absolute
// errors are the same (measured: 4.9e-8 m median either way on the 72-joint rig) but the relative geometry of
// neighbouring joints -- what the finger angles are solved from -- is 2-3 x noisier: the solve's distance to the double
// run sat at 1.5e-6 median / 8e-6 worst of 1024 instead of the float instantiation's 0.7e-6 / 2.6e-6
// (scripts/diag_step_noise.py, diag_g_stages.py; profiles/r04_fk_noise.txt).  With the products in double the result is
// the exact product of the single-precision local transforms rounded ONCE, which is tighter than the sequential
// single-precision product.  (v_fma_f64 issues at half the rate of v_fma_f32 on gfx950; the rounds are bound by the LDS
// round trips and the barriers, not by the arithmetic: +2.8 % on the whole solve, profiles/r04_exp_fused.txt.)
//
// Buffers: bufA / bufB, channel-major doubles [kFkCh][fkPad(J)]: t (3) q (4) s | jump target + 1 (an int in the
// ninth 64-bit slot).  Precondition (barrier done): the local transforms and the parents are in bufA (fkStoreLocalD).  The last
// round writes the world transform (t, q, s as floats) to js[kJs j + 0..7]; with rounds == 0 the caller writes js itself.
constexpr int kFkCh = 9;
__host__ __device__ __forceinline__ int fkPad(int J) {
  return (J + 1) & ~1;
}
__host__ __device__ __forceinline__ size_t fkBufFloats(int J) { // one buffer, in floats (a multiple of 4)
  return 2 * size_t(kFkCh) * size_t(fkPad(J));
}
struct FkXf {
  double tx, ty, tz, qx, qy, qz, qw, s;
  int jl;
};
__device__ __forceinline__ FkXf fkLoadD(const double* buf, int Jp, int j) {
  FkXf x;
  x.tx = buf[j], x.ty = buf[Jp + j], x.tz = buf[2 * Jp + j];
  x.qx = buf[3 * Jp + j], x.qy = buf[4 * Jp + j], x.qz = buf[5 * Jp + j], x.qw = buf[6 * Jp + j];
  x.s = buf[7 * Jp + j];
  x.jl = reinterpret_cast<const int*>(buf + 8 * Jp)[2 * j];
  return x;
}
__device__ __forceinline__ void fkStoreD(double* buf, int Jp, int j, const FkXf& x) {
  buf[j] = x.tx, buf[Jp + j] = x.ty, buf[2 * Jp + j] = x.tz;
  buf[3 * Jp + j] = x.qx, buf[4 * Jp + j] = x.qy, buf[5 * Jp + j] = x.qz, buf[6 * Jp + j] = x.qw;
  buf[7 * Jp + j] = x.s;
  reinterpret_cast<int*>(buf + 8 * Jp)[2 * j] = x.jl;
}
// the local transform o[0..7] = (t, q, s) of fkLocalFromParams and the joint's parent (+1) into bufA
__device__ __forceinline__ void fkStoreLocalD(double* bufA, int Jp, int j, const float* o, int parentPlus1) {
  FkXf x;
  x.tx = o[0], x.ty = o[1], x.tz = o[2], x.qx = o[3], x.qy = o[4], x.qz = o[5], x.qw = o[6];
  x.s = o[7], x.jl = parentPlus1;
  fkStoreD(bufA, Jp, j, x);
}
// x <- p * x (transform.h:124-129: t = tp + qp (sp t), q = qp q, s = sp s) in double; Eigen's _transformVector form
__device__ __forceinline__ void fkComposeD(const FkXf& p, FkXf& x) {
  const double sp = p.s;
  const double vx = sp * x.tx, vy = sp * x.ty, vz = sp * x.tz;
  double ux = p.qy * vz - p.qz * vy, uy = p.qz * vx - p.qx * vz, uz = p.qx * vy - p.qy * vx;
  ux += ux, uy += uy, uz += uz;
  const double tx = p.tx + (vx + p.qw * ux + (p.qy * uz - p.qz * uy));
  const double ty = p.ty + (vy + p.qw * uy + (p.qz * ux - p.qx * uz));
  const double tz = p.tz + (vz + p.qw * uz + (p.qx * uy - p.qy * ux));
  const double qx = p.qw * x.qx + p.qx * x.qw + p.qy * x.qz - p.qz * x.qy;
  const double qy = p.qw * x.qy + p.qy * x.qw + p.qz * x.qx - p.qx * x.qz;
  const double qz = p.qw * x.qz + p.qz * x.qw + p.qx * x.qy - p.qy * x.qx;
  const double qw = p.qw * x.qw - p.qx * x.qx - p.qy * x.qy - p.qz * x.qz;
  x.tx = tx, x.ty = ty, x.tz = tz, x.qx = qx, x.qy = qy, x.qz = qz, x.qw = qw;
  x.s = p.s * x.s;
  x.jl = p.jl;
}
template <class T, int kStride = kJs> // T: what the joint states are stored as (float; double in the mixed-precision solve)
__device__ __forceinline__ void fkJumpRoundsT(T* js, double* bufA, double* bufB, int J, int rounds, int tid, int nthreads) {
  const int Jp = fkPad(J);
  for (int r = 0; r < rounds; ++r) {
    const double* src = (r & 1) ? bufB : bufA;
    double* dst = (r & 1) ? bufA : bufB;
    const bool last = r == rounds - 1;
    for (int j = tid; j < J; j += nthreads) {
      FkXf x = fkLoadD(src, Jp, j);
      const int a = x.jl - 1;
      if (a >= 0) {
        const FkXf p = fkLoadD(src, Jp, a);
        fkComposeD(p, x);
      }
      if (last) {
        T* w = js + kStride * j;
        w[0] = T(x.tx), w[1] = T(x.ty), w[2] = T(x.tz);
        w[3] = T(x.qx), w[4] = T(x.qy), w[5] = T(x.qz), w[6] = T(x.qw);
        w[7] = T(x.s);
      } else {
        fkStoreD(dst, Jp, j, x);
      }
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void fkJumpRoundsD(float* js, double* bufA, double* bufB, int J, int rounds, int tid, int nthreads) {
  fkJumpRoundsT<float, kJs>(js, bufA, bufB, J, rounds, tid, nthreads);
}

// the seven joint parameters of a joint from its two-slot ELL transform rows (in registers) and theta
// in LDS.  Same products in the same order as the CSR walk (parameter_transform.cpp:124).
__device__ __forceinline__ void jointParamsFromRows(const int4* rows, const float* ptOff, const float* thetaLds, float* jpv) {
#pragma unroll
  for (int d = 0; d < 7; ++d) {
    const int4 e = rows[d];
    float acc = 0.f;
    if (e.x >= 0) {
      acc += __int_as_float(e.y) * thetaLds[e.x];
    }
    if (e.z >= 0) {
      acc += __int_as_float(e.w) * thetaLds[e.z];
    }
    jpv[d] = acc + ptOff[d];
  }
}

// fkLocalTo with the joint's seven transform rows already in registers (RigDev::ptEll) and theta
// in LDS: one round of independent global loads instead of the outer -> inner -> theta chain.
// Same products in the same order as the CSR walk (parameter_transform.cpp:124).
__device__ __forceinline__ void
fkLocalFromRows(const RigDev& rig, int j, const int4* rows, const float* ptOff, const float* thetaLds, float* o) {
  float jpv[7];
#pragma unroll
  for (int d = 0; d < 7; ++d) {
    const int4 e = rows[d];
    float acc = 0.f;
    if (e.x >= 0) {
      acc += __int_as_float(e.y) * thetaLds[e.x];
    }
    if (e.z >= 0) {
      acc += __int_as_float(e.w) * thetaLds[e.z];
    }
    jpv[d] = acc + ptOff[d];
  }
  fkLocalFromParams(jpv, rig.preRot + 4 * j, rig.offset + 3 * j, o, o + 8);
}

// One 20-float slot per joint, used in place: [0..7] holds the local (t, q_l, s) until the joint is
// composed with its ancestors, then the world (t, q, s); [8..15] holds q1, q2 until the axes pass
// replaces them by the rotation axes [8..16].
template <class RigT>
__device__ __forceinline__ void fkLocalInPlace(const RigT& rig, int j, const float* __restrict__ theta, float* js) {
  fkLocalTo(rig, j, theta, js + kJs * j);
}

__device__ __forceinline__ void fkComposeInPlaceP(int j, int par, float* js) {
  F3 tp{0.f, 0.f, 0.f};
  Q4 qp{0.f, 0.f, 0.f, 1.f};
  float sp = 1.f;
  if (par >= 0) {
    const float* p = js + kJs * par;
    tp = F3{p[0], p[1], p[2]};
    qp = Q4{p[3], p[4], p[5], p[6]};
    sp = p[7];
  }
  float* o = js + kJs * j;
  const F3 t = tp + qrot(qp, sp * F3{o[0], o[1], o[2]});
  const Q4 q = qmul(qp, Q4{o[3], o[4], o[5], o[6]});
  const float sc = sp * o[7];
  o[0] = t.x, o[1] = t.y, o[2] = t.z;
  o[3] = q.x, o[4] = q.y, o[5] = q.z, o[6] = q.w;
  o[7] = sc;
}

// rotation axes of joint j with its pre-rotation `pre` (x,y,z,w) handed in (registers)
__device__ __forceinline__ void fkAxesInPlaceQ(const float* pre, int j, int par, float* js) {
  Q4 qp{0.f, 0.f, 0.f, 1.f};
  if (par >= 0) {
    const float* p = js + kJs * par;
    qp = Q4{p[3], p[4], p[5], p[6]};
  }
  float* o = js + kJs * j;
  const F3 az = qrot(qmul(qp, Q4{pre[0], pre[1], pre[2], pre[3]}), F3{0.f, 0.f, 1.f});
  const F3 ay = qrot(qmul(qp, Q4{o[8], o[9], o[10], o[11]}), F3{0.f, 1.f, 0.f});
  const F3 ax = qrot(qmul(qp, Q4{o[12], o[13], o[14], o[15]}), F3{1.f, 0.f, 0.f});
  o[8] = ax.x, o[9] = ax.y, o[10] = ax.z;
  o[11] = ay.x, o[12] = ay.y, o[13] = ay.z;
  o[14] = az.x, o[15] = az.y, o[16] = az.z;
}
template <class RigT>
__device__ __forceinline__ void fkAxesInPlaceP(const RigT& rig, int j, int par, float* js) {
  fkAxesInPlaceQ(rig.preRot + 4 * j, j, par, js);
}


// One constraint vector ("unit") = 3 Jacobian rows 3u..3u+2.  Position constraint c -> unit c
// (a point); orientation constraint c -> units Kp+3c+k, k = 0..2 (directions = columns of
// R(world) * R(offset)).
struct Unit {
  F3 v; // world point / direction (v[k] of evalFunction)
  F3 f; // residual rows before scaling
  float sigma; // sqrt(w * loss'(.)) with the L2 loss (generalized_loss.cpp:25-32) = sqrt(w)
  float werr; // this unit's share of the block error: w * |f|^2
  int tin; // DFS index of the constraint's joint
  bool isPoint;
  bool valid;
};

// the global-memory part of a unit: issued before FK so that the HBM latency hides behind it
struct UnitInput {
  int joint, tin;
  float a[4]; // position: offset (3) ; orientation: offset quaternion
  float t[4]; // position: target (3) ; orientation: target quaternion
  float cw; // constraint weight
};

__device__ __forceinline__ UnitInput loadUnitInput(const ProblemDev& pb, int b, int u) {
  UnitInput in;
  in.joint = 0, in.tin = -1, in.cw = 0.f;
  in.a[0] = in.a[1] = in.a[2] = in.a[3] = 0.f;
  in.t[0] = in.t[1] = in.t[2] = in.t[3] = 0.f;
  if (u >= pb.U) {
    return in;
  }
  in.joint = asGlobal(pb.unitJoint)[u];
  in.tin = asGlobal(pb.unitTin)[u];
  if (u < pb.Kp) {
    const size_t c = size_t(b) * pb.Kp + u;
    if (pb.instPosParent != nullptr) { // ConstraintData::parent of THIS element
      in.joint = pb.instPosParent[c];
      in.tin = pb.jointTin[in.joint];
    }
    const float* po = asGlobal(pb.posOffset) + 3 * c;
    const float* pt = asGlobal(pb.posTarget) + 3 * c;
    in.a[0] = po[0], in.a[1] = po[1], in.a[2] = po[2];
    in.t[0] = pt[0], in.t[1] = pt[1], in.t[2] = pt[2];
    in.cw = asGlobal(pb.posWeight)[c];
  } else {
    const int co = (u - pb.Kp) / 3;
    const size_t c = size_t(b) * pb.Ko + co;
    if (pb.instOriParent != nullptr) {
      in.joint = pb.instOriParent[c];
      in.tin = pb.jointTin[in.joint];
    }
    const float* oo = asGlobal(pb.oriOffset) + 4 * c; // caller-owned pointers: no alignment assumed
    const float* ot = asGlobal(pb.oriTarget) + 4 * c;
    in.a[0] = oo[0], in.a[1] = oo[1], in.a[2] = oo[2], in.a[3] = oo[3];
    in.t[0] = ot[0], in.t[1] = ot[1], in.t[2] = ot[2], in.t[3] = ot[3];
    in.cw = asGlobal(pb.oriWeight)[c];
  }
  return in;
}

__device__ __forceinline__ Unit evalUnitFrom(const ProblemDev& pb, const UnitInput& in, const float* js, int u) {
  Unit un;
  un.valid = u < pb.U;
  un.v = F3{0.f, 0.f, 0.f};
  un.f = un.v;
  un.sigma = 0.f;
  un.werr = 0.f;
  un.tin = -1;
  un.isPoint = u < pb.Kp;
  if (!un.valid) {
    return un;
  }
  un.tin = in.tin;
  const float* w = js + kJs * in.joint;
  const F3 t{w[0], w[1], w[2]};
  const Q4 q{w[3], w[4], w[5], w[6]};
  const float s = w[7];
  float fw;
  float sqr; // |f|^2 of the whole CONSTRAINT (the argument of the loss)
  bool first = true; // the unit that reports the constraint's error
  int ltype;
  if (un.isPoint) {
    // PositionErrorFunctionT::evalFunction (position_error_function.cpp:23-26)
    un.v = t + qrot(q, s * F3{in.a[0], in.a[1], in.a[2]});
    un.f = un.v - F3{in.t[0], in.t[1], in.t[2]};
    fw = pb.wPos;
    sqr = dot(un.f, un.f);
    ltype = pb.lossPos.type;
  } else {
    // OrientationErrorFunctionT::evalFunction (orientation_error_function.cpp:23-39)
    const int uo = u - pb.Kp;
    const int k = uo - 3 * (uo / 3);
    const Q4 qo = qnormalized(Q4{in.a[0], in.a[1], in.a[2], in.a[3]}); // ctor normalises (:33-35)
    const Q4 qt = qnormalized(Q4{in.t[0], in.t[1], in.t[2], in.t[3]});
    un.v = qrot(q, qmatCol(qo, k));
    un.f = un.v - qmatCol(qt, k);
    fw = pb.wOri;
    sqr = dot(un.f, un.f);
    ltype = pb.lossOri.type;
    if (ltype != 0) { // a robust loss sees all nine rows of the constraint: add the two other columns
      first = k == 0;
      for (int kk = 0; kk < 3; ++kk) {
        if (kk != k) {
          const F3 fo = qrot(q, qmatCol(qo, kk)) - qmatCol(qt, kk);
          sqr += dot(fo, fo);
        }
      }
    }
  }
  // joint_error_function-inl.h:197-213 ; a block with weight_ <= 0 is skipped entirely
  // (skeleton_solver_function.cpp:223-231) and a constraint with weight == 0 keeps zero rows
  if (in.cw != 0.f && fw > 0.f) {
    const float wgt = in.cw * fw;
    if (ltype == 0) { // L2 (the hot path): error and scale split per unit, invC2 folded in
      const float ic = un.isPoint ? pb.lossPos.invC2 : pb.lossOri.invC2;
      un.werr = wgt * (sqr * ic);
      un.sigma = sqrtf(wgt * ic);
    } else {
      const LossDev& ls = un.isPoint ? pb.lossPos : pb.lossOri;
      un.werr = first ? wgt * lossValue(ls, sqr) : 0.f; // :207, once per constraint
      un.sigma = sqrtf(wgt * lossDeriv(ls, sqr)); // :208
    }
  }
  return un;
}

__device__ __forceinline__ Unit evalUnit(const ProblemDev& pb, const float* js, int b, int u) {
  return evalUnitFrom(pb, loadUnitInput(pb, b, u), js, u);
}

// d(unit vector)/d(joint-parameter row (a,dof)) for a source term of a column; the three
// formulas of joint_error_function-inl.h:248-291 with joint_state.cpp:68-82.
__device__ __forceinline__ F3 sourceDerivative(const ColumnSourceDev& s, const float* js, const Unit& un, bool& applies) {
  const bool anc = (s.tin <= un.tin) && (un.tin < s.tout);
  const float* a = js + kJs * s.joint;
  if (s.dof >= 3 && s.dof < 6) { // rotation: axis x (v - t_a) for points, axis x v for directions
    const float* ax = a + 8 + 3 * (s.dof - 3);
    const F3 off = un.isPoint ? un.v - F3{a[0], a[1], a[2]} : un.v;
    applies = anc;
    return cross(F3{ax[0], ax[1], ax[2]}, off);
  }
  applies = anc && un.isPoint;
  if (s.dof < 3) { // translation: column dof of parent.toLinear() = R(q_p) * s_p; identity for a root
    if (s.parent < 0) {
      return F3{s.dof == 0 ? 1.f : 0.f, s.dof == 1 ? 1.f : 0.f, s.dof == 2 ? 1.f : 0.f};
    }
    const float* p = js + kJs * s.parent;
    const F3 c = qmatCol(Q4{p[3], p[4], p[5], p[6]}, s.dof);
    return F3{c.x * p[7], c.y * p[7], c.z * p[7]};
  }
  return kLn2 * (un.v - F3{a[0], a[1], a[2]}); // scale: (v - t_a) * ln2
}

// ---------------------------------------------------------------------------------------------
// evalFunction of the further JointErrorFunctionT specialisations + the weighting of
// JointErrorFunctionT::getJacobian (joint_error_function-inl.h:197-226): one constraint with up to
// one point v_p (NumPos) and one direction v_n, FuncDim = nrows in {1, 3}, df/dv_p = dp, df/dv_n = dn
// (row-major, rows >= nrows zero).
// ---------------------------------------------------------------------------------------------
struct JointEval {
  F3 vp, vn;
  float dp[9], dn[9];
  float f[3];
  float sigma; // derivScale = sqrt(w * loss'(|f|^2)) ; 0 when the rows stay zero
  float werr; // w * loss(|f|^2)
  int nrows;
  bool hasPoint, hasDir;
};

__device__ __forceinline__ F3 normalizedOrSame(const F3& a) { // Eigen normalized(): unchanged when the norm is zero
  const float n2 = dot(a, a);
  return n2 > 0.f ? (1.f / sqrtf(n2)) * a : a;
}

__device__ __forceinline__ int jointBlockFuncDim(int type) {
  return (type == MMX_JC_AIM_DIST || type == MMX_JC_AIM_DIR || type == MMX_JC_FIXED_AXIS_DIFF) ? 3 : 1;
}

__device__ __forceinline__ JointEval evalJointConstraint(const JointBlockDev& k, const float* js, int joint, size_t c) {
  JointEval o;
  o.vp = o.vn = F3{0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    o.dp[i] = o.dn[i] = 0.f;
  }
  o.f[0] = o.f[1] = o.f[2] = 0.f;
  o.sigma = o.werr = 0.f;
  o.nrows = jointBlockFuncDim(k.type);
  o.hasPoint = k.type != MMX_JC_FIXED_AXIS_DIFF && k.type != MMX_JC_FIXED_AXIS_COS && k.type != MMX_JC_FIXED_AXIS_ANGLE;
  o.hasDir = k.type != MMX_JC_PLANE && k.type != MMX_JC_HALF_PLANE;
  const float* w = js + kJs * joint;
  const F3 t{w[0], w[1], w[2]};
  const Q4 q{w[3], w[4], w[5], w[6]};
  const float s = w[7];
  auto vec = [&](const float* a) { return F3{a[3 * c], a[3 * c + 1], a[3 * c + 2]}; };
  auto setRow = [](float* m, const F3& a, float sc) { m[0] = sc * a.x, m[1] = sc * a.y, m[2] = sc * a.z; };
  auto addOuter = [](float* m, const F3& a, const F3& b, float sc) { // m += sc * a b^T
    m[0] += sc * a.x * b.x, m[1] += sc * a.x * b.y, m[2] += sc * a.x * b.z;
    m[3] += sc * a.y * b.x, m[4] += sc * a.y * b.y, m[5] += sc * a.y * b.z;
    m[6] += sc * a.z * b.x, m[7] += sc * a.z * b.y, m[8] += sc * a.z * b.z;
  };
  const F3 gl = vec(k.global);
  if (o.hasPoint) {
    o.vp = t + qrot(q, s * vec(k.localPoint)); // state.transform * point
  }
  if (o.hasDir) {
    o.vn = qrot(q, normalizedOrSame(vec(k.localDir))); // state.rotation() * dir, normalised by the data ctor
  }
  switch (k.type) {
    case MMX_JC_PLANE:
    case MMX_JC_HALF_PLANE: { // plane_error_function.cpp:52-71
      const F3 n = normalizedOrSame(gl);
      float val = dot(o.vp, n) - k.planeD[c];
      const bool half = k.type == MMX_JC_HALF_PLANE;
      if (half && val > 0.f) {
        val = 0.f;
      }
      o.f[0] = val;
      if (!half || val < 0.f) {
        setRow(o.dp, n, 1.f);
      }
      break;
    }
    case MMX_JC_AIM_DIST: { // aim_error_function.cpp:15-36
      const F3 tgt = gl - o.vp;
      const float proj = dot(o.vn, tgt);
      const F3 r = proj * o.vn - tgt;
      o.f[0] = r.x, o.f[1] = r.y, o.f[2] = r.z;
      o.dp[0] = o.dp[4] = o.dp[8] = 1.f;
      addOuter(o.dp, o.vn, o.vn, -1.f);
      addOuter(o.dn, o.vn, tgt, 1.f);
      o.dn[0] += proj, o.dn[4] += proj, o.dn[8] += proj;
      break;
    }
    case MMX_JC_AIM_DIR: { // aim_error_function.cpp:39-67
      const F3 tgt = gl - o.vp;
      const float nrm = sqrtf(dot(tgt, tgt));
      F3 dir{0.f, 0.f, 0.f};
      if (nrm > 1e-16f) {
        dir = (1.f / nrm) * tgt;
        addOuter(o.dp, dir, dir, -1.f / nrm);
        o.dp[0] += 1.f / nrm, o.dp[4] += 1.f / nrm, o.dp[8] += 1.f / nrm;
      }
      const F3 r = o.vn - dir;
      o.f[0] = r.x, o.f[1] = r.y, o.f[2] = r.z;
      o.dn[0] = o.dn[4] = o.dn[8] = 1.f;
      break;
    }
    case MMX_JC_FIXED_AXIS_DIFF: { // fixed_axis_error_function.cpp:15-26
      const F3 r = o.vn - normalizedOrSame(gl);
      o.f[0] = r.x, o.f[1] = r.y, o.f[2] = r.z;
      o.dn[0] = o.dn[4] = o.dn[8] = 1.f;
      break;
    }
    case MMX_JC_FIXED_AXIS_COS: { // :28-39
      const F3 ga = normalizedOrSame(gl);
      o.f[0] = 1.f - dot(o.vn, ga);
      setRow(o.dn, ga, -1.f);
      break;
    }
    case MMX_JC_FIXED_AXIS_ANGLE: { // :41-66
      const F3 ga = normalizedOrSame(gl);
      const float d = dot(o.vn, ga);
      o.f[0] = acosf(fminf(fmaxf(d, -1.f), 1.f));
      const float sine = sqrtf(1.f - d * d);
      if (sine > 1e-9f) {
        setRow(o.dn, ga, -1.f / sine);
      }
      break;
    }
    default: { // MMX_JC_NORMAL, normal_error_function.cpp:14-31
      const F3 dist = o.vp - gl;
      o.f[0] = dot(o.vn, dist);
      setRow(o.dp, o.vn, 1.f);
      setRow(o.dn, dist, 1.f);
      break;
    }
  }
  const float cw = k.weight[c];
  if (cw != 0.f && k.fw > 0.f) { // :197-199 ; blocks with weight_ <= 0 are skipped (skeleton_solver_function.cpp:223-231)
    const float sqr = o.f[0] * o.f[0] + o.f[1] * o.f[1] + o.f[2] * o.f[2];
    const float wgt = cw * k.fw;
    o.werr = wgt * lossValue(k.loss, sqr); // :207
    o.sigma = sqrtf(wgt * lossDeriv(k.loss, sqr)); // :208
  }
  return o;
}

// computeEllipsoidError / the point and weights of computeEllipsoidJacobian
// (momentum/character_solver/limit_error_function.cpp:173-195,702-737): position of the constrained
// point, its offset from the projection onto the ellipsoid, jwgt = sqrt(tWeight * 1e-4 * weight).
struct EllipsoidEval {
  F3 position, diff;
  float jwgt, werr;
};
__device__ __forceinline__ F3 affineApply(const float* a, const F3& p) { // 3 x 4 row-major
  return F3{a[0] * p.x + a[1] * p.y + a[2] * p.z + a[3], a[4] * p.x + a[5] * p.y + a[6] * p.z + a[7], a[8] * p.x + a[9] * p.y + a[10] * p.z + a[11]};
}
__device__ __forceinline__ EllipsoidEval evalEllipsoid(const EllipsoidDev& ct, const float* js, float tWeight) {
  const float* wp = js + kJs * ct.parent;
  const float* we = js + kJs * ct.ellipsoidParent;
  const F3 te{we[0], we[1], we[2]};
  const Q4 qe{we[3], we[4], we[5], we[6]};
  EllipsoidEval o;
  o.position = F3{wp[0], wp[1], wp[2]} + qrot(Q4{wp[3], wp[4], wp[5], wp[6]}, wp[7] * F3{ct.offset[0], ct.offset[1], ct.offset[2]});
  const F3 local = (1.f / we[7]) * qrot(Q4{-qe.x, -qe.y, -qe.z, qe.w}, o.position - te); // transform.inverse() * position
  const F3 nrm = normalizedOrSame(affineApply(ct.ellipsoidInv, local));
  const F3 proj = affineApply(ct.ellipsoid, nrm);
  o.diff = o.position - (te + qrot(qe, we[7] * proj));
  const float w = tWeight * 1e-4f * ct.weight; // kLimitWeight * weight_ * kPositionWeight * limit.weight
  o.werr = w * dot(o.diff, o.diff);
  o.jwgt = sqrtf(w);
  return o;
}

// One row of LimitErrorFunctionT::getJacobian with the L2 loss for the limit types that act on
// model or joint parameters (momentum/character_solver/limit_error_function.cpp:
// computeMinMaxJacobian :460-503, computeMinMaxJointJacobian :503-558, computeLinearJacobian
// :561-595, computeLinearJointJacobian :597-656, computeHalfPlaneJacobian :659-695).
constexpr int kLimitEntries = 4; // non-zero Jacobian entries of a row (joint limits: <= 2 per transform row)
template <class T> // float: the single-precision kernels; double: mmx_solve_f64 (LimitErrorFunctionT<double>)
struct LimitRowT {
  int idx[kLimitEntries]; // model parameters with a non-zero entry (-1: unused)
  T coef[kLimitEntries]; // the entries
  T r; // residual entry
  T err; // this row's error term
};
using LimitRow = LimitRowT<float>;
__device__ __forceinline__ float limitSqrt(float x) {
  return sqrtf(x);
}
__device__ __forceinline__ double limitSqrt(double x) {
  return sqrt(x);
}

// joint parameter `row` of the parameter transform: value (transform * theta + offsets) and
// whether the row has an enabled column (activeJointParams, parameter_transform.cpp:97-107)
template <class T>
__device__ __forceinline__ T limitJointParam(const RigDev& rig, const T* th, const uint8_t* enabled, int row, bool& active) {
  T v = T(rig.ptOffsets[row]);
  active = false;
  const int k1 = rig.ptOuter[row + 1];
  for (int k = rig.ptOuter[row]; k < k1; ++k) {
    const int p = rig.ptInner[k];
    v += T(rig.ptValue[k]) * th[p];
    active = active || enabled[p] != 0;
  }
  return v;
}

// adds weight * T[row, :] to the entries (jacobian_jointParams_to_modelParams, error_function_utils.h:77-91:
// every column of the transform row, enabled or not; duplicates of a parameter add up)
template <class T>
__device__ __forceinline__ void limitScatterRow(const RigDev& rig, LimitRowT<T>& o, int row, T weight) {
  const int k1 = rig.ptOuter[row + 1];
  for (int k = rig.ptOuter[row]; k < k1; ++k) {
    const int p = rig.ptInner[k];
    const T c = weight * T(rig.ptValue[k]);
    bool placed = false;
#pragma unroll
    for (int e = 0; e < kLimitEntries; ++e) {
      if (!placed && (o.idx[e] == p || o.idx[e] < 0)) {
        o.coef[e] = o.idx[e] == p ? o.coef[e] + c : c;
        o.idx[e] = p;
        placed = true;
      }
    }
  }
}

template <class T>
__device__ __forceinline__ LimitRowT<T> evalLimit(const RigDev& rig, const LimitDev& lm, const T* th, const uint8_t* enabled, T tWeight) {
  LimitRowT<T> o;
#pragma unroll
  for (int e = 0; e < kLimitEntries; ++e) {
    o.idx[e] = -1;
    o.coef[e] = T(0);
  }
  o.r = o.err = T(0);
  const T wgt = limitSqrt(tWeight * T(lm.weight)); // :1018-1021
  if (lm.type == 0) { // MinMax
    const int p = lm.index0;
    if (!enabled[p]) {
      return o;
    }
    T val = T(0);
    bool hit = false;
    if (th[p] < T(lm.v[0])) {
      val = th[p] - T(lm.v[0]);
      hit = true;
    }
    if (th[p] > T(lm.v[1])) {
      val = th[p] - T(lm.v[1]);
      hit = true;
    }
    if (hit) {
      o.idx[0] = p;
      o.coef[0] = wgt;
      o.r = val * wgt;
      o.err = tWeight * T(lm.weight) * (val * val);
    }
  } else if (lm.type == 1) { // MinMaxJoint: limits on the joint parameter index0
    bool active;
    const T jp = limitJointParam(rig, th, enabled, lm.index0, active);
    if (!active) {
      return o;
    }
    T val = T(0);
    bool hit = false;
    if (jp < T(lm.v[0])) {
      val = jp - T(lm.v[0]);
      hit = true;
    } else if (jp > T(lm.v[1])) {
      val = jp - T(lm.v[1]);
      hit = true;
    }
    if (hit) {
      limitScatterRow(rig, o, lm.index0, wgt);
      o.r = val * wgt;
      o.err = tWeight * T(lm.weight) * (val * val);
    }
  } else if (lm.type == 3) { // Linear: p_ref = scale * p_tgt - offset
    const int ref = lm.index0, tgt = lm.index1;
    const float tgtF = float(th[tgt]); // isInRange takes a float (parameter_limits.cpp:105-113)
    const bool inRange = (lm.v[2] == 0.f && lm.v[3] == 0.f) || (tgtF >= lm.v[2] && tgtF < lm.v[3]);
    if ((!enabled[tgt] && !enabled[ref]) || !inRange) {
      return o;
    }
    const T rs = th[tgt] * T(lm.v[0]) - T(lm.v[1]) - th[ref];
    o.r = rs * wgt;
    if (enabled[tgt]) {
      o.idx[0] = tgt;
      o.coef[0] = T(lm.v[0]) * wgt;
    }
    if (enabled[ref]) {
      o.idx[1] = ref;
      o.coef[1] = -wgt;
    }
    if (o.idx[0] == o.idx[1]) {
      o.idx[0] = -1; // jacobian(row, ref) = ... is assigned after (row, tgt) and overwrites it (:590-594)
      o.coef[0] = T(0);
    }
    o.err = tWeight * T(lm.weight) * (rs * rs);
  } else if (lm.type == 4) { // LinearJoint: the same relation between two joint parameters
    bool actRef, actTgt;
    const T jr = limitJointParam(rig, th, enabled, lm.index0, actRef);
    const T jt = limitJointParam(rig, th, enabled, lm.index1, actTgt);
    const float jtF = float(jt);
    const bool inRange = (lm.v[2] == 0.f && lm.v[3] == 0.f) || (jtF >= lm.v[2] && jtF < lm.v[3]);
    if ((!actRef && !actTgt) || !inRange) {
      return o;
    }
    const T rs = jt * T(lm.v[0]) - T(lm.v[1]) - jr;
    o.r = rs * wgt;
    if (actTgt) {
      limitScatterRow(rig, o, lm.index1, T(lm.v[0]) * wgt);
    }
    if (actRef) {
      limitScatterRow(rig, o, lm.index0, -wgt);
    }
    o.err = tWeight * T(lm.weight) * (rs * rs);
  } else if (lm.type == 6) { // HalfPlane: (p1, p2) . normal - offset >= 0
    const int p1 = lm.index0, p2 = lm.index1;
    if (!enabled[p1] && !enabled[p2]) {
      return o;
    }
    const T rs = th[p1] * T(lm.v[0]) + th[p2] * T(lm.v[1]) - T(lm.v[2]);
    if (rs >= T(0)) {
      return o;
    }
    o.r = rs * wgt;
    if (enabled[p1]) {
      o.idx[0] = p1;
      o.coef[0] = T(lm.v[0]) * wgt;
    }
    if (enabled[p2]) {
      o.idx[1] = p2;
      o.coef[1] = T(lm.v[1]) * wgt;
    }
    if (o.idx[0] == o.idx[1]) {
      o.idx[0] = -1; // (:689-694)
      o.coef[0] = T(0);
    }
    o.err = tWeight * T(lm.weight) * (rs * rs);
  }
  return o;
}

// ---- 16x16 fp32 tile helpers shared by the fused solver and the large-system Cholesky
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTan = 9; // tangent-pass floats per joint: C(3) W(3) S(1) jump target (odd stride)

// swizzled address of element (row, col) inside a 16x16 fp32 tile (row-major, 16-byte chunks
// XOR-ed with the row group so that b128 reads of one chunk column hit distinct banks)
__device__ __forceinline__ int tileAddr(int row, int col) {
  return row * 16 + ((((col >> 2) ^ (row >> 2)) & 3) << 2) + (col & 3);
}
__device__ __forceinline__ int tileIndex(int I, int Jc) { // I >= Jc
  return I * (I + 1) / 2 + Jc;
}
__device__ __forceinline__ void tileDecode(int t, int& I, int& Jc) {
  int i = int((sqrtf(8.f * float(t) + 1.f) - 1.f) * 0.5f);
  while ((i + 1) * (i + 2) / 2 <= t) {
    ++i;
  }
  while (i * (i + 1) / 2 > t) {
    --i;
  }
  I = i;
  Jc = t - i * (i + 1) / 2;
}

__device__ __forceinline__ float readLaneF(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// One elimination step of the panel chains (a lane holds a row of the panel, a[0..15]; column j is scaled): the row's entries
// right of column j take their update, a[c] -= a[j] * L(c, j) with L(c, j) = lane c's a[j] (v_readlane).
// panelRowUpdate: TWO columns per instruction (v_pk_fma_f32 with the two multipliers as an SGPR pair; the odd leftover of an
// even j on its own): 64 instead of 120 multiply-adds per sixteen steps.  The chain is the longest VALU sequence of the
// one-launch solve and its wave shares a SIMD with three others: BASELINE configs[1] 1.83e6 -> 1.91e6 solves/s, cfg3 + 2 %
// (profiles/r05_exp_fused.txt).  The packed form rounds like the plain one only almost everywhere (last bits of the pose move).
// panelRowUpdate1: one column per instruction -- the wide route keeps it: there the packed form is worth 1.5 % (cfg5) and
// moved one of the 12 288 instances test_config5_every_instance_within_1e5 pins over the bound (1.13e-5).
// (j: the loop variable of an unrolled loop -- a constant where these are inlined)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void panelRowUpdate1(float (&a)[16], int j) {
#pragma unroll
  for (int c = j + 1; c < 16; ++c) {
    a[c] -= a[j] * readLaneF(a[j], c);
  }
}
__device__ __forceinline__ void panelRowUpdate(float (&a)[16], int j) {
  if ((j & 1) == 0) {
    a[j + 1] -= a[j] * readLaneF(a[j], j + 1);
  }
#pragma unroll
  for (int c = (j + 2) & ~1; c < 16; c += 2) {
    const v2f mlt{readLaneF(a[j], c), readLaneF(a[j], c + 1)};
    const v2f aj2{a[j], a[j]};
    v2f acc{a[c], a[c + 1]};
    acc = __builtin_elementwise_fma(-aj2, mlt, acc);
    a[c] = acc.x, a[c + 1] = acc.y;
  }
}

__device__ __forceinline__ float4 ldsRow4(const float* tile, int row, int chunk) { // 4 consecutive columns
  return *reinterpret_cast<const float4*>(tile + row * 16 + (((chunk ^ (row >> 2)) & 3) << 2));
}
// the four products as two packed multiply-adds + one add (the sum's order differs from dot4's): the one-launch solve's
// single-wave triangular solves, where the instruction count is the time (+ 2.3 % on BASELINE configs[1], r05_exp_fused.txt)
__device__ __forceinline__ float dot4pk(float4 a, float4 b, float acc) {
  v2f s{acc, 0.f};
  s = __builtin_elementwise_fma(v2f{a.x, a.y}, v2f{b.x, b.y}, s);
  s = __builtin_elementwise_fma(v2f{a.z, a.w}, v2f{b.z, b.w}, s);
  return s.x + s.y;
}
__device__ __forceinline__ float dot4(float4 a, float4 b, float acc) {
  acc += a.x * b.x;
  acc += a.y * b.y;
  acc += a.z * b.z;
  acc += a.w * b.w;
  return acc;
}

// Sum over the 64 lanes of a wave, the same value in every lane (in an SGPR): four DPP steps inside each row of sixteen
// lanes (lanes ^ 1, lanes ^ 2, half-row mirror, row mirror: no LDS crossbar trip, unlike __shfl_xor = ds_bpermute), then the
// four row sums by v_readlane.  Measured on the headline kernel (cfg2): a six-step __shfl_xor butterfly costs ~ 1 k cycles
// on a wave's critical path, this ~ 100.
template <int kCtrl>
__device__ __forceinline__ float dppMoveF(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xF, 0xF, true));
}
__device__ __forceinline__ float waveReduceSumF(float v) {
  v += dppMoveF<0xB1>(v); // quad_perm [1,0,3,2]
  v += dppMoveF<0x4E>(v); // quad_perm [2,3,0,1]
  v += dppMoveF<0x141>(v); // row_half_mirror
  v += dppMoveF<0x140>(v); // row_mirror: every lane of a row holds the row's sum
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float waveReduceMaxF(float v) { // (the same four DPP steps; a NaN does not survive fmaxf: callers that care test it first)
  v = fmaxf(v, dppMoveF<0xB1>(v));
  v = fmaxf(v, dppMoveF<0x4E>(v));
  v = fmaxf(v, dppMoveF<0x141>(v));
  v = fmaxf(v, dppMoveF<0x140>(v));
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
template <int kCtrl>
__device__ __forceinline__ double dppMoveD(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, int(b), kCtrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, int(b >> 32), kCtrl, 0xF, 0xF, true);
  return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<long long>(static_cast<unsigned int>(lo)));
}
__device__ __forceinline__ double waveReduceSum(double v) {
  v += dppMoveD<0xB1>(v);
  v += dppMoveD<0x4E>(v);
  v += dppMoveD<0x141>(v);
  v += dppMoveD<0x140>(v);
  const long long b = __double_as_longlong(v);
  auto row = [&](int lane) {
    const int lo = __builtin_amdgcn_readlane(int(b), lane), hi = __builtin_amdgcn_readlane(int(b >> 32), lane);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<long long>(static_cast<unsigned int>(lo)));
  };
  return (row(0) + row(16)) + (row(32) + row(48));
}

// this thread's share of the blocks' error at the parameters `th`.  kJacobianRows: the value
// getJacobian returns (model rows with weight <= 0 are skipped, model_parameters_error_function.cpp:113),
// else the one getError returns (:54-58).
template <bool kJacobianRows>
__device__ __forceinline__ double paramRowsError(const RigDev& rig, const ProblemDev& pb, int P, const float* th, int b, int tid) {
  double e = 0.0;
  if (pb.NL > 0 && pb.wLimit > 0.f) {
    const float tWeight = 1e+1f * pb.wLimit;
    for (int l = tid; l < pb.NL; l += 256) {
      e += double(evalLimit(rig, pb.limits[l], th, pb.enabledMask, tWeight).err);
    }
  }
  if (pb.hasModel && pb.wModel > 0.f) {
    const float* tp = pb.mpTarget + size_t(b) * P;
    const float* tw = pb.mpWeights + size_t(b) * P;
    double em = 0.0;
    for (int i = tid; i < P; i += 256) {
      if (pb.enabledMask[i] != 0) {
        const float w = tw[i];
        if (!kJacobianRows || w > 0.f) {
          const float pd = w * (th[i] - tp[i]);
          em += double(pd * pd);
        }
      }
    }
    e += em * double(pb.wModel) * double(1e-1f);
  }
  return e;
}


} // namespace mmx
