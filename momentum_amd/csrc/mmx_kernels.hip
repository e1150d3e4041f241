// mmx_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the batched-IK hot path.
//
//   fkJacobianKernel       one or four wavefronts per skeleton instance: ParameterTransform apply ->
//                          FK by pointer jumping in LDS -> residual rows -> dense Jacobian, every
//                          column gathered and written with coalesced 768-byte wave stores (the
//                          graded, HBM-write-bound kernel: mmx_eval_jacobian).
//   jointBlocksKernel / parameterRowsKernel   rows of the further error functions and of the limits.
//   normalEquations[Mfma]Kernel  H = J[:,E]^T J[:,E], g = J[:,E]^T r from the dense Jacobian
//                          (register tiles / matrix cores).
//   choleskyStep[Global]Kernel   (H + lambda I) d = g by blocked Cholesky in LDS (or with the factor in
//                          HBM), corrected-seminormal refinement through J, theta -= d, bookkeeping.
//   stepUpdateKernel       line search (both rules of the reference) / LM schedule trial steps.
//
// Reference semantics: see the citations in mmx_device.hpp and at each kernel.
#include <hip/hip_ext.h>

#include "mmx_device.hpp"
#include "mmx_kernels.hpp"

#include <algorithm>
#include <cfloat>
#include <type_traits>
#include <cstdlib>

namespace mmx {

// =============================================================================================
// Kernel 1: FK + residual + Jacobian assembly.  grid = B, block = 64 * WPI (WPI = 1 or 4 wavefronts
// per instance, launchFkJacobian), dynamic LDS = fkJacobianLdsBytes().
//
// Replaces SkeletonSolverFunctionT::initializeJacobianComputation + computeJacobianBlock
// (momentum/character_solver/skeleton_solver_function.cpp:200-261) -> JointErrorFunctionT::
// getJacobian (joint_error_function-inl.h:179-297) for the position and orientation blocks.
// Output layout: column-major M x P per instance (the reference's Eigen layout), rows 3u..3u+2 of
// unit u.  Lane u owns three consecutive floats of every column, so one wave store covers 768
// contiguous bytes; every element is written (structural zeros included), no read-modify-write.
// =============================================================================================
// the lane's three rows of a column.  kNt = streaming (non-temporal) stores.  A template parameter on
// purpose: with a run-time `bool nt` argument the optimiser merged the two branches of the helper
// before inlining it and dropped the nontemporal flag (no `global_store ... nt` was ever emitted).
template <bool kNt>
__device__ __forceinline__ void store3(float* o, float x, float y, float z) {
  if constexpr (kNt) {
    __builtin_nontemporal_store(x, o);
    __builtin_nontemporal_store(y, o + 1);
    __builtin_nontemporal_store(z, o + 2);
  } else {
    o[0] = x, o[1] = y, o[2] = z;
  }
}

// vmcnt counts loads and stores in ONE in-order counter, so a wait for a load that is placed after a
// store drains every store issued in between.  MMX_ARRIVED ties the registers of earlier loads to a
// value the following stores depend on (`dep`, e.g. the zero they write): the compiler has to wait
// for those loads before the first store, and knows afterwards that they have arrived.  Not
// `volatile`: a volatile asm counts as a write to unknown memory, which turns every wave-uniform
// table load after it from a scalar load into a vector load.
#define MMX_ARRIVED4(dep_, a_, b_, c_, d_) asm("" : "+v"(dep_) : "v"(a_), "v"(b_), "v"(c_), "v"(d_))

// The column program for rigs with more units than lanes: a lane carries up to kChunks units
// (u = u0 + 64 c + lane) and a column is written chunk after chunk, back to back, so that the
// pieces of a column (M * 4 bytes, not a multiple of a 128-byte line in general: 3600 B at cfg5)
// meet in the write-combining L2 instead of reaching HBM as partial lines.  Plain stores for the
// same reason.  Measured on the bare store pattern at cfg5 (scripts/store_cfg5.hip): chunk-outer
// 3.1 TB/s (3.3 streaming), column-outer 5.6 TB/s (4.3 streaming).
constexpr int kChunks = 6;
struct UnitLite {
  F3 v;
  float sigma;
  int tin;
  bool isPoint, valid;
};
template <int WPI>
__device__ __forceinline__ void
writeUnitColumnsMulti(const ProblemDev& pb, const float* js, const UnitLite* un, float* jb, size_t M, int wave) {
  int curJoint = -1;
  bool anc[kChunks];
  F3 off[kChunks];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    anc[c] = false;
    off[c] = F3{0.f, 0.f, 0.f};
  }
  for (int i0 = 4 * wave; i0 < pb.numJacRecs; i0 += 4 * WPI) {
    JacRecDev rec[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rec[k] = pb.jacRecs[i0 + k]; // wave-uniform
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const JacRecDev& r = rec[k];
      const float* a = js + kJs * r.joint;
      if (r.joint != curJoint) {
        curJoint = r.joint;
        const F3 t{a[0], a[1], a[2]};
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
          anc[c] = (r.tin <= un[c].tin) && (un[c].tin < r.tout);
          off[c] = un[c].isPoint ? un[c].v - t : un[c].v;
        }
      }
      const float* axp = a + 8 + 3 * (r.dof - 3);
      const F3 ax{axp[0], axp[1], axp[2]};
      float* o = jb + size_t(r.col) * M;
#pragma unroll
      for (int c = 0; c < kChunks; ++c) {
        const F3 g = cross(ax, off[c]);
        const float w = anc[c] ? r.weight : 0.f;
        if (un[c].valid) {
          store3<false>(o + 192 * c, (un[c].sigma * g.x) * w, (un[c].sigma * g.y) * w, (un[c].sigma * g.z) * w);
        }
      }
    }
  }
  for (int i = wave; i < pb.numMultiCols; i += WPI) {
    const int p = pb.multiCols[i];
    F3 acc[kChunks];
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      acc[c] = F3{0.f, 0.f, 0.f};
    }
    const int e1 = pb.colStart[p + 1];
    for (int e = pb.colStart[p]; e < e1; ++e) {
      const ColumnSourceDev s = pb.colSources[e]; // wave-uniform
#pragma unroll
      for (int c = 0; c < kChunks; ++c) {
        Unit u1;
        u1.v = un[c].v, u1.tin = un[c].tin, u1.isPoint = un[c].isPoint;
        bool applies;
        const F3 g = sourceDerivative(s, js, u1, applies);
        const float w = applies ? s.weight : 0.f;
        acc[c].x += (un[c].sigma * g.x) * w;
        acc[c].y += (un[c].sigma * g.y) * w;
        acc[c].z += (un[c].sigma * g.z) * w;
      }
    }
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      if (un[c].valid) {
        store3<false>(jb + size_t(p) * M + 192 * c, acc[c].x, acc[c].y, acc[c].z);
      }
    }
  }
}

// the structurally zero columns of an instance: lane = unit u writes rows 3u..3u+2 of each
// (the columns are dealt to the waves w0 .. w0 + nw - 1 of the workgroup)
template <bool kNt>
__device__ __forceinline__ void writeZeroColumns(const ProblemDev& pb, float* jz, float zero, int lane, int wave, int w0, int nw) {
  if (wave < w0) {
    return;
  }
  for (int u0 = 0; u0 < pb.U; u0 += 64) {
    const int u = u0 + lane;
    if (u < pb.U) {
      for (int i = wave - w0; i < pb.numZeroCols; i += nw) {
        store3<kNt>(jz + size_t(pb.zeroCols[i]) * size_t(pb.M) + 3 * size_t(u), zero, zero, zero);
      }
    }
  }
}

// The column program of one unit (lane = unit u, rows 3u..3u+2 of every non-zero column).
template <int WPI, bool kNt>
__device__ __forceinline__ void
writeUnitColumns(const ProblemDev& pb, const float* js, const Unit& un, float* jb, size_t M, int wave) {
  // (1) single-source rotation columns, grouped by joint: four records (one 128-byte run of
  //     scalar loads) per trip; ancestor test and v - t_joint refreshed when the joint changes.
  //     jc = derivScale * dfdv * (axis x off) ; jac.col(p) = jc * value
  //     (joint_error_function-inl.h:265-278, joint_state.cpp:68-71)
  int curJoint = -1;
  bool anc = false;
  F3 off{0.f, 0.f, 0.f};
  for (int i0 = 4 * wave; i0 < pb.numJacRecs; i0 += 4 * WPI) {
    JacRecDev rec[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rec[k] = pb.jacRecs[i0 + k]; // wave-uniform
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const JacRecDev& r = rec[k];
      const float* a = js + kJs * r.joint;
      if (r.joint != curJoint) {
        curJoint = r.joint;
        anc = (r.tin <= un.tin) && (un.tin < r.tout);
        off = un.isPoint ? un.v - F3{a[0], a[1], a[2]} : un.v;
      }
      const float* ax = a + 8 + 3 * (r.dof - 3);
      const F3 g = cross(F3{ax[0], ax[1], ax[2]}, off);
      const float w = anc ? r.weight : 0.f;
      if (un.valid) {
        float* o = jb + size_t(r.col) * M;
        store3<kNt>(o, (un.sigma * g.x) * w, (un.sigma * g.y) * w, (un.sigma * g.z) * w);
      }
    }
  }
  // (2) every other non-empty column (shared parameters, translation / scale dofs): generic gather
  for (int i = wave; i < pb.numMultiCols; i += WPI) {
    const int p = pb.multiCols[i];
    F3 acc{0.f, 0.f, 0.f};
    const int e1 = pb.colStart[p + 1];
    for (int e = pb.colStart[p]; e < e1; ++e) {
      const ColumnSourceDev s = pb.colSources[e]; // wave-uniform
      bool applies;
      const F3 g = sourceDerivative(s, js, un, applies);
      const float w = applies ? s.weight : 0.f;
      acc.x += (un.sigma * g.x) * w;
      acc.y += (un.sigma * g.y) * w;
      acc.z += (un.sigma * g.z) * w;
    }
    if (un.valid) {
      store3<kNt>(jb + size_t(p) * M, acc.x, acc.y, acc.z);
    }
  }
}

// dynamic LDS of fkJacobianKernel in floats: joint slots | theta | packed parent table | evaluated units of the multi-chunk J path
__host__ __device__ __forceinline__ size_t fkJacobianLdsFloats(int J, int P, int U) {
  const size_t unitStash = U > 64 ? 5 * size_t(U) : 0;
  return ((size_t(kJs) * size_t(J) + 3) & ~size_t(3)) + ((size_t(P) + 3) & ~size_t(3)) + ((size_t(J) + 3) & ~size_t(3)) + unitStash;
}

// WPI = wavefronts per instance: 1 (block = 64) for large batches, 4 (block = 256: FK over 256
// threads, the column program dealt to the four waves) when the batch alone cannot fill the chip.
template <bool kWriteJac, int WPI, bool kStream>
__global__ void __launch_bounds__(64 * WPI) fkJacobianKernel(
    RigDev rig,
    ProblemDev pb,
    const float* __restrict__ theta, // [B][P]
    float* __restrict__ jac, // [B][M*P] column-major, or null
    float* __restrict__ res, // [B][M] or null
    double* __restrict__ err, // [B] or null
    float* __restrict__ state, // [B][J][8] or null
    const int32_t* __restrict__ done, // [B] or null: skip finished instances
    int zeroPhase) { // when the structurally zero columns are written: 0 first, 1 alternating, 2 last
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // one 20-float slot per joint, used in place: [0..7] local t,s,q -> world t,q,s ; [8..15] partial
  // rotations q1,q2 -> [8..16] rotation axes
  float* js = smem;
  float* thL = js + ((kJs * rig.J + 3) & ~3); // [P] theta of this instance
  int* jl = reinterpret_cast<int*>(thL + ((rig.P + 3) & ~3)); // [J] (parent + 1) << 16 | (jump target + 1)
  float* ul = reinterpret_cast<float*>(jl + ((rig.J + 3) & ~3)); // [U][5] evaluated units (v, sigma, tin), only when U > 64 and J is written
  const bool fkDouble = (zeroPhase & 0x100) != 0; // the jump rounds in double: two channel-major buffers behind the unit stash
  zeroPhase &= 0xff;
  double* fkA = reinterpret_cast<double*>(smem + ((fkJacobianLdsFloats(rig.J, rig.P, pb.U) + 1) & ~size_t(1)));
  double* fkB = fkA + fkBufFloats(rig.J) / 2;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  selectInstanceRig(rig, b);
  selectInstanceWeights(pb, b);
  // wave-uniform on purpose: it indexes the column program, which must stay on the scalar unit
  const int wave = WPI == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = 64 * WPI;
  if (done != nullptr && done[b] != 0) {
    return;
  }
  // workgroups go round-robin to the 8 XCDs and then to an XCD's CUs: bits 3.. pick the CU, bits 8..
  // the slot on it -- both kinds of instance on every CU, alternating between neighbouring CUs
  const bool manyUnits = kWriteJac && pb.U > 64; // several 64-unit chunks: see the unit loop
  const bool zeroLast = zeroPhase == 2 || (zeroPhase == 1 && (((b >> 3) ^ (b >> 8)) & 1) != 0);
  const float* th = theta + size_t(b) * rig.P;

  // Everything the instance needs from global memory is requested up front, in ONE round of
  // independent loads: theta and the level/parent table (-> LDS), the first joint's transform
  // rows, the pre-rotations of the thread's first two joints and the constraint payload of the
  // first 64 units (-> registers).  FK then runs on LDS, and -- when J is written -- nothing is
  // loaded from global memory any more until the last column store has been issued: vmcnt is ONE
  // in-order counter for loads and stores, so any wait for a load after the first store would
  // drain all the stores issued before it (measured: the zero-column stores used to be waited for
  // in full before FK began).
  const bool ell = rig.ptEll != nullptr;
  int4 rows[7];
  float ptOff[7];
  float preA[4] = {0.f, 0.f, 0.f, 1.f}, preB[4] = {0.f, 0.f, 0.f, 1.f}, offA[3] = {0.f, 0.f, 0.f};
  if (tid < rig.J) {
    if (ell) {
#pragma unroll
      for (int d = 0; d < 7; ++d) {
        rows[d] = rig.ptEll[7 * tid + d];
        ptOff[d] = rig.ptOffsets[7 * tid + d];
      }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      preA[d] = rig.preRot[4 * tid + d];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      offA[d] = rig.offset[3 * tid + d];
    }
  }
  if (tid + NT < rig.J) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      preB[d] = rig.preRot[4 * (tid + NT) + d];
    }
  }
  const bool needUnits = kWriteJac || res != nullptr || err != nullptr;
  UnitInput uin0 = loadUnitInput(pb, b, needUnits ? lane : pb.U);
  for (int i = tid; i < rig.P; i += NT) {
    thL[i] = th[i];
  }
  for (int i = tid; i < rig.J; i += NT) {
    jl[i] = rig.jumpParent[i];
  }
  __syncthreads();
  // local transforms of all joints at once (ParameterTransformT::apply + the theta-only part of
  // JointStateT::set); SkeletonStateT::set's parent-before-child sweep follows as pointer jumping
  // that only composes world = parent * local, then the rotation axes of all joints at once
  if (ell) {
    if (tid < rig.J) {
      float jpv[7];
      jointParamsFromRows(rows, ptOff, thL, jpv);
      fkLocalFromParams(jpv, preA, offA, js + kJs * tid, js + kJs * tid + 8);
    }
    for (int j = tid + NT; j < rig.J; j += NT) {
#pragma unroll
      for (int d = 0; d < 7; ++d) {
        rows[d] = rig.ptEll[7 * j + d];
        ptOff[d] = rig.ptOffsets[7 * j + d];
      }
      fkLocalFromRows(rig, j, rows, ptOff, thL, js + kJs * j);
    }
  } else {
    for (int j = tid; j < rig.J; j += NT) {
      fkLocalInPlace(rig, j, thL, js);
    }
  }
  if (kWriteJac) {
    // every load above has to have arrived before the first store (see the top of the kernel)
    float zero = 0.f;
    MMX_ARRIVED4(zero, preA[0], preA[1], preA[2], preA[3]);
    MMX_ARRIVED4(zero, preB[0], preB[1], preB[2], preB[3]);
    MMX_ARRIVED4(zero, uin0.a[0], uin0.a[1], uin0.a[2], uin0.a[3]);
    MMX_ARRIVED4(zero, uin0.t[0], uin0.t[1], uin0.t[2], uin0.t[3]);
    MMX_ARRIVED4(zero, uin0.joint, uin0.tin, uin0.cw, uin0.cw);
    // structurally zero columns (disabled parameters, and parameters none of whose joints carries a
    // constraint below it) need no kinematics (every element of J is still written).  Half of the
    // instances write them right here, so that their stores keep HBM busy while FK runs; the
    // other half writes them last: a wave cannot run ahead of the stores it has issued by more
    // than the depth of the CU's store path, so if every wave began with these stores they would
    // all sit here until the stores have drained and then all run FK with HBM idle.
    // With several waves per instance the waves that hold no joint (72 joints: waves 2 and 3 of 4)
    // write them while the others run FK.
    if (!zeroLast && !manyUnits) {
      const int fkWaves = (rig.J + 63) >> 6;
      const int w0 = (WPI > fkWaves && zeroPhase != 3) ? fkWaves : 0;
      writeZeroColumns<kStream>(pb, jac + size_t(b) * size_t(pb.M) * size_t(rig.P), zero, lane, wave, w0, WPI - w0);
    }
  }
  __syncthreads();
  // World transforms by pointer jumping instead of a sweep over the tree levels: in every round
  // each joint composes its partial product with the one of its current jump target and inherits
  // that joint's target, T_j <- T_a * T_j, a_j <- a_a (parent-before-child composition of
  // SkeletonStateT::set, skeleton_state.cpp:100-121, re-associated).  ceil(log2(depth)) rounds with
  // every lane busy replace `depth` rounds with a handful of lanes each.  A joint may read a
  // target that was already advanced in the same round: (T_a, a_a) are always read as a
  // consistent pair, which keeps the invariant "T_j = product of the locals on the path
  // (a_j, j]" -- within one wave by program order, across waves by the two barriers.
  if (fkDouble) { // (the locals are in js[0..7]; the result comes back there)
    const int Jp = fkPad(rig.J);
    if (rig.jumpRounds > 0) {
      for (int j = tid; j < rig.J; j += NT) {
        fkStoreLocalD(fkA, Jp, j, js + kJs * j, jl[j] >> 16);
      }
      __syncthreads();
      fkJumpRoundsD(js, fkA, fkB, rig.J, rig.jumpRounds, tid, NT);
    }
  }
  for (int r = 0; r < (fkDouble ? 0 : rig.jumpRounds); ++r) {
    for (int j0 = 0; j0 < rig.J; j0 += NT) {
      const int j = j0 + tid;
      const int mine = j < rig.J ? jl[j] : 0;
      const int a = (mine & 0xffff) - 1;
      float* o = js + kJs * (j < rig.J ? j : 0);
      F3 t{0.f, 0.f, 0.f};
      Q4 q{0.f, 0.f, 0.f, 1.f};
      float sc = 1.f;
      int next = 0;
      if (a >= 0) {
        const float* p = js + kJs * a;
        const F3 tp{p[0], p[1], p[2]};
        const Q4 qp{p[3], p[4], p[5], p[6]};
        const float sp = p[7];
        next = jl[a] & 0xffff;
        t = tp + qrot(qp, sp * F3{o[0], o[1], o[2]}); // transform.h:124-129
        q = qmul(qp, Q4{o[3], o[4], o[5], o[6]});
        sc = sp * o[7];
      }
      if (WPI > 1) {
        __syncthreads();
      }
      if (a >= 0) {
        o[0] = t.x, o[1] = t.y, o[2] = t.z;
        o[3] = q.x, o[4] = q.y, o[5] = q.z, o[6] = q.w;
        o[7] = sc;
        jl[j] = (mine & ~0xffff) | next;
      }
      if (WPI > 1) {
        __syncthreads();
      }
    }
  }
  __syncthreads();
  if (kWriteJac) {
    if (tid < rig.J) {
      fkAxesInPlaceQ(preA, tid, (jl[tid] >> 16) - 1, js);
    }
    if (tid + NT < rig.J) {
      fkAxesInPlaceQ(preB, tid + NT, (jl[tid + NT] >> 16) - 1, js);
    }
    for (int j = tid + 2 * NT; j < rig.J; j += NT) { // rigs beyond 2 joints per thread: loads (and waits) here
      fkAxesInPlaceP(rig, j, (jl[j] >> 16) - 1, js);
    }
    __syncthreads();
  }
  if (state != nullptr) {
    float* so = state + size_t(b) * rig.J * 8;
    for (int i = tid; i < rig.J * 8; i += NT) {
      so[i] = js[kJs * (i >> 3) + (i & 7)];
    }
  }
  if (!kWriteJac && res == nullptr && err == nullptr) {
    return;
  }

  double errAcc = 0.0;
  const size_t M = size_t(pb.M);
  if (manyUnits) {
    // More units than lanes (cfg5: 300): every unit is evaluated BEFORE the first column store --
    // one unit per thread and trip, the result (v, sigma, DFS index) stashed in LDS -- so that the
    // payload loads of the later chunks do not sit between column stores (each wait for one would
    // drain the stores issued so far).  Then zero columns and column program per 64-unit chunk.
    for (int u = tid; u < pb.U; u += NT) {
      const Unit un = evalUnitFrom(pb, u == lane ? uin0 : loadUnitInput(pb, b, u), js, u);
      errAcc += double(un.werr);
      if (res != nullptr) {
        store3<false>(res + size_t(b) * M + 3 * size_t(u), un.sigma * un.f.x, un.sigma * un.f.y, un.sigma * un.f.z);
      }
      float* o = ul + 5 * u;
      o[0] = un.v.x, o[1] = un.v.y, o[2] = un.v.z, o[3] = un.sigma, o[4] = __int_as_float(un.tin);
    }
    __syncthreads();
    float* jz = jac + size_t(b) * M * size_t(rig.P);
    // zero columns, column-outer as well (plain stores: see writeUnitColumnsMulti)
    for (int i = wave; i < pb.numZeroCols; i += WPI) {
      float* o = jz + size_t(pb.zeroCols[i]) * M;
      for (int u = lane; u < pb.U; u += 64) {
        store3<false>(o + 3 * size_t(u), 0.f, 0.f, 0.f);
      }
    }
    for (int u0 = 0; u0 < pb.U; u0 += 64 * kChunks) {
      UnitLite un[kChunks];
#pragma unroll
      for (int c = 0; c < kChunks; ++c) {
        const int u = u0 + 64 * c + lane;
        un[c].valid = u < pb.U;
        un[c].isPoint = u < pb.Kp;
        const float* o = ul + 5 * (un[c].valid ? u : 0);
        un[c].v = F3{o[0], o[1], o[2]};
        un[c].sigma = un[c].valid ? o[3] : 0.f;
        un[c].tin = un[c].valid ? __float_as_int(o[4]) : -1;
      }
      writeUnitColumnsMulti<WPI>(pb, js, un, jz + 3 * size_t(u0 + lane), M, wave);
    }
    if (WPI > 1) { // the per-thread error shares are summed over the workgroup through LDS
      __syncthreads();
      const double e = waveReduceSum(errAcc);
      double* red = reinterpret_cast<double*>(ul);
      if (lane == 0) {
        red[wave] = e;
      }
      __syncthreads();
      errAcc = 0.0;
      if (tid == 0) {
        for (int w = 0; w < WPI; ++w) {
          errAcc += red[w];
        }
      }
    }
  } else {
  // the first 64 units: their payload is already in registers, so there is no load -- hence no
  // vmcnt wait -- between the zero-column stores and the column stores
  {
    const Unit un = evalUnitFrom(pb, uin0, js, lane);
    errAcc += double(un.werr);
    if (res != nullptr && un.valid && wave == 0) {
      store3<false>(res + size_t(b) * M + 3 * size_t(lane), un.sigma * un.f.x, un.sigma * un.f.y, un.sigma * un.f.z);
    }
    if (kWriteJac) {
      writeUnitColumns<WPI, kStream>(pb, js, un, jac + size_t(b) * M * size_t(rig.P) + 3 * size_t(lane), M, wave);
    }
  }
  for (int u0 = 64; u0 < pb.U; u0 += 64) { // (only without J: residual / error of the further chunks)
    const int u = u0 + lane;
    const Unit un = evalUnitFrom(pb, loadUnitInput(pb, b, u), js, u);
    errAcc += double(un.werr);
    if (res != nullptr && un.valid && wave == 0) {
      store3<false>(res + size_t(b) * M + 3 * size_t(u), un.sigma * un.f.x, un.sigma * un.f.y, un.sigma * un.f.z);
    }
  }
  }
  if (kWriteJac && zeroLast && !manyUnits) {
    writeZeroColumns<kStream>(pb, jac + size_t(b) * M * size_t(rig.P), 0.f, lane, wave, 0, WPI);
  }
  if (err != nullptr) {
    const double e = waveReduceSum(errAcc);
    if (tid == 0) {
      err[b] = e;
    }
  }
}

// =============================================================================================
// Kernel 1b: the parameter-space rows of J / r (rows rowsJoint .. M-1 of every column) and their
// error.  grid = B, block = 256.  Launched after fkJacobianKernel when the problem carries limits
// or model-parameter targets; adds its error to err[b].
//
// Replaces LimitErrorFunctionT::getJacobian for the model-parameter limit types with the L2 loss
// (momentum/character_solver/limit_error_function.cpp:992-1122: computeMinMaxJacobian :460-503,
// computeLinearJacobian :561-595, computeHalfPlaneJacobian :659-695; kLimitWeight
// limit_error_function.h:91) and ModelParametersErrorFunctionT::getJacobian
// (model_parameters_error_function.cpp:95-131; used rows compacted like `out` there).
// =============================================================================================
__global__ void __launch_bounds__(256) parameterRowsKernel(
    RigDev rig,
    ProblemDev pb,
    int P,
    const float* __restrict__ theta,
    float* __restrict__ jac, // or null
    float* __restrict__ res, // or null
    double* __restrict__ err, // or null: err[b] += error of these blocks
    const int32_t* __restrict__ done) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (done != nullptr && done[b] != 0) {
    return;
  }
  selectInstanceWeights(pb, b);
  const int NL = pb.NL, R0 = pb.rowsJoint;
  const int Pm = pb.hasModel ? P : 0;
  int* outOf = reinterpret_cast<int*>(smem); // [P] compacted model row of parameter i, or -1
  int* eidx = outOf + P; // [kLimitEntries][NL] non-zero entries of the limit rows
  float* ecoef = reinterpret_cast<float*>(eidx + kLimitEntries * NL);
  double* red = reinterpret_cast<double*>(smem + ((P + 2 * kLimitEntries * NL + 1) & ~1)); // [4]
  __shared__ int numUsed;
  const float* th = theta + size_t(b) * P;
  const size_t M = size_t(pb.M);
  float* rb = res != nullptr ? res + size_t(b) * M : nullptr;
  double e = 0.0;
  // ---- limits: one thread per limit
  const bool limOn = pb.wLimit > 0.f; // a block with weight_ <= 0 is skipped (skeleton_solver_function.cpp:228,250)
  const float tWeight = 1e+1f * pb.wLimit;
  for (int l = tid; l < NL; l += 256) {
    LimitRow o;
#pragma unroll
    for (int e = 0; e < kLimitEntries; ++e) {
      o.idx[e] = -1;
      o.coef[e] = 0.f;
    }
    o.r = o.err = 0.f;
    if (limOn) {
      o = evalLimit(rig, pb.limits[l], th, pb.enabledMask, tWeight);
    }
#pragma unroll
    for (int e = 0; e < kLimitEntries; ++e) {
      eidx[e * NL + l] = o.idx[e];
      ecoef[e * NL + l] = o.coef[e];
    }
    if (rb != nullptr) {
      rb[R0 + l] = o.r;
    }
    e += double(o.err);
  }
  // ---- model-parameter rows: compaction map by wave 0 (ballot prefix), then one thread per parameter
  const bool mpOn = Pm > 0 && pb.wModel > 0.f;
  const float* tp = Pm > 0 ? pb.mpTarget + size_t(b) * P : nullptr;
  const float* tw = Pm > 0 ? pb.mpWeights + size_t(b) * P : nullptr;
  if (Pm > 0 && wave == 0) {
    int base = 0;
    for (int i0 = 0; i0 < P; i0 += 64) {
      const int i = i0 + lane;
      const bool f = mpOn && i < P && pb.enabledMask[i] != 0 && tw[i] > 0.f;
      const unsigned long long m = __ballot(f);
      if (i < P) {
        outOf[i] = f ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
      }
      base += __popcll(m);
    }
    if (lane == 0) {
      numUsed = base;
    }
  }
  __syncthreads();
  if (Pm > 0) {
    const float sWeight = sqrtf(pb.wModel * 1e-1f); // kMotionWeight, model_parameters_error_function.h:61
    double em = 0.0;
    for (int i = tid; i < P; i += 256) {
      const int out = outOf[i];
      if (out >= 0) {
        const float pdiff = tw[i] * (th[i] - tp[i]);
        em += double(pdiff * pdiff);
        if (rb != nullptr) {
          rb[R0 + NL + out] = pdiff * sWeight;
        }
      }
    }
    e += em * double(pb.wModel) * double(1e-1f);
    if (rb != nullptr) {
      for (int r = numUsed + tid; r < P; r += 256) {
        rb[R0 + NL + r] = 0.f;
      }
    }
  }
  if (err != nullptr) {
    e = waveReduceSum(e);
    if (lane == 0) {
      red[wave] = e;
    }
    __syncthreads();
    if (tid == 0) {
      err[b] += (red[0] + red[1]) + (red[2] + red[3]);
    }
  }
  // ---- the Jacobian rows: column p = a wave, rows = lanes; every element is written
  if (jac != nullptr) {
    const int R = NL + Pm;
    const float sWeight = Pm > 0 ? sqrtf(pb.wModel * 1e-1f) : 0.f;
    for (int p = wave; p < P; p += 4) {
      float* col = jac + size_t(b) * M * size_t(P) + size_t(p) * M + R0;
      const int mine = Pm > 0 ? outOf[p] : -1;
      const float mval = mine >= 0 ? sWeight * tw[p] : 0.f;
      for (int rr = lane; rr < R; rr += 64) {
        float v = 0.f;
        if (rr < NL) {
#pragma unroll
          for (int e = 0; e < kLimitEntries; ++e) {
            v += eidx[e * NL + rr] == p ? ecoef[e * NL + rr] : 0.f;
          }
        } else if (rr - NL == mine) {
          v = mval;
        }
        col[rr] = v;
      }
    }
  }
}

size_t parameterRowsLdsBytes(int P, int NL) {
  return (size_t((P + 2 * kLimitEntries * NL + 1) & ~1) + 8) * sizeof(float);
}

// =============================================================================================
// Kernel 1c: the rows of the further joint-constraint blocks (Plane / HalfPlane / AimDist / AimDir /
// FixedAxisDiff / Cos / Angle / Normal error functions; rows 3 U .. rowsJoint-1 of every column),
// their residual and error.  grid = B, block = 256.  Launched after fkJacobianKernel when the
// problem carries such blocks; adds its error to err[b].
//
// Replaces JointErrorFunctionT<T, Data, FuncDim, NumVec, NumPos>::getJacobian
// (momentum/character_solver/joint_error_function-inl.h:179-297) for general df/dv: the kernel
// repeats forward kinematics for its instance in LDS (cheap next to the rows it writes), evaluates
// every constraint once (one thread each; sigma * df/dv staged in LDS), then one thread per
// (constraint, column) gathers the column's sources -- the ancestor walk turned inside out, as in
// fkJacobianKernel -- and writes FuncDim rows.  Every element of those rows is written.
// =============================================================================================
constexpr int kEvWords = 29; // vp(3) vn(3) sigma*dp(9) sigma*dn(9) tin row nrows|flags tinStop pad ; odd stride: conflict-free

struct SideLds {
  float* js; // [J][kJs]
  double* fkA; // [kFkCh][fkPad(J)] the two buffers of the pointer-jumping FK (mmx_device.hpp fkJumpRoundsD)
  double* fkB;
};

__device__ __forceinline__ SideLds carveSideLds(float* smem, int J, float** next) {
  SideLds s;
  s.js = smem;
  float* p = s.js + ((kJs * J + 3) & ~3);
  s.fkA = reinterpret_cast<double*>(p);
  s.fkB = reinterpret_cast<double*>(p + fkBufFloats(J));
  *next = p + 2 * fkBufFloats(J);
  return s;
}

size_t sideFkLdsFloats(int J) {
  return size_t((kJs * J + 3) & ~3) + 2 * fkBufFloats(J);
}

// forward kinematics of one instance by 256 threads: local transforms of all joints at once,
// pointer-jumping composition, optionally the rotation axes (see fkJacobianKernel)
__device__ __forceinline__ void sideFk(const RigDev& rig, const SideLds& s, const float* th, int tid, bool withAxes) {
  const int Jp = fkPad(rig.J);
  for (int j = tid; j < rig.J; j += 256) {
    float* slot = s.js + kJs * j;
    float loc[8];
    fkLocalSplit(rig, j, th, loc, slot + 8);
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
  fkJumpRoundsD(s.js, s.fkA, s.fkB, rig.J, rig.jumpRounds, tid, 256);
  if (withAxes) {
    for (int j = tid; j < rig.J; j += 256) {
      fkAxesInPlaceP(rig, j, rig.parent[j], s.js);
    }
    __syncthreads();
  }
}

template <bool kWriteJac>
__global__ void __launch_bounds__(256) jointBlocksKernel(
    RigDev rig,
    ProblemDev pb,
    const float* __restrict__ theta,
    float* __restrict__ jac, // or null
    float* __restrict__ res, // or null
    double* __restrict__ err, // or null: err[b] += error of these blocks
    const int32_t* __restrict__ done) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ double red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  selectInstanceRig(rig, b);
  selectInstanceWeights(pb, b);
  if (done != nullptr && done[b] != 0) {
    return;
  }
  const int P = rig.P, G = pb.G;
  float* thL;
  const SideLds sl = carveSideLds(smem, rig.J, &thL);
  float* ev = thL + ((P + 3) & ~3);
  int* evi = reinterpret_cast<int*>(ev);
  for (int i = tid; i < P; i += 256) {
    thL[i] = theta[size_t(b) * P + i];
  }
  __syncthreads();
  sideFk(rig, sl, thL, tid, kWriteJac);
  const float* js = sl.js;
  const size_t M = size_t(pb.M);
  double e = 0.0;
  for (int g = tid; g < G; g += 256) {
    const JointBlockDev k = jointBlockOf(pb, b, pb.genBlock[g]);
    const int i = g - k.first;
    const JointEval o = evalJointConstraint(k, js, pb.genJoint[g], size_t(b) * size_t(k.count) + size_t(i));
    const int row = k.rowStart + o.nrows * i;
    e += double(o.werr);
    if (res != nullptr) {
      float* r = res + size_t(b) * M + row;
      for (int q = 0; q < o.nrows; ++q) {
        r[q] = o.sigma * o.f[q];
      }
    }
    if (kWriteJac) {
      const float sg = fabsf(o.sigma) <= 1e-9f ? 0.f : o.sigma; // early termination (:216): the rows stay zero
      float* w = ev + kEvWords * g;
      w[0] = o.vp.x, w[1] = o.vp.y, w[2] = o.vp.z;
      w[3] = o.vn.x, w[4] = o.vn.y, w[5] = o.vn.z;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        w[6 + q] = sg * o.dp[q];
        w[15 + q] = sg * o.dn[q];
      }
      evi[kEvWords * g + 24] = pb.genTin[g];
      evi[kEvWords * g + 25] = row;
      evi[kEvWords * g + 26] = o.nrows | (o.hasPoint ? 16 : 0) | (o.hasDir ? 32 : 0);
      evi[kEvWords * g + 27] = -1;
    }
  }
  // LimitType::Ellipsoid entries: a point constraint whose target (the projection onto the ellipsoid)
  // is held constant and whose walk stops at ellipsoidParent (limit_error_function.cpp:702-790)
  const float tWeightE = 1e+1f * pb.wLimit;
  for (int q = tid; q < pb.NE; q += 256) {
    const EllipsoidDev ct = pb.ellipsoids[q];
    const int row = pb.rowsJoint - 3 * pb.NE + 3 * q, g = G + q;
    EllipsoidEval o = evalEllipsoid(ct, js, tWeightE);
    if (!(pb.wLimit > 0.f)) { // a block with weight_ <= 0 is skipped; its rows stay zero
      o.jwgt = o.werr = 0.f;
    }
    e += double(o.werr);
    if (res != nullptr) {
      float* r = res + size_t(b) * M + row;
      r[0] = o.diff.x * o.jwgt, r[1] = o.diff.y * o.jwgt, r[2] = o.diff.z * o.jwgt;
    }
    if (kWriteJac) {
      float* w = ev + kEvWords * g;
      w[0] = o.position.x, w[1] = o.position.y, w[2] = o.position.z;
      w[3] = w[4] = w[5] = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        w[6 + k] = (k == 0 || k == 4 || k == 8) ? o.jwgt : 0.f;
        w[15 + k] = 0.f;
      }
      evi[kEvWords * g + 24] = ct.tinParent;
      evi[kEvWords * g + 25] = row;
      evi[kEvWords * g + 26] = 3 | 16;
      evi[kEvWords * g + 27] = ct.tinStop;
    }
  }
  if (err != nullptr) {
    e = waveReduceSum(e);
    if (lane == 0) {
      red[wave] = e;
    }
    __syncthreads();
    if (tid == 0) {
      err[b] += (red[0] + red[1]) + (red[2] + red[3]);
    }
  }
  if (!kWriteJac) {
    return;
  }
  __syncthreads();
  float* jb = jac + size_t(b) * M * size_t(P);
  const int GT = G + pb.NE;
  const int items = GT * P;
  for (int item = tid; item < items; item += 256) {
    const int p = item / GT, g = item - p * GT; // constraints fastest: neighbouring threads write neighbouring rows
    const float* w = ev + kEvWords * g;
    const int tin = evi[kEvWords * g + 24], row = evi[kEvWords * g + 25], fl = evi[kEvWords * g + 26];
    const int tinStop = evi[kEvWords * g + 27];
    const int nrows = fl & 15;
    const bool hasPoint = (fl & 16) != 0, hasDir = (fl & 32) != 0;
    const F3 vp{w[0], w[1], w[2]}, vn{w[3], w[4], w[5]};
    float acc[3] = {0.f, 0.f, 0.f};
    const int e1 = pb.colStart[p + 1];
    for (int ei = pb.colStart[p]; ei < e1; ++ei) {
      const ColumnSourceDev s = pb.colSources[ei];
      if (!((s.tin <= tin) && (tin < s.tout))) {
        continue; // the source's joint is not an ancestor of the constraint's joint
      }
      if (tinStop >= 0 && s.tin <= tinStop && tinStop < s.tout) {
        continue; // ellipsoid limit: the walk stopped before this joint
      }
      const float* a = js + kJs * s.joint;
      F3 gp{0.f, 0.f, 0.f}, gn{0.f, 0.f, 0.f};
      if (s.dof >= 3 && s.dof < 6) { // rotation: axis x (v - t_a) for points, axis x v for directions (:265-278)
        const float* ax = a + 8 + 3 * (s.dof - 3);
        const F3 axis{ax[0], ax[1], ax[2]};
        if (hasPoint) {
          gp = cross(axis, vp - F3{a[0], a[1], a[2]});
        }
        if (hasDir) {
          gn = cross(axis, vn);
        }
      } else if (hasPoint) {
        if (s.dof < 3) { // translation (:248-262): column dof of parent.toLinear(), identity for a root
          if (s.parent < 0) {
            gp = F3{s.dof == 0 ? 1.f : 0.f, s.dof == 1 ? 1.f : 0.f, s.dof == 2 ? 1.f : 0.f};
          } else {
            const float* pp = js + kJs * s.parent;
            const F3 c = qmatCol(Q4{pp[3], pp[4], pp[5], pp[6]}, s.dof);
            gp = F3{c.x * pp[7], c.y * pp[7], c.z * pp[7]};
          }
        } else { // scale (:281-291)
          gp = kLn2 * (vp - F3{a[0], a[1], a[2]});
        }
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float jc = (w[6 + 3 * q] * gp.x + w[7 + 3 * q] * gp.y + w[8 + 3 * q] * gp.z) +
            (w[15 + 3 * q] * gn.x + w[16 + 3 * q] * gn.y + w[17 + 3 * q] * gn.z);
        acc[q] += jc * s.weight;
      }
    }
    float* col = jb + size_t(p) * M + row;
    col[0] = acc[0];
    if (nrows == 3) {
      col[1] = acc[1];
      col[2] = acc[2];
    }
  }
}

size_t jointBlocksLdsBytes(int J, int P, int G) { // G = constraints of the blocks + ellipsoid limits
  return (sideFkLdsFloats(J) + size_t((P + 3) & ~3) + size_t(kEvWords) * size_t(G)) * sizeof(float);
}

// =============================================================================================
// Kernel 2: normal equations from the dense Jacobian.  grid = B, block = 256.
// H = J[:,E]^T J[:,E] (full symmetric n x n), g = J[:,E]^T r, E = enabled parameter list.
// Replaces the column compaction + `H.triangularView<Lower>() += J^T J; JtR += J^T r` of
// GaussNewtonSolverT::computeJtJFromJacobianBlocks (momentum/solver/gauss_newton_solver.cpp:204-216).
// Row chunks of J are staged in LDS ([k][s], padded); each thread accumulates 4x4 tiles of the
// lower triangle in registers over ALL rows (tile loop outside, chunk loop inside would re-stage J,
// so tiles are kept in a small register set and the matrix is swept once per tile batch).
// =============================================================================================
constexpr int kNeChunk = 32; // rows of J staged per step (16 when 32 rows of a very wide system do not fit the LDS)
constexpr int kNeCols = 8; // columns of g per thread: n <= 256 * kNeCols = kMaxSolved (mmx_kernels.hpp)
// a further refinement step is taken while |correction|^2 > kRefineTol2 |step|^2 (at most three steps)
constexpr float kRefineTol2 = 1e-6f;
// ... and a correction is only TAKEN when it is a contraction: |correction|^2 <= kRefineMax2 |step|^2, and not larger than
// the correction before it.  With a well-conditioned factor the first correction is 1e-3 ... 1e-5 of the step; where the
// fp32 factor is no preconditioner any more (pivots at rounding level: rank-deficient J with lambda ~ 1e-7) the
// iteration d += M^-1 (g - A d) diverges -- measured in round 3: cfg2 at lambda = 1e-7 ended at a median error of 74 with
// the unguarded refinement, 6 without any (the double solver: 5e-6 ... 16) -- so such a correction is undone and the
// refinement stops.
constexpr float kRefineMax2 = 0.25f;
constexpr int kNeTilesPerThread = 4; // 4x4 tiles held per thread -> n <= 4*sqrt(2*256*4) ~ 180

__global__ void __launch_bounds__(256) normalEquationsKernel(
    ProblemDev pb,
    int P,
    const float* __restrict__ jac, // [B][M*P] column-major
    const float* __restrict__ res, // [B][M]
    float* __restrict__ jtj, // [B][n*n]
    float* __restrict__ jtr, // [B][n]
    const int32_t* __restrict__ done,
    int chunk) { // rows of J per staged chunk: kNeChunk or half of it (normalEquationsChunkRows)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (done != nullptr && done[b] != 0) {
    return;
  }
  const int n = pb.n, M = pb.M;
  const int nT = (n + 3) >> 2; // tiles per dimension
  const int ld = 4 * nT + 4; // padded row length of the staged chunk (multiple of 4 for b128 reads)
  float* Jc = smem; // [chunk][ld]
  float* rc = smem + chunk * ld; // [chunk]
  const float* Jb = jac + size_t(b) * size_t(M) * size_t(P);
  const float* rb = res + size_t(b) * size_t(M);
  const int numTiles = nT * (nT + 1) / 2;

  for (int base = 0; base < numTiles; base += 256 * kNeTilesPerThread) {
    float acc[kNeTilesPerThread][16];
    int ti[kNeTilesPerThread], tj[kNeTilesPerThread];
#pragma unroll
    for (int q = 0; q < kNeTilesPerThread; ++q) {
      // tile index -> (ti >= tj) of the lower triangle, row-major enumeration
      const int t = base + q * 256 + tid;
      int i = 0;
      if (t < numTiles) {
        i = int((sqrtf(8.f * float(t) + 1.f) - 1.f) * 0.5f);
        while ((i + 1) * (i + 2) / 2 <= t) {
          ++i;
        }
        while (i * (i + 1) / 2 > t) {
          --i;
        }
      }
      ti[q] = i;
      tj[q] = t < numTiles ? t - i * (i + 1) / 2 : 0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[q][e] = 0.f;
      }
    }
    float gacc[kNeCols]; // threads accumulate g[tid], g[tid + 256], ... (first tile batch only)
#pragma unroll
    for (int h = 0; h < kNeCols; ++h) {
      gacc[h] = 0.f;
    }
    for (int k0 = 0; k0 < M; k0 += chunk) {
      const int kc = min(chunk, M - k0);
      __syncthreads();
      // stage rows k0..k0+kc of the compacted Jacobian: element (k, s) = J[k0+k + E[s]*M]
      for (int idx = tid; idx < chunk * 4 * nT; idx += 256) {
        const int k = idx % chunk, s = idx / chunk;
        float v = 0.f;
        if (k < kc && s < n) {
          v = Jb[size_t(pb.enabledList[s]) * M + k0 + k];
        }
        Jc[k * ld + s] = v;
      }
      if (tid < chunk) {
        rc[tid] = tid < kc ? rb[k0 + tid] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kNeTilesPerThread; ++q) {
        if (base + q * 256 + tid < numTiles) {
          const float* pa = Jc + 4 * ti[q];
          const float* pbb = Jc + 4 * tj[q];
          for (int k = 0; k < chunk; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(pa + k * ld);
            const float4 c = *reinterpret_cast<const float4*>(pbb + k * ld);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int x = 0; x < 4; ++x) {
#pragma unroll
              for (int y = 0; y < 4; ++y) {
                acc[q][4 * x + y] += av[x] * cv[y];
              }
            }
          }
        }
      }
      if (base == 0) {
#pragma unroll
        for (int h = 0; h < kNeCols; ++h) {
          const int s = tid + 256 * h;
          if (s < n) {
            for (int k = 0; k < chunk; ++k) {
              gacc[h] += Jc[k * ld + s] * rc[k];
            }
          }
        }
      }
    }
    // write tiles (both triangles)
    float* Hb = jtj + size_t(b) * size_t(n) * size_t(n);
#pragma unroll
    for (int q = 0; q < kNeTilesPerThread; ++q) {
      if (base + q * 256 + tid < numTiles) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
#pragma unroll
          for (int y = 0; y < 4; ++y) {
            const int i = 4 * ti[q] + x, j = 4 * tj[q] + y;
            if (i < n && j < n) {
              Hb[size_t(i) * n + j] = acc[q][4 * x + y];
              Hb[size_t(j) * n + i] = acc[q][4 * x + y];
            }
          }
        }
      }
    }
    if (base == 0) {
#pragma unroll
      for (int h = 0; h < kNeCols; ++h) {
        if (tid + 256 * h < n) {
          jtr[size_t(b) * n + tid + 256 * h] = gacc[h];
        }
      }
    }
  }
}

// =============================================================================================
// Kernel 2b: normal equations from the dense Jacobian on the matrix cores (wide systems:
// BASELINE configs[4], P = 300).  grid = B, block = 256 (one wave per SIMD, one workgroup per CU).
// Same contract as normalEquationsKernel.
//
// The compacted J^T (n x M) is consumed in chunks of 32 rows of J: a chunk (all n columns) is
// staged in LDS column-major with a padded column stride, double-buffered so that the global
// loads of chunk c+1 are in flight while chunk c is multiplied.  H is cut into 16x16 tiles of the
// lower triangle; tile t belongs to wave t & 3 and stays in that wave's accumulator registers for
// the whole sweep over M, so J is read from HBM exactly once (one pass holds 4 * TPW tiles; wider
// systems take several passes).  Per tile and 16 rows: two 16-byte LDS reads and four
// v_mfma_f32_16x16x4_f32 (the k index is permuted identically in both operands, which a
// contraction does not see).
// =============================================================================================
constexpr int kMfKc = 32; // rows of J per chunk
constexpr int kMfLd = 40; // LDS column stride in floats: with 10 16-byte slots per column the four 16-lane groups of a
                           // ds_read_b128 ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md, LDS) hit 16 distinct slots

template <int TPW, bool kVec>
__global__ void __launch_bounds__(256, 1) normalEquationsMfmaKernel(
    ProblemDev pb,
    int P,
    const float* __restrict__ jac, // [B][M*P] column-major
    const float* __restrict__ res, // [B][M]
    float* __restrict__ jtj, // [B][n*n]
    float* __restrict__ jtr, // [B][n]
    const int32_t* __restrict__ done,
    int mirror) { // also write the upper triangle (the solver only reads the lower one)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (done != nullptr && done[b] != 0) {
    return;
  }
  const int n = pb.n, M = pb.M;
  const int NB = (n + 15) >> 4, NP = 16 * NB;
  const int T = NB * (NB + 1) / 2;
  const size_t bufFloats = size_t(NP) * kMfLd;
  float* buf0 = smem;
  float* buf1 = smem + bufFloats;
  float* rc = buf1 + bufFloats; // [2][kMfKc]
  int* tileIJ = reinterpret_cast<int*>(rc + 2 * kMfKc); // [T] I << 16 | J
  const float* Jb = jac + size_t(b) * size_t(M) * size_t(P);
  const float* rb = res + size_t(b) * size_t(M);
  for (int t = tid; t < T; t += 256) {
    int I, Jc;
    tileDecode(t, I, Jc);
    tileIJ[t] = (I << 16) | Jc;
  }
  const int pieces = NP * (kMfKc / 4); // 16-byte pieces of a chunk
  const int numChunks = (M + kMfKc - 1) / kMfKc;
  constexpr int kMaxPieces = 12; // per thread: covers n <= 384 per pass of the staging loop
  // this thread's pieces: the column source offsets are fixed for the whole kernel (no dependent
  // index load inside the chunk loop)
  int srcOff[kMaxPieces]; // float offset of the piece's column inside the instance's J, or -1
#pragma unroll
  for (int q = 0; q < kMaxPieces; ++q) {
    const int e = tid + 256 * q;
    const bool used = e < pieces && (e >> 3) < n;
    const int col = pb.enabledList[min(e >> 3, n - 1)]; // unconditional (clamped index): the twelve loads are independent
    srcOff[q] = used ? col * M : -1;
  }
  float4 stage[kMaxPieces];
  float rstage = 0.f;
  // global -> registers for chunk c (zeros beyond M and for the padding columns)
  auto fetch = [&](int c) {
    const int k0 = c * kMfKc;
#pragma unroll
    for (int q = 0; q < kMaxPieces; ++q) {
      const int e = tid + 256 * q;
      // branch-free: a piece that is not needed reads the first piece of the instance's J instead and
      // is zeroed afterwards, so all loads of a chunk are issued back to back (a branch per piece made
      // the compiler wait for every load before issuing the next one)
      const int kk = k0 + 4 * (e & 7); // 8 pieces per column
      const bool ok = srcOff[q] >= 0 && kk < M;
      const float* src = Jb + (ok ? srcOff[q] + kk : 0);
      float4 v;
      if (kVec) { // M % 4 == 0 and a 16-byte aligned J: the piece never straddles M
        v = *reinterpret_cast<const float4*>(src);
      } else {
        v.x = src[0];
        v.y = src[(ok && kk + 1 < M) ? 1 : 0];
        v.z = src[(ok && kk + 2 < M) ? 2 : 0];
        v.w = src[(ok && kk + 3 < M) ? 3 : 0];
        v.y = kk + 1 < M ? v.y : 0.f;
        v.z = kk + 2 < M ? v.z : 0.f;
        v.w = kk + 3 < M ? v.w : 0.f;
      }
      stage[q] = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
    rstage = (tid < kMfKc && k0 + tid < M) ? rb[k0 + tid] : 0.f;
  };
  auto commit = [&](float* buf, int slot) {
#pragma unroll
    for (int q = 0; q < kMaxPieces; ++q) {
      const int e = tid + 256 * q;
      if (e < pieces) {
        *reinterpret_cast<float4*>(buf + (e >> 3) * kMfLd + 4 * (e & 7)) = stage[q];
      }
    }
    if (tid < kMfKc) {
      rc[slot * kMfKc + tid] = rstage;
    }
  };
  const int col16 = lane & 15, g = lane >> 4;
  float* Hb = jtj + size_t(b) * size_t(n) * size_t(n);
  for (int base = 0; base < T; base += 4 * TPW) { // one pass holds 4 * TPW tiles
    v4f acc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      acc[q] = v4f{0.f, 0.f, 0.f, 0.f};
    }
    float gacc[2] = {0.f, 0.f};
    __syncthreads(); // the previous pass is done with the buffers; tileIJ is visible
    fetch(0);
    commit(buf0, 0);
    __syncthreads();
    for (int c = 0; c < numChunks; ++c) {
      const float* cur = (c & 1) ? buf1 : buf0;
      const float* rcur = rc + (c & 1) * kMfKc;
      if (c + 1 < numChunks) {
        fetch(c + 1); // in flight while this chunk is multiplied
      }
#pragma unroll
      for (int q0 = 0; q0 < TPW; q0 += 4) {
        // four tiles at a time: operands of both 16-row halves, then the MFMAs interleaved across
        // the tiles (independent accumulators back to back).  Slots beyond the last tile repeat
        // it: their accumulators are never stored, and the loop body stays free of branches.
        float4 a[4][2], bb[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = min(base + 4 * (q0 + u) + wave, T - 1);
          const int ij = tileIJ[t]; // same address in every lane: an LDS broadcast
          const float* pa = cur + (16 * (ij >> 16) + col16) * kMfLd + 4 * g;
          const float* pbv = cur + (16 * (ij & 0xffff) + col16) * kMfLd + 4 * g;
          a[u][0] = *reinterpret_cast<const float4*>(pa);
          a[u][1] = *reinterpret_cast<const float4*>(pa + 16);
          bb[u][0] = *reinterpret_cast<const float4*>(pbv);
          bb[u][1] = *reinterpret_cast<const float4*>(pbv + 16);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc[q0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][h].x, bb[u][h].x, acc[q0 + u], 0, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc[q0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][h].y, bb[u][h].y, acc[q0 + u], 0, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc[q0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][h].z, bb[u][h].z, acc[q0 + u], 0, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc[q0 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][h].w, bb[u][h].w, acc[q0 + u], 0, 0, 0);
          }
        }
      }
      if (base == 0) { // g = J^T r rides along on the first pass
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int sIdx = tid + 256 * h;
          if (sIdx < n) {
            const float4* colp = reinterpret_cast<const float4*>(cur + sIdx * kMfLd);
            const float4* rp = reinterpret_cast<const float4*>(rcur);
            float accg = gacc[h];
#pragma unroll
            for (int k4 = 0; k4 < kMfKc / 4; ++k4) {
              accg = dot4(colp[k4], rp[k4], accg);
            }
            gacc[h] = accg;
          }
        }
      }
      if (c + 1 < numChunks) {
        commit((c & 1) ? buf0 : buf1, (c + 1) & 1); // the other buffer: last read two barriers ago
      }
      __syncthreads();
    }
    // write the tiles of this pass (both triangles; C layout: col = lane & 15, row = 4 (lane >> 4) + r)
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int t = base + 4 * q + wave;
      if (t < T) {
        const int ij = tileIJ[t];
        const int I = ij >> 16, Jc = ij & 0xffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + 4 * g + r, j = 16 * Jc + col16;
          if (i < n && j < n) {
            Hb[size_t(i) * n + j] = acc[q][r];
            if (mirror) {
              Hb[size_t(j) * n + i] = acc[q][r];
            }
          }
        }
      }
    }
    if (base == 0) {
#pragma unroll
      for (int h = 0; h < kNeCols; ++h) {
        if (tid + 256 * h < n) {
          jtr[size_t(b) * n + tid + 256 * h] = gacc[h];
        }
      }
    }
  }
}

size_t normalEquationsMfmaLdsBytes(int n) {
  const size_t NB = size_t(n + 15) >> 4, NP = 16 * NB, T = NB * (NB + 1) / 2;
  return (2 * NP * kMfLd + 2 * kMfKc + T) * sizeof(float);
}

// =============================================================================================
// Kernel 3: dense GN step.  grid = B, block = 256, dynamic LDS = n*(n+1) + 3n + M + 4 floats.
//   H diag += lambda ; L L^T = H ; d0 = solve(g)                 (gauss_newton_solver.cpp:248-251)
//   rho = J^T (r - J d0) - lambda d0 ; d = d0 + solve(rho)        (one refinement step, fp32: the
//        corrected seminormal equations -- brings the fp32 step to ~1e-6 of the reference's
//        double-precision solve, DESIGN.md "Numerics")
//   theta[E] -= d                                                 (skeleton_solver_function.cpp:158)
//   convergence / history bookkeeping of SolverT::solve           (solver.cpp:92-119)
// =============================================================================================
// (L L^T) x = b in place for all 256 threads of the workgroup: L in LDS (column-major, leading
// dimension ld, the RECIPROCAL of the diagonal stored on the diagonal), x in LDS.  Blocked by 16:
// the diagonal block is solved by one wave in registers (lane = row, pivots broadcast with
// v_readlane: 16 steps without touching LDS), then every thread owns rows outside the block and
// subtracts the block's 16 columns from them (16 independent LDS reads).  Two barriers per block
// instead of one LDS round trip per unknown.
__device__ __forceinline__ void triangularSolves(const float* A, int ld, int n, float* x, int tid) {
  const int NB = (n + 15) >> 4, lane = tid & 63;
  for (int k = 0; k < NB; ++k) { // forward: L y = b
    const int k0 = 16 * k;
    if (tid < 64) {
      const int i = lane & 15, row = k0 + i;
      const int rowc = min(row, n - 1); // clamped: the reads are unconditional (a branch per read serialises them)
      float a[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float v = A[min(k0 + c, n - 1) * ld + rowc];
        a[c] = (c <= i && row < n) ? v : 0.f;
      }
      const float xr = x[rowc];
      float bi = row < n ? xr : 0.f, invd = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        invd = c == i ? a[c] : invd; // 1 / L(row,row)
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float yc = readLaneF(bi, c) * readLaneF(invd, c);
        bi = i == c ? yc : bi - a[c] * yc; // a[c] = 0 above the diagonal
      }
      if (lane < 16 && row < n) {
        x[row] = bi;
      }
    }
    __syncthreads();
    for (int r = k0 + 16 + tid; r < n; r += 256) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int cc = min(k0 + c, n - 1);
        const float t = A[cc * ld + r] * x[cc];
        acc += (k0 + c < n) ? t : 0.f;
      }
      x[r] -= acc;
    }
    __syncthreads();
  }
  for (int k = NB - 1; k >= 0; --k) { // backward: L^T x = y
    const int k0 = 16 * k;
    if (tid < 64) {
      const int i = lane & 15, row = k0 + i;
      const int rowc = min(row, n - 1);
      float at[16]; // column i of the diagonal block of L^T = row entries L(k0 + c, k0 + i), c >= i
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float v = A[rowc * ld + min(k0 + c, n - 1)];
        at[c] = (c >= i && k0 + c < n && row < n) ? v : 0.f;
      }
      const float xr = x[rowc];
      float bi = row < n ? xr : 0.f, invd = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        invd = c == i ? at[c] : invd;
      }
#pragma unroll
      for (int c = 15; c >= 0; --c) {
        const float xc = readLaneF(bi, c) * readLaneF(invd, c);
        bi = i == c ? xc : bi - at[c] * xc; // at[c] = 0 below the diagonal of L^T
      }
      if (lane < 16 && row < n) {
        x[row] = bi;
      }
    }
    __syncthreads();
    for (int r = tid; r < k0; r += 256) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int cc = min(k0 + c, n - 1);
        const float t = A[r * ld + cc] * x[cc];
        acc += (k0 + c < n) ? t : 0.f;
      }
      x[r] -= acc;
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) choleskyStepKernel(
    ProblemDev pb,
    int P,
    const float* __restrict__ jac, // [B][M*P]
    const float* __restrict__ res, // [B][M]
    const float* __restrict__ jtj, // [B][n*n]
    const float* __restrict__ jtr, // [B][n]
    const double* __restrict__ errIter, // [B] error at the theta used to build J
    float* __restrict__ theta, // [B][P] in/out
    SolveStateDev st,
    StepParams sp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (st.done[b] != 0) {
    return;
  }
  const int n = pb.n, M = pb.M;
  const float lambda = sp.lambdaPer != nullptr ? sp.lambdaPer[b] : sp.lambda;
  const int ld = (n + 1) | 1; // odd stride: the column walks of neighbouring threads hit different LDS banks
  float* A = smem; // [n][ld] column-major: A[j*ld + i] = H(i,j)
  float* g = A + n * ld; // [n]
  float* d0 = g + n; // [n]
  float* rho = d0 + n; // [n]
  float* w = rho + n; // [M]
  int* notPdPtr = reinterpret_cast<int*>(w + M); // all LDS scratch lives in the dynamic region
  if (tid == 0) {
    *notPdPtr = 0;
  }
  long long tclk = clock64();
#define MMX_SCLK(slot)                                   \
  if (sp.clk != nullptr && b == 0) {                     \
    __syncthreads();                                     \
    if (tid == 0) {                                      \
      const long long now = clock64();                   \
      sp.clk[slot] += now - tclk;                        \
      tclk = now;                                        \
    }                                                    \
  }
  const float* Hb = jtj + size_t(b) * n * n;
  // damping of the FACTOR: at least kFactorDamping of the mean diagonal (mmx_device.hpp; every wave sums the trace for
  // itself); the refinement below measures its residual with the caller's lambda
  float lambdaF;
  {
    float tr = 0.f;
    for (int i = tid & 63; i < n; i += 64) {
      tr += Hb[size_t(i) * n + i];
    }
    lambdaF = fmaxf(lambda, kFactorDamping * waveReduceSumF(tr) / float(n > 0 ? n : 1));
    if (tid == 0 && lambdaF > lambda) {
      st.status[b] |= 4; // MMX_SOLVE_DAMPING_FLOORED
    }
  }
  for (int idx = tid; idx < n * n; idx += 256) {
    const int i = idx % n, j = idx / n;
    float v = Hb[idx];
    if (i == j) {
      v += lambdaF;
    }
    A[j * ld + i] = v;
  }
  for (int i = tid; i < n; i += 256) {
    g[i] = jtr[size_t(b) * n + i];
    d0[i] = g[i];
    rho[i] = kPivotFloor * (Hb[size_t(i) * n + i] + lambdaF); // the row's pivot floor (rho is free until the refinement)
  }
  __syncthreads();
  MMX_SCLK(0)
  // Blocked right-looking Cholesky, lower, in place; the diagonal ends up holding 1 / L(k,k) for
  // triangularSolvesT.  Per block of 16 columns, three barriers:
  //   panel   : every lane owns one ROW of the panel in 16 registers -- lanes 0..15 of each wave the
  //             rows of the diagonal block (each wave factors it redundantly), lanes 16..63 forty-eight
  //             rows below it -- and the 16 elimination steps exchange pivots with v_readlane: no
  //             barrier, no LDS traffic inside the panel (the scheme of mmx_fused.hip phase H);
  //   trailing: rank-16 update of the tiles right of the panel on the matrix cores
  //             (v_mfma_f32_16x16x4_f32, operands read straight from the column-major factor).
  {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    auto Aval = [&](int r, int c) -> float { // lower triangle of the padded matrix (identity beyond n)
      const float v = A[min(c, n - 1) * ld + min(r, n - 1)]; // unconditional read, clamped
      return (r < n && c < n) ? v : (r == c ? 1.f : 0.f);
    };
    const int NB = (n + 15) >> 4;
    int notPd = 0;
    for (int kbk = 0; kbk < NB; ++kbk) {
      const int k0 = 16 * kbk, kb = k0 + 16;
      {
        const bool diagLane = lane < 16;
        const int prow = diagLane ? k0 + lane : kb + 48 * wave + (lane - 16);
        float a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          // a diagonal-block row needs its upper part too: element (r, c), c > r, mirrored from (c, r)
          a[c] = (diagLane && c > lane) ? Aval(k0 + c, prow) : Aval(prow, k0 + c);
        }
        const float floorRow = k0 + (lane & 15) < n ? rho[k0 + (lane & 15)] : 0.f;
        __syncthreads(); // every wave has read the diagonal block before wave 0 overwrites it
        float invd = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const float djj = readLaneF(a[jj], jj);
          notPd |= !(djj > 0.f) ? 1 : 0;
          const float inv = djj > readLaneF(floorRow, jj) ? __builtin_amdgcn_rsqf(djj) : 0.f; // see kPivotFloor
          a[jj] *= inv;
          if (lane == jj) {
            invd = inv;
          }
#pragma unroll
          for (int c = jj + 1; c < 16; ++c) {
            a[c] -= a[jj] * readLaneF(a[jj], c);
          }
        }
        if (!diagLane || wave == 0) {
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            if (prow < n && k0 + c < n && (!diagLane || c <= lane)) {
              A[(k0 + c) * ld + prow] = (diagLane && c == lane) ? invd : a[c];
            }
          }
        }
      }
      __syncthreads();
      // trailing tiles (I, Jc), I >= Jc > kbk, dealt round-robin to the four waves
      int t = 0;
      for (int Jc = kbk + 1; Jc < NB; ++Jc) {
        for (int I = Jc; I < NB; ++I, ++t) {
          if ((t & 3) != wave) {
            continue;
          }
          const int ar = 16 * I + (lane & 15), br = 16 * Jc + (lane & 15), p0 = k0 + 4 * (lane >> 4);
          v4f c;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            c[q] = Aval(16 * I + 4 * (lane >> 4) + q, 16 * Jc + (lane & 15));
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // the diagonal of the panel holds reciprocals, but rows ar / br lie below the panel
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(-Aval(ar, p0 + q), Aval(br, p0 + q), c, 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r = 16 * I + 4 * (lane >> 4) + q, cc = 16 * Jc + (lane & 15);
            if (r < n && cc < n && r >= cc) {
              A[cc * ld + r] = c[q];
            }
          }
        }
      }
      __syncthreads();
    }
    if (notPd != 0) { // a raw pivot was not positive (Eigen's LLT would stop with NumericalIssue): reported, the step is taken
      *notPdPtr = 1;
    }
  }
  __syncthreads();
  const bool badPivot = *notPdPtr != 0;
  constexpr bool bad = false; // (since the pivot floor the factorisation always completes; the reference never skips a step either)
  MMX_SCLK(1)
  if (!bad) {
    triangularSolves(A, ld, n, d0, tid);
  }
  __syncthreads();
  MMX_SCLK(2)
  // up to three refinement steps: a further one only while the last correction exceeded 1e-3 of the
  // step (same rule as the fused kernel; one step is the normal case)
  float prevCorr2 = FLT_MAX;
  for (int rf = 0; rf < sp.refine && !bad && n > 0; ++rf) {
    // w = r - J d0   (rows over threads, coalesced down each column)
    const float* Jb = jac + size_t(b) * size_t(M) * size_t(P);
    const float* rb = res + size_t(b) * size_t(M);
    for (int k = tid; k < M; k += 256) {
      float acc = rb[k];
      int s = 0;
      for (; s + 8 <= n; s += 8) { // eight independent loads in flight per trip
        float jv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          jv[u] = Jb[size_t(pb.enabledList[s + u]) * M + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc -= jv[u] * d0[s + u];
        }
      }
      for (; s < n; ++s) {
        acc -= Jb[size_t(pb.enabledList[s]) * M + k] * d0[s];
      }
      w[k] = acc;
    }
    __syncthreads();
    MMX_SCLK(3)
    // rho = J^T w - lambda d0  (one wavefront per column, coalesced, shuffle reduction)
    const int wave = tid >> 6, lane = tid & 63;
    for (int s0 = 4 * wave; s0 < n; s0 += 16) { // four columns per trip: their loads overlap
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      const float* col[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        col[u] = Jb + size_t(pb.enabledList[s0 + u < n ? s0 + u : s0]) * M;
      }
      for (int k = lane; k < M; k += 64) {
        const float wk = w[k];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u] += col[u][k] * wk;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float t = waveReduceSumF(acc[u]);
        if (lane == 0 && s0 + u < n) {
          rho[s0 + u] = t - lambda * d0[s0 + u];
        }
      }
    }
    __syncthreads();
    MMX_SCLK(4)
    triangularSolves(A, ld, n, rho, tid);
    __syncthreads();
    MMX_SCLK(5)
    float c2 = 0.f, d2 = 0.f;
    for (int i = tid; i < n; i += 256) {
      const float cr = rho[i], dn = d0[i] + cr;
      d0[i] = dn;
      c2 += cr * cr;
      d2 += dn * dn;
    }
    c2 = waveReduceSumF(c2);
    d2 = waveReduceSumF(d2);
    __syncthreads();
    if (lane == 0) {
      w[wave] = c2; // w is free between the refinement steps (M >= 8 or the rows pad: see choleskyStepLdsBytes' + 12)
      w[4 + wave] = d2;
    }
    __syncthreads();
    const float corr2 = w[0] + w[1] + w[2] + w[3], step2 = w[4] + w[5] + w[6] + w[7];
    __syncthreads();
    if (corr2 > kRefineMax2 * step2 || corr2 > prevCorr2) { // not a contraction: undo, stop (kRefineMax2)
      for (int i = tid; i < n; i += 256) {
        d0[i] -= rho[i];
      }
      __syncthreads();
      break;
    }
    prevCorr2 = corr2;
    if (!(corr2 > kRefineTol2 * step2)) {
      break;
    }
  }
  // theta -= delta (scatter to the full parameter space, gauss_newton_solver.cpp:254-257)
  if (sp.delta != nullptr) { // line search / damping schedule: stepUpdateKernel applies the step
    for (int s = tid; s < n; s += 256) {
      sp.delta[size_t(b) * n + s] = bad ? 0.f : d0[s];
    }
    if (tid == 0) {
      sp.stepIter[b] = bad ? -(sp.iteration + 1) : sp.iteration + 1;
    }
  } else if (!bad) {
    float* th = theta + size_t(b) * P;
    for (int s = tid; s < n; s += 256) {
      th[pb.enabledList[s]] -= d0[s];
    }
  }
  // SolverT::solve bookkeeping (solver.cpp:92-119)
  if (tid == 0) {
    const double e = errIter[b];
    const double last = st.lastError[b];
    if (st.errorHistory != nullptr) {
      st.errorHistory[size_t(b) * sp.maxIterations + sp.iteration] = e;
    }
    st.iterations[b] = sp.iteration + 1;
    st.finalError[b] = e;
    if (badPivot) {
      st.status[b] |= 2; // MMX_SOLVE_NOT_PD
    }
    const bool converged = fabs(last - e) / (fabs(e) + double(FLT_MIN)) <= double(sp.threshold) * double(FLT_EPSILON);
    if (sp.iteration >= sp.minIterations && converged) {
      st.done[b] = 1;
    }
    st.lastError[b] = e;
  }
}

// =============================================================================================
// Kernel 3c: the large-system GN step (n up to kMaxSolved solved parameters), left-looking: the same step as
// choleskyStepKernel with H and the factor in HBM.  (Round 1's right-looking form, factored in place, re-read
// every trailing tile right after writing it -- ~40 dependent round trips per block column -- and was removed in
// round 3.)  The order of the memory traffic:
//   - H is read once and never written: block column k of the factor is
//       C(I,k) = H(I,k) [+ lambda] - sum_{j<k} L(I,j) L(k,j)^T
//     with ALL operand loads independent of each other (finished tiles only), so they go out in batches
//     of twenty 16-byte requests per lane; no load ever waits for a store (the right-looking form re-read
//     every trailing tile right after writing it: ~40 dependent round trips per block column),
//   - L goes to its own tile-major scratch ([tile (I,j)][16][16] row-major, tile (I,j) at I(I+1)/2 + j):
//     a lane's MFMA operand is one aligned 16-byte load whatever n is, a row block is one contiguous run,
//   - the forward substitution of g rides along with the factorisation (y_k from row block k, which is
//     complete when column k is), so the first solve is the backward sweep only,
//   - both sweeps prefetch the next block's tiles before the 16-step chain of the current one,
//   - the refinement reads J ONCE: `chunkRows` rows at a time through LDS, w = r - J d (double sums) for
//     those rows, then their contribution to rho = J^T w; the next chunk's loads are in flight meanwhile.
// dynamic LDS = max(NP*16, n*(chunkRows+1)) + 4*NP + chunkRows + 2*256 + 8 floats.
// =============================================================================================
constexpr int kJm = 2; // ... of the masked (tile-sparse) products: the lists are short (four: spills, cfg5 -6 %)
constexpr int kJb = 4; // finished block columns per trip of the tile products (4 tiles x kJb + kJb 16-byte loads in flight per lane)
constexpr int kChunkLoads = 10; // 16-byte loads a thread keeps in flight for the next J chunk (n * chunkRows / 4 <= 256 * 10)

struct TiledLds { // the LDS carve of the tiled step's kernels
  float *pan, *g, *d0, *rho, *invDiag, *wch;
  double* part;
  int* flags;
};
__host__ __device__ inline size_t tiledLdsFloats(int n, int chunkRows, TiledLds* out, float* base) {
  const size_t NP = (size_t(n) + 15) & ~size_t(15);
  const size_t chunk = chunkRows > 0 ? size_t(n) * size_t(chunkRows + 1) : 0;
  const size_t panels = chunkRows > 0 ? NP * 16 : 2 * NP * 16; // (chunkRows = 0: the factor stage alone, two panels)
  const size_t panFloats = ((panels > chunk ? panels : chunk) + 3) & ~size_t(3);
  const size_t oG = panFloats, oD = oG, oRho = oD + NP, oInv = oRho + NP, oW = oInv + NP; // (d0 = g: y = L^-1 g is solved in place)
  const size_t oPart = oW + ((size_t(chunkRows) + 3) & ~size_t(3)); // doubles: even float offset
  const size_t oFlags = oPart + (chunkRows > 0 ? 512 : 16);
  if (out != nullptr) {
    out->pan = base, out->g = base + oG, out->d0 = base + oD, out->rho = base + oRho, out->invDiag = base + oInv, out->wch = base + oW;
    out->part = reinterpret_cast<double*>(base + oPart);
    out->flags = reinterpret_cast<int*>(base + oFlags);
  }
  return oFlags + 8;
}

// Left-looking blocked Cholesky of H + lambda I (H: [n][n], lower triangle, read only) into the tile-major L;
// t.g holds g on entry and y = L^-1 g on exit; t.flags[0] = 1 when a pivot was not positive.
__device__ __forceinline__ void tiledFactor(
    const float* __restrict__ H, float* __restrict__ L, int n, float lambda, const TiledLds& t, const StepParams& sp, int b, int tid, long long& tclk) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NP = (n + 15) & ~15, NB = NP >> 4;
  float* pan = t.pan;
  float* g = t.g;
  float* invDiag = t.invDiag;
  int* flags = t.flags;
  const int lrow = lane & 15, lkg = lane >> 4; // operand layout of v_mfma_f32_16x16x4_f32: row, k group
  const int opOff = lrow * 16 + 4 * lkg; // a lane's 16 bytes inside a row-major tile
  for (int k = 0; k < NB; ++k) {
    const int nt = NB - k;
    // (a) the wave's tiles of block column k (I = k + wave, + 4, ...), four per trip
    for (int I0 = k + wave; I0 < NB; I0 += 16) {
      v4f c[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int I = I0 + 4 * t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * I + 4 * lkg + q, cc = 16 * k + lrow;
          const float v = H[size_t(min(r, n - 1)) * n + min(cc, n - 1)]; // clamped: unconditional, independent loads
          c[t][q] = (r < n && cc < n) ? (r > cc ? v : (r == cc ? v + lambda : 0.f)) : (r == cc ? 1.f : 0.f);
        }
      }
      for (int j0 = 0; j0 < k; j0 += kJb) {
        float4 bv[kJb], av[4][kJb];
#pragma unroll
        for (int u = 0; u < kJb; ++u) {
          const int j = min(j0 + u, k - 1);
          bv[u] = *reinterpret_cast<const float4*>(L + size_t(tileIndex(k, j)) * 256 + opOff);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            av[t][u] = *reinterpret_cast<const float4*>(L + size_t(tileIndex(min(I0 + 4 * t, NB - 1), j)) * 256 + opOff);
          }
        }
#pragma unroll
        for (int u = 0; u < kJb; ++u) {
          if (j0 + u < k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[t][u].x, bv[u].x, c[t], 0, 0, 0);
              c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[t][u].y, bv[u].y, c[t], 0, 0, 0);
              c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[t][u].z, bv[u].z, c[t], 0, 0, 0);
              c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[t][u].w, bv[u].w, c[t], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int I = I0 + 4 * t;
        if (I < NB) {
          float* Tl = pan + 256 * (I - k);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Tl[tileAddr(4 * lkg + q, lrow)] = c[t][q];
          }
        }
      }
    }
    // the forward substitution rides along: s_k = g_k - sum_{j<k} L(k,j) y_j (the wave with the fewest tiles)
    if (wave == 3 && k > 0) {
      float acc = 0.f;
      for (int j0 = 0; j0 < k; j0 += 8) {
        float4 lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          lv[u] = *reinterpret_cast<const float4*>(L + size_t(tileIndex(k, min(j0 + u, k - 1))) * 256 + opOff);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (j0 + u < k) {
            acc = dot4(lv[u], *reinterpret_cast<const float4*>(g + 16 * (j0 + u) + 4 * lkg), acc);
          }
        }
      }
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);
      if (lane < 16) {
        g[16 * k + lane] -= acc;
      }
    }
    __syncthreads();
    MMX_SCLK(6)
    // (b) panel factorisation (mmx_fused.hip phase H): lanes 0-15 of every wave the diagonal block,
    // redundantly; lanes 16-63 forty-eight rows below it; pivots by v_readlane
    {
      float* Dk = pan;
      const bool diagLane = lane < 16;
      const int prow = 16 + 48 * wave + (lane - 16);
      const bool active = diagLane || prow < 16 * nt;
      float* Tl = diagLane ? Dk : pan + 256 * ((active ? prow : 0) >> 4);
      const int trow = diagLane ? lane : (prow & 15);
      float a[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = active ? ldsRow4(Tl, trow, q) : float4{0.f, 0.f, 0.f, 0.f};
        a[4 * q] = v.x, a[4 * q + 1] = v.y, a[4 * q + 2] = v.z, a[4 * q + 3] = v.w;
      }
      float bi = g[16 * k + lrow]; // s_k
      const int frow = 16 * k + lrow;
      const float floorRow = frow < n ? kPivotFloor * (H[size_t(frow) * n + frow] + lambda) : 0.f; // see kPivotFloor
      __syncthreads();
      float invd = 0.f;
      bool bad = false;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float djj = readLaneF(a[j], j);
        bad = bad || !(djj > 0.f);
        const float inv = djj > readLaneF(floorRow, j) ? __builtin_amdgcn_rsqf(djj) : 0.f; // see kPivotFloor
        a[j] *= inv;
        if (lane == j) {
          invd = inv;
        }
        panelRowUpdate1(a, j);
      }
      if (diagLane) {
        if (wave == 0) {
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            Dk[tileAddr(lane, c)] = c <= lane ? a[c] : 0.f;
          }
          invDiag[16 * k + lane] = invd;
          if (bad) {
            flags[0] = 1;
          }
        }
      } else if (active) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          Tl[tileAddr(trow, c)] = a[c];
        }
      }
      if (wave == 0) { // L_kk y_k = s_k with the rows still in registers
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float yj = readLaneF(bi, j) * readLaneF(invd, j);
          bi = (lane == j) ? yj : (j < lane ? bi - a[j] * yj : bi);
        }
        if (lane < 16) {
          g[16 * k + lane] = bi;
        }
      }
      __syncthreads();
      for (int pr = 16 + 192 + tid; pr < 16 * nt; pr += 256) { // rows beyond 4 x 48: substitution
        float* Tr = pan + 256 * (pr >> 4);
        float x[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = ldsRow4(Tr, pr & 15, q);
          x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float sum = x[j];
#pragma unroll
          for (int c = 0; c < j; ++c) {
            sum -= x[c] * Dk[tileAddr(j, c)];
          }
          x[j] = sum * invDiag[16 * k + j];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          Tr[tileAddr(pr & 15, c)] = x[c];
        }
      }
      if (16 * nt > 16 + 192) {
        __syncthreads();
      }
    }
    MMX_SCLK(7)
    // (c) every finished tile is written once
    for (int I = k + wave; I < NB; I += 4) {
      *reinterpret_cast<float4*>(L + size_t(tileIndex(I, k)) * 256 + opOff) = ldsRow4(pan + 256 * (I - k), lrow, lkg);
    }
    __syncthreads(); // the panel buffer is reused; the tiles are visible to the workgroup
    MMX_SCLK(1)
  }
}

// Panel factorisation of block column k whose nt tiles sit contiguously in `pan` (LDS; mmx_fused.hip phase H): lanes 0-15 of
// every wave the diagonal block, redundantly; lanes 16-63 forty-eight rows below it; pivots by v_readlane.  Wave 0 also
// finishes y_k (L_kk y_k = s_k with the rows still in registers).  floorRow: the pivot floor of row 16 k + (lane & 15)
// (kPivotFloor x the original diagonal).  Ends with the panel complete and a barrier.  256 threads -- or more: the waves
// beyond the fourth only take part in the barriers (the rows' assignment, and with it every bit of the result, stays).
__device__ __forceinline__ void tiledPanelFactor(float* pan, int nt, int k, float* g, float* invDiag, int* flags, float floorRow, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15;
  float* Dk = pan;
  const bool diagLane = lane < 16;
  const int prow = 16 + 48 * wave + (lane - 16);
  const bool active = wave < 4 && (diagLane || prow < 16 * nt);
  float* Tl = diagLane ? Dk : pan + 256 * ((active ? prow : 0) >> 4);
  const int trow = diagLane ? lane : (prow & 15);
  // a wave whose forty-eight rows all lie beyond the panel only takes part in the barriers: its copy of the chain would
  // compete for the issue slots of its SIMD with the co-resident workgroup's wave (wave-uniform branch)
  const bool waveWorks = wave == 0 || (wave < 4 && 16 + 48 * wave < 16 * nt);
  float a[16] = {};
  float bi = 0.f;
  if (waveWorks) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = active ? ldsRow4(Tl, trow, q) : float4{0.f, 0.f, 0.f, 0.f};
      a[4 * q] = v.x, a[4 * q + 1] = v.y, a[4 * q + 2] = v.z, a[4 * q + 3] = v.w;
    }
    bi = g[16 * k + lrow]; // s_k
  }
  __syncthreads();
  float invd = 0.f;
  bool bad = false;
  if (waveWorks) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float djj = readLaneF(a[j], j);
      bad = bad || !(djj > 0.f);
      const float inv = djj > readLaneF(floorRow, j) ? __builtin_amdgcn_rsqf(djj) : 0.f; // see kPivotFloor
      a[j] *= inv;
      if (lane == j) {
        invd = inv;
      }
      // L_kk y_k = s_k rides along (lanes 0-15: row j's entry is final once column j is scaled; the other lanes carry a dummy)
      const float yj = readLaneF(bi, j) * inv;
      bi = (lane == j) ? yj : (lane > j ? bi - a[j] * yj : bi); // (a row's entries right of the diagonal are scratch)
      panelRowUpdate1(a, j);
    }
  }
  if (waveWorks && diagLane) {
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        Dk[tileAddr(lane, c)] = c <= lane ? a[c] : 0.f;
      }
      invDiag[16 * k + lane] = invd;
      if (bad) {
        flags[0] = 1;
      }
    }
  } else if (waveWorks && active) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      Tl[tileAddr(trow, c)] = a[c];
    }
  }
  if (wave == 0 && lane < 16) {
    g[16 * k + lane] = bi;
  }
  __syncthreads();
  if (16 * nt > 16 + 192) { // rows beyond 4 x 48: substitution against the finished diagonal block
    for (int pr = 16 + 192 + tid; pr < (tid < 256 ? 16 * nt : 0); pr += 256) {
      float* Tr = pan + 256 * (pr >> 4);
      float x[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = ldsRow4(Tr, pr & 15, q);
        x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float sum = x[j];
#pragma unroll
        for (int c = 0; c < j; ++c) {
          sum -= x[c] * Dk[tileAddr(j, c)];
        }
        x[j] = sum * invDiag[16 * k + j];
      }
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        Tr[tileAddr(pr & 15, c)] = x[c];
      }
    }
    __syncthreads();
  }
}

// tiledPanelFactor for a GROUP of waves: several independent block columns (one level of TileMasks::levelSteps) are factored
// side by side, each by its own waves of the workgroup -- `mine`: this wave has a column; wrel: its index inside the
// column's group (0: writes the diagonal block and y_k); lanes 16-63 of wave wrel carry rows 16 + 48 wrel ... of the panel.
// The same arithmetic per column as tiledPanelFactor (the results are bit-identical); every wave of the workgroup takes
// part in the two barriers.  The group's waves must cover the panel (16 + 48 x waves rows).
__device__ __forceinline__ void
tiledPanelFactorGroup(float* pan, int nt, int k, float* g, float* invDiag, int* flags, float floorRow, int lane, int wrel, bool mine) {
  const int lrow = lane & 15;
  float* Dk = pan;
  const bool diagLane = lane < 16;
  const int prow = 16 + 48 * wrel + (lane - 16);
  const bool active = mine && (diagLane || prow < 16 * nt);
  float* Tl = diagLane ? Dk : pan + 256 * ((active ? prow : 0) >> 4);
  const int trow = diagLane ? lane : (prow & 15);
  const bool waveWorks = mine && (wrel == 0 || 16 + 48 * wrel < 16 * nt);
  float a[16] = {};
  float bi = 0.f;
  if (waveWorks) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = active ? ldsRow4(Tl, trow, q) : float4{0.f, 0.f, 0.f, 0.f};
      a[4 * q] = v.x, a[4 * q + 1] = v.y, a[4 * q + 2] = v.z, a[4 * q + 3] = v.w;
    }
    bi = g[16 * k + lrow]; // s_k
  }
  __syncthreads();
  float invd = 0.f;
  bool bad = false;
  if (waveWorks) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float djj = readLaneF(a[j], j);
      bad = bad || !(djj > 0.f);
      const float inv = djj > readLaneF(floorRow, j) ? __builtin_amdgcn_rsqf(djj) : 0.f; // see kPivotFloor
      a[j] *= inv;
      if (lane == j) {
        invd = inv;
      }
      const float yj = readLaneF(bi, j) * inv;
      bi = (lane == j) ? yj : (lane > j ? bi - a[j] * yj : bi);
      panelRowUpdate1(a, j);
    }
  }
  if (waveWorks && diagLane) {
    if (wrel == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        Dk[tileAddr(lane, c)] = c <= lane ? a[c] : 0.f;
      }
      invDiag[16 * k + lane] = invd;
      if (bad) {
        flags[0] = 1;
      }
    }
  } else if (waveWorks && active) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      Tl[tileAddr(trow, c)] = a[c];
    }
  }
  if (mine && wrel == 0 && lane < 16) {
    g[16 * k + lane] = bi;
  }
  __syncthreads();
}

// The 2 x 32 mask words of mmx::TileMasks in the lanes of two registers (lane i, i + 32: block i's word): a v_readlane picks one.
struct TileMaskLanes {
  uint32_t row, col;
};
__device__ __forceinline__ TileMaskLanes loadTileMaskLanes(const uint32_t* __restrict__ masks, int tid) {
  return TileMaskLanes{masks[tid & 31], masks[32 + (tid & 31)]};
}

// The same factorisation two block columns at a time (tile-major H only): the tiles L(I, j < k) are loaded ONCE for
// the columns k and k + 1 -- the finished-tile reads, which are what the factor stage's HBM traffic consists of, halve.
// Column k + 1 lacks its j = k term after that pass; it gets it from the LDS panel of column k once that is factored.
// LDS: two panels (2 NP * 16 floats).
__device__ __forceinline__ void tiledFactorPairs(
    const float* __restrict__ H, float* __restrict__ L, int n, float lambda, const TiledLds& t, const StepParams& sp, int b, int tid, long long& tclk, const TileMaskLanes& ml) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NP = (n + 15) & ~15, NB = NP >> 4;
  float* g = t.g;
  float* invDiag = t.invDiag;
  int* flags = t.flags;
  const int lrow = lane & 15, lkg = lane >> 4;
  const int opOff = lrow * 16 + 4 * lkg;
  auto loadH = [&](int I, int k) { // tile (I, k) of H + lambda I, padded with the identity, strict upper part of a diagonal tile zero
    const float4 hv = *reinterpret_cast<const float4*>(H + size_t(tileIndex(min(max(I, k), NB - 1), min(k, NB - 1))) * 256 + opOff);
    const float hq[4] = {hv.x, hv.y, hv.z, hv.w};
    v4f c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 16 * I + 4 * lkg + q, cc = 16 * k + lrow;
      c[q] = (r < n && cc < n) ? (r > cc ? hq[q] : (r == cc ? hq[q] + lambda : 0.f)) : (r == cc ? 1.f : 0.f);
    }
    return c;
  };
  auto factorPanel = [&](float* pan, int nt, int k) {
    // the row's pivot floor (kPivotFloor) from the original diagonal: tile (k, k) of the tile-major H, [col][row] inside
    const float floorRow = 16 * k + lrow < n ? kPivotFloor * (H[size_t(tileIndex(k, k)) * 256 + lrow * 17] + lambda) : 0.f;
    tiledPanelFactor(pan, nt, k, g, invDiag, flags, floorRow, tid);
  };
  float* pan0 = t.pan;
  float* pan1 = t.pan + size_t(NP) * 16;
  // tile structure (mmx::TileMasks): only the structurally non-zero tiles are computed, kept in the panels (compacted: a
  // panel holds column k's non-zero tiles in row order) and written; all of it wave-uniform integer work
  // (the 2 x 32 mask words live in the lanes of two registers: a v_readlane picks one)
  const uint32_t vRowMask = ml.row, vColMask = ml.col;
  auto rowMask = [&](int I) { return uint32_t(__builtin_amdgcn_readlane(int(vRowMask), I)); };
  auto colMask = [&](int kk) { return uint32_t(__builtin_amdgcn_readlane(int(vColMask), kk)); };
  auto below = [](int i) { return (1u << i) - 1u; }; // bits 0 .. i-1 (i <= 31)
  auto tileL = [&](int I, int j) { return L + size_t(tileIndex(I, j)) * 256 + opOff; };
  for (int k = 0; k < NB; k += 2) {
    const bool two = k + 1 < NB; // (uniform)
    const uint32_t cm0 = colMask(k), cm1 = two ? colMask(k + 1) : 0u; // rows of the two columns (bits >= k / >= k + 1)
    const uint32_t rk0 = rowMask(k) & below(k), rk1 = two ? rowMask(k + 1) & below(k) : 0u; // finished columns j < k they reach
    // (a) the wave's tiles of BOTH columns: every fourth row of the union, one row (two tiles) per trip
    {
      uint32_t rem = cm0 | cm1;
      for (int idx = 0; rem != 0u; ++idx) {
        const int I = __builtin_ctz(rem);
        rem &= rem - 1u;
        if ((idx & 3) != wave) {
          continue;
        }
        const bool in0 = (cm0 >> I & 1u) != 0u, in1 = (cm1 >> I & 1u) != 0u;
        const uint32_t rI = rowMask(I);
        const uint32_t m0 = in0 ? rI & rk0 : 0u, m1 = in1 ? rI & rk1 : 0u;
        v4f c0{0.f, 0.f, 0.f, 0.f}, c1{0.f, 0.f, 0.f, 0.f};
        if (in0) {
          c0 = loadH(I, k);
        }
        if (in1) {
          c1 = loadH(I, k + 1);
        }
        uint32_t mm = m0 | m1;
        while (mm != 0u) {
          int jj[kJm];
          bool use0[kJm], use1[kJm];
#pragma unroll
          for (int u = 0; u < kJm; ++u) {
            const bool have = mm != 0u;
            jj[u] = have ? __builtin_ctz(mm) : 0;
            use0[u] = have && (m0 >> jj[u] & 1u) != 0u;
            use1[u] = have && (m1 >> jj[u] & 1u) != 0u;
            mm = have ? mm & (mm - 1u) : 0u;
          }
          float4 b0[kJm], b1[kJm], av[kJm];
#pragma unroll
          for (int u = 0; u < kJm; ++u) {
            if (use0[u] || use1[u]) {
              av[u] = *reinterpret_cast<const float4*>(tileL(I, jj[u]));
            }
            if (use0[u]) {
              b0[u] = *reinterpret_cast<const float4*>(tileL(k, jj[u]));
            }
            if (use1[u]) {
              b1[u] = *reinterpret_cast<const float4*>(tileL(k + 1, jj[u]));
            }
          }
#pragma unroll
          for (int u = 0; u < kJm; ++u) {
            if (use0[u]) {
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].x, b0[u].x, c0, 0, 0, 0);
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].y, b0[u].y, c0, 0, 0, 0);
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].z, b0[u].z, c0, 0, 0, 0);
              c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].w, b0[u].w, c0, 0, 0, 0);
            }
            if (use1[u]) {
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].x, b1[u].x, c1, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].y, b1[u].y, c1, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].z, b1[u].z, c1, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].w, b1[u].w, c1, 0, 0, 0);
            }
          }
        }
        if (in0) {
          float* T0 = pan0 + 256 * __builtin_popcount(cm0 & below(I));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            T0[tileAddr(4 * lkg + q, lrow)] = c0[q];
          }
        }
        if (in1) {
          float* T1 = pan1 + 256 * __builtin_popcount(cm1 & below(I));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            T1[tileAddr(4 * lkg + q, lrow)] = c1[q];
          }
        }
      }
    }
    // the forward substitution rides along: s = g - sum_{j<k} L(.,j) y_j for the rows of both columns
    if (wave == 3 && (rk0 | rk1) != 0u) {
      float acc0 = 0.f, acc1 = 0.f;
      uint32_t mm = rk0 | rk1;
      while (mm != 0u) {
        int jj[4];
        bool use0[4], use1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool have = mm != 0u;
          jj[u] = have ? __builtin_ctz(mm) : 0;
          use0[u] = have && (rk0 >> jj[u] & 1u) != 0u;
          use1[u] = have && (rk1 >> jj[u] & 1u) != 0u;
          mm = have ? mm & (mm - 1u) : 0u;
        }
        float4 l0[4], l1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (use0[u]) {
            l0[u] = *reinterpret_cast<const float4*>(tileL(k, jj[u]));
          }
          if (use1[u]) {
            l1[u] = *reinterpret_cast<const float4*>(tileL(k + 1, jj[u]));
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (use0[u] || use1[u]) {
            const float4 yv = *reinterpret_cast<const float4*>(g + 16 * jj[u] + 4 * lkg);
            if (use0[u]) {
              acc0 = dot4(l0[u], yv, acc0);
            }
            if (use1[u]) {
              acc1 = dot4(l1[u], yv, acc1);
            }
          }
        }
      }
      acc0 += __shfl_xor(acc0, 16, 64);
      acc0 += __shfl_xor(acc0, 32, 64);
      acc1 += __shfl_xor(acc1, 16, 64);
      acc1 += __shfl_xor(acc1, 32, 64);
      if (lane < 16) {
        g[16 * k + lane] -= acc0;
        if (two) {
          g[16 * (k + 1) + lane] -= acc1;
        }
      }
    }
    __syncthreads();
    MMX_SCLK(6)
    factorPanel(pan0, __builtin_popcount(cm0), k);
    if (two) {
      // the missing term of column k + 1: C(I, k+1) -= L(I,k) L(k+1,k)^T, operands from column k's LDS panel -- when
      // L(k+1,k) is structurally non-zero (it is the second tile of column k's panel then), for the rows both columns hold
      if ((cm0 >> (k + 1) & 1u) != 0u) {
        uint32_t rem = cm0 & cm1;
        for (int idx = 0; rem != 0u; ++idx) {
          const int I = __builtin_ctz(rem);
          rem &= rem - 1u;
          if ((idx & 3) != wave) {
            continue;
          }
          float* T1 = pan1 + 256 * __builtin_popcount(cm1 & below(I));
          v4f c;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            c[q] = T1[tileAddr(4 * lkg + q, lrow)];
          }
          const float4 av = ldsRow4(pan0 + 256 * __builtin_popcount(cm0 & below(I)), lrow, lkg);
          const float4 bv = ldsRow4(pan0 + 256, lrow, lkg);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.x, bv.x, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.y, bv.y, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.z, bv.z, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.w, bv.w, c, 0, 0, 0);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            T1[tileAddr(4 * lkg + q, lrow)] = c[q];
          }
        }
        if (wave == 3 && lane < 16) { // ... and of s_{k+1}: - L(k+1,k) y_k
          float acc = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc = dot4(ldsRow4(pan0 + 256, lane, q), *reinterpret_cast<const float4*>(g + 16 * k + 4 * q), acc);
          }
          g[16 * (k + 1) + lane] -= acc;
        }
        __syncthreads();
      }
      factorPanel(pan1, __builtin_popcount(cm1), k + 1);
    }
    MMX_SCLK(7)
    // (c) every finished tile is written once
    {
      uint32_t rem = cm0;
      for (int idx = 0; rem != 0u; ++idx) {
        const int I = __builtin_ctz(rem);
        rem &= rem - 1u;
        if ((idx & 3) == wave) {
          *reinterpret_cast<float4*>(L + size_t(tileIndex(I, k)) * 256 + opOff) = ldsRow4(pan0 + 256 * idx, lrow, lkg);
        }
      }
      rem = cm1;
      for (int idx = 0; rem != 0u; ++idx) {
        const int I = __builtin_ctz(rem);
        rem &= rem - 1u;
        if ((idx & 3) == wave) {
          *reinterpret_cast<float4*>(L + size_t(tileIndex(I, k + 1)) * 256 + opOff) = ldsRow4(pan1 + 256 * idx, lrow, lkg);
        }
      }
    }
    __syncthreads(); // the panel buffers are reused; the tiles are visible to the workgroup
    MMX_SCLK(1)
  }
}

// Substitutions on the tile-major factor.  forward: L y = x, column by column (after y_k every row below
// subtracts block k's sixteen columns); backward: L^T z = y, row block by row block.  The tiles of the NEXT
// step are requested before the sixteen-step chain of the current one.  x: LDS, NP floats, in place.
// masks: the factor's tile structure (mmx::TileMasks; tiles outside it were never written and are not read), or null: dense
// vMask: the mask words in the lanes of a register (lane i: block i's word; colMask for the forward sweep, rowMask for
// the backward one; TileMaskLanes below), loaded by the caller ahead of time
// kSkip: how a lane whose tile is structurally zero stays away from it -- true: its loads are skipped (divergent; what the
// factor kernel's sweep over the tiles it has just written wants: 1.53 against 1.95 ms for the stage on cfg5), false: it reads
// the diagonal tile instead (in cache) and takes zeros (uniform control flow; what the finish kernel's sweeps over a factor
// coming from HBM want: 0.53 against 0.67 ms)
// (kM: 256-row slabs a thread covers below / left of a block -- 2 up to 512 solved parameters, kNeCols beyond)
template <bool forward, bool kSkip = false, int kM = 2>
__device__ __forceinline__ void tiledSweep(const float* __restrict__ L, int NB, float* x, int tid, uint32_t vMask = 0xffffffffu) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NP = 16 * NB, lrow = lane & 15;
  {
    float dg[16] = {}, dgNext[16] = {}, pv[kM][16] = {}, pvNext[kM][16] = {};
    float di = 1.f, diNext = 1.f; // L(i,i) of the lane's row / column of the diagonal tile
    auto request = [&](int k, float (&dgo)[16], float (&pvo)[kM][16], float& dio) {
      if (k < 0 || k >= NB) {
        return;
      }
      const float* Dt = L + size_t(tileIndex(k, k)) * 256;
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { // forward: row `lrow` of the diagonal tile; backward: its column
          dgo[c] = forward ? Dt[lrow * 16 + c] : Dt[c * 16 + lrow];
        }
        dio = Dt[lrow * 17];
      }
      // column k's tiles below the diagonal (forward) / row k's tiles left of it (backward) that are structurally non-zero.
      // The mask words cover 32 blocks (512 solved parameters: the tree routes); the kM > 2 instantiation runs the
      // explicit-Jacobian route's systems of up to 128 blocks, always dense: no word is looked up there (a shift by a block
      // index >= 32 would be undefined, and so would a v_readlane of lane k >= 64)
      const uint32_t present = kM > 2 ? 0xffffffffu : uint32_t(__builtin_amdgcn_readlane(int(vMask), k));
#pragma unroll
      for (int m = 0; m < kM; ++m) {
        if (forward) {
          const int r = 16 * (k + 1) + tid + 256 * m;
          const int rb = min(r, NP - 1) >> 4;
          const bool have = kM > 2 || (present >> (rb & 31) & 1u) != 0u;
          const float* Tr = L + size_t(tileIndex(kSkip || have ? rb : k, k)) * 256 + (r & 15) * 16;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v{0.f, 0.f, 0.f, 0.f};
            if (!kSkip || have) {
              v = *reinterpret_cast<const float4*>(Tr + 4 * q);
            }
            pvo[m][4 * q] = have ? v.x : 0.f, pvo[m][4 * q + 1] = have ? v.y : 0.f, pvo[m][4 * q + 2] = have ? v.z : 0.f, pvo[m][4 * q + 3] = have ? v.w : 0.f;
          }
        } else {
          const int cidx = min(tid + 256 * m, max(16 * k - 1, 0));
          const bool have = kM > 2 || (present >> ((cidx >> 4) & 31) & 1u) != 0u;
          const float* Tc = L + size_t(tileIndex(k, kSkip || have ? cidx >> 4 : k)) * 256 + (cidx & 15);
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) {
            float v = 0.f;
            if (!kSkip || have) {
              v = Tc[rr * 16];
            }
            pvo[m][rr] = have ? v : 0.f;
          }
        }
      }
    };
    const int kFirst = forward ? 0 : NB - 1, kStep = forward ? 1 : -1;
    request(kFirst, dg, pv, di);
    for (int k = kFirst; k >= 0 && k < NB; k += kStep) {
      request(k + kStep, dgNext, pvNext, diNext);
      if (wave == 0) {
        float bi = x[16 * k + lrow];
        const float invd = di > 0.f ? 1.f / di : 0.f; // (a dropped column has l_jj = 0: zero step, see kPivotFloor)
        if (forward) {
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float yj = readLaneF(bi, j) * readLaneF(invd, j);
            bi = (lrow == j) ? yj : bi - dg[j] * yj; // dg[j] = 0 above the diagonal
          }
        } else {
#pragma unroll
          for (int j = 15; j >= 0; --j) {
            const float xj = readLaneF(bi, j) * readLaneF(invd, j);
            bi = (lrow == j) ? xj : bi - dg[j] * xj;
          }
        }
        if (lane < 16) {
          x[16 * k + lane] = bi;
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < kM; ++m) {
        const int idx = tid + 256 * m;
        const int target = forward ? 16 * (k + 1) + idx : idx;
        if (forward ? target < NP : target < 16 * k) {
          float acc = 0.f;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            acc += pv[m][c] * x[16 * k + c];
          }
          x[target] -= acc;
        }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        dg[c] = dgNext[c];
#pragma unroll
        for (int m = 0; m < kM; ++m) {
          pv[m][c] = pvNext[m][c];
        }
      }
      di = diNext;
    }
  }
}

// theta -= delta (or the step handed to stepUpdateKernel) and SolverT::solve's bookkeeping (solver.cpp:92-119)
__device__ __forceinline__ void applyStepAndBook(
    const ProblemDev& pb, int P, int b, const float* d0, bool badPivot, const double* errIter, float* theta, const SolveStateDev& st, const StepParams& sp, int tid) {
  const int n = pb.n;
  if (sp.delta != nullptr) {
    for (int s2 = tid; s2 < n; s2 += 256) {
      sp.delta[size_t(b) * n + s2] = d0[s2];
    }
    if (tid == 0) {
      sp.stepIter[b] = sp.iteration + 1;
    }
  } else {
    float* th = theta + size_t(b) * P;
    for (int s2 = tid; s2 < n; s2 += 256) {
      th[pb.enabledList[s2]] -= d0[s2];
    }
  }
  if (sp.stepRule == MMX_STEP_TRUST_REGION) { // several linear solves per iteration: trustEndKernel books it once
    if (tid == 0 && badPivot) {
      st.status[b] |= 2;
    }
    return;
  }
  if (tid == 0) { // SolverT::solve bookkeeping (solver.cpp:92-119)
    const double e = errIter[b];
    const double last = st.lastError[b];
    if (st.errorHistory != nullptr) {
      st.errorHistory[size_t(b) * sp.maxIterations + sp.iteration] = e;
    }
    st.iterations[b] = sp.iteration + 1;
    st.finalError[b] = e;
    if (badPivot) { // a raw pivot was not positive; floored (kPivotFloor), the step taken
      st.status[b] |= 2;
    }
    const bool converged = fabs(last - e) / (fabs(e) + double(FLT_MIN)) <= double(sp.threshold) * double(FLT_EPSILON);
    if (sp.iteration >= sp.minIterations && converged) {
      st.done[b] = 1;
    }
    st.lastError[b] = e;
  }
}

template <int kM> // 256-column slabs per thread: 2 (n <= 512, three workgroups per CU) or kNeCols (to kMaxSolved, one)
__global__ void __launch_bounds__(256, kM == 2 ? 3 : 1) choleskyStepTiledKernel(
    ProblemDev pb,
    int P,
    const float* __restrict__ jac, // [B][M*P]
    const float* __restrict__ res, // [B][M]
    const float* __restrict__ jtj, // [B][n*n] H, lower triangle, read only
    const float* __restrict__ jtr, // [B][n]
    float* __restrict__ factor, // [B][NB(NB+1)/2][256] L, tile-major
    const double* __restrict__ errIter,
    float* __restrict__ theta,
    SolveStateDev st,
    StepParams sp,
    int chunkRows) { // 8, 16 or 32
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (st.done[b] != 0) {
    return;
  }
  const int n = pb.n, M = pb.M;
  const float lambda = sp.lambdaPer != nullptr ? sp.lambdaPer[b] : sp.lambda;
  const int NP = (n + 15) & ~15, NB = NP >> 4;
  const int cs = chunkRows + 1; // LDS stride of one J column inside a chunk (odd)
  TiledLds t;
  tiledLdsFloats(n, chunkRows, &t, smem);
  float* pan = t.pan; // block column being factored (swizzled tiles) / J chunk [n][cs]
  float* d0 = t.d0;
  float* rho = t.rho;
  float* wch = t.wch; // [chunkRows] w of the current chunk
  double* part = t.part; // [256] partial sums of w
  float* L = factor + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
  if (tid == 0) {
    t.flags[0] = 0;
  }
  for (int i = tid; i < NP; i += 256) {
    t.g[i] = i < n ? jtr[size_t(b) * n + i] : 0.f;
  }
  __syncthreads();
  long long tclk = clock64();
  float lambdaF; // damping of the FACTOR (kFactorDamping, mmx_device.hpp); the refinement keeps the caller's lambda
  {
    const float* Hb = jtj + size_t(b) * n * n;
    float tr = 0.f;
    for (int i = tid & 63; i < n; i += 64) {
      tr += Hb[size_t(i) * n + i];
    }
    lambdaF = fmaxf(lambda, kFactorDamping * waveReduceSumF(tr) / float(n > 0 ? n : 1));
    if (tid == 0 && lambdaF > lambda) {
      st.status[b] |= 4; // MMX_SOLVE_DAMPING_FLOORED
    }
  }
  tiledFactor(jtj + size_t(b) * n * n, L, n, lambdaF, t, sp, b, tid, tclk);
  const bool badPivot = t.flags[0] != 0;
  constexpr bool bad = false; // (pivot floor: the factorisation always completes, the step is always taken)
  MMX_SCLK(0) // (d0 is t.g: tiledFactor left y = L^-1 g there, behind its last barrier)
  if (!bad) {
    tiledSweep<false, false, kM>(L, NB, d0, tid);
  }
  MMX_SCLK(2)
  float prevCorr2 = FLT_MAX;
  for (int rf = 0; rf < sp.refine && !bad; ++rf) { // see choleskyStepKernel
    const float* Jb = jac + size_t(b) * size_t(M) * size_t(P);
    const float* rb = res + size_t(b) * size_t(M);
    const bool vec4 = (M & 3) == 0 && (reinterpret_cast<uintptr_t>(jac) & 15) == 0;
    const int rowShift = chunkRows == 32 ? 5 : (chunkRows == 16 ? 4 : 3), quadShift = rowShift - 2, quadMask = (1 << quadShift) - 1;
    const int items = n << quadShift; // (column, 16-byte row group) pairs of one chunk
    const int slices = 256 >> rowShift, wRow = tid & (chunkRows - 1), wSlice = tid >> rowShift;
    float4 nx[kChunkLoads];
    float rNext = 0.f;
    // the 16-byte pieces base + tid + 256 i of chunk m0 into nx (base = 0: the part that is prefetched a chunk ahead;
    // systems of more than 256 kChunkLoads pieces per chunk -- beyond 640 solved parameters -- fetch the rest in place)
    auto requestItems = [&](int m0, int base) {
#pragma unroll
      for (int i = 0; i < kChunkLoads; ++i) {
        const int it = base + tid + 256 * i;
        if (it < items) {
          const int col = it >> quadShift, row = m0 + 4 * (it & quadMask);
          const float* src = Jb + size_t(pb.enabledList[col]) * M;
          if (vec4) {
            nx[i] = row < M ? *reinterpret_cast<const float4*>(src + row) : float4{0.f, 0.f, 0.f, 0.f};
          } else {
            nx[i].x = row < M ? src[row] : 0.f;
            nx[i].y = row + 1 < M ? src[row + 1] : 0.f;
            nx[i].z = row + 2 < M ? src[row + 2] : 0.f;
            nx[i].w = row + 3 < M ? src[row + 3] : 0.f;
          }
        }
      }
    };
    auto commitItems = [&](int base) {
#pragma unroll
      for (int i = 0; i < kChunkLoads; ++i) {
        const int it = base + tid + 256 * i;
        if (it < items) {
          float* dst = pan + (it >> quadShift) * cs + 4 * (it & quadMask);
          dst[0] = nx[i].x, dst[1] = nx[i].y, dst[2] = nx[i].z, dst[3] = nx[i].w;
        }
      }
    };
    auto requestChunk = [&](int m0) {
      requestItems(m0, 0);
      if (tid < chunkRows) {
        rNext = m0 + tid < M ? rb[m0 + tid] : 0.f;
      }
    };
    float racc[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) {
      racc[m] = 0.f;
    }
    requestChunk(0);
    for (int m0 = 0; m0 < M; m0 += chunkRows) {
      commitItems(0);
      for (int base = 256 * kChunkLoads; base < items; base += 256 * kChunkLoads) {
        requestItems(m0, base);
        commitItems(base);
      }
      const float rCur = rNext;
      __syncthreads();
      if (m0 + chunkRows < M) {
        requestChunk(m0 + chunkRows);
      }
      // w = r - J d for the chunk's rows: the n-term sums cancel near the solution and bound what the
      // refinement can recover -- double
      {
        double acc = 0.0;
        for (int s2 = wSlice; s2 < n; s2 += slices) {
          acc += double(pan[s2 * cs + wRow]) * double(d0[s2]);
        }
        part[tid] = acc;
      }
      __syncthreads();
      if (tid < chunkRows) {
        double acc = double(rCur);
        for (int s2 = 0; s2 < slices; ++s2) {
          acc -= part[s2 * chunkRows + tid];
        }
        wch[tid] = m0 + tid < M ? float(acc) : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < kM; ++m) {
        const int col = tid + 256 * m;
        if (col < n) {
          float acc = racc[m];
          for (int rr = 0; rr < chunkRows; ++rr) {
            acc += pan[col * cs + rr] * wch[rr];
          }
          racc[m] = acc;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < kM; ++m) {
      const int col = tid + 256 * m;
      if (col < NP) {
        rho[col] = col < n ? racc[m] - lambda * d0[col] : 0.f;
      }
    }
    __syncthreads();
    MMX_SCLK(3)
    tiledSweep<true, false, kM>(L, NB, rho, tid);
    tiledSweep<false, false, kM>(L, NB, rho, tid);
    MMX_SCLK(5)
    float c2 = 0.f, d2 = 0.f;
    for (int i = tid; i < n; i += 256) {
      const float cr = rho[i], dn = d0[i] + cr;
      d0[i] = dn;
      c2 += cr * cr;
      d2 += dn * dn;
    }
    c2 = waveReduceSumF(c2);
    d2 = waveReduceSumF(d2);
    __syncthreads();
    if (lane == 0) {
      wch[wave] = c2; // (the chunk's w is consumed: chunkRows >= 16 floats)
      wch[4 + wave] = d2;
    }
    __syncthreads();
    const float corr2 = wch[0] + wch[1] + wch[2] + wch[3], step2 = wch[4] + wch[5] + wch[6] + wch[7];
    __syncthreads();
    if (corr2 > kRefineMax2 * step2 || corr2 > prevCorr2) { // not a contraction: undo, stop (kRefineMax2)
      for (int i = tid; i < n; i += 256) {
        d0[i] -= rho[i];
      }
      __syncthreads();
      break;
    }
    prevCorr2 = corr2;
    if (!(corr2 > kRefineTol2 * step2)) {
      break;
    }
  }
  applyStepAndBook(pb, P, b, d0, badPivot, errIter, theta, st, sp, tid);
}

// The same step for systems whose refinement goes through the tree (treeRefineKernel, mmx_fused.hip) instead of a
// dense J: stage 1 factors and solves (d0 to `dvec`), stage 2 -- once per refinement round, after treeRefineKernel
// left rho = J^T (r - J d) - lambda d in `rhoVec` -- solves for the correction and, when it was the last one,
// applies the step.  refState[b]: 0 = a refinement round is due, 1 = the iteration's step has been applied.
// (Two block columns per step on the tile-major hand-over of treeNormalEquationsKernel: tiledFactorPairs.)
// (three workgroups per CU: the masked products' integer work does not fit the 128 registers of four -- 56 spilled, the
// stage 13 % slower on cfg5)
__global__ void __launch_bounds__(256, 3) choleskyFactorTiledKernel(
    ProblemDev pb,
    int P,
    const float* __restrict__ jtj,
    const float* __restrict__ jtr,
    float* __restrict__ factor,
    float* __restrict__ dvec, // [B][NP]
    int32_t* __restrict__ refState, // [B]
    const double* __restrict__ errIter,
    float* __restrict__ theta,
    SolveStateDev st,
    StepParams sp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (st.done[b] != 0) {
    if (tid == 0) {
      refState[b] = 1;
    }
    return;
  }
  const int n = pb.n;
  const TileMaskLanes ml = loadTileMaskLanes(sp.tileMasks, tid);
  float lambda = sp.lambdaPer != nullptr ? sp.lambdaPer[b] : sp.lambda;
  const int NP = (n + 15) & ~15, NB = NP >> 4;
  TiledLds t;
  tiledLdsFloats(n, 0, &t, smem);
  float* L = factor + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
  if (tid == 0) {
    t.flags[0] = 0;
  }
  for (int i = tid; i < NP; i += 256) {
    t.g[i] = i < n ? jtr[size_t(b) * n + i] : 0.f;
  }
  {
    // the FACTOR is damped by at least kFactorDamping of the mean diagonal (mmx_device.hpp; as in fusedSolveKernel); the
    // refinement measures its residual with the true damping through J
    float tr = 0.f;
    const float* Hd = jtj + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
    for (int i = tid; i < n; i += 256) {
      tr += Hd[size_t(tileIndex(i >> 4, i >> 4)) * 256 + (i & 15) * 17];
    }
    tr = waveReduceSumF(tr);
    __syncthreads();
    if ((tid & 63) == 0) {
      t.rho[tid >> 6] = tr;
    }
    __syncthreads();
    const float lambdaFloor = kFactorDamping * ((t.rho[0] + t.rho[1]) + (t.rho[2] + t.rho[3])) / float(n > 0 ? n : 1);
    if (tid == 0 && lambdaFloor > lambda) {
      st.status[b] |= 4; // MMX_SOLVE_DAMPING_FLOORED
    }
    lambda = fmaxf(lambda, lambdaFloor);
  }
  __syncthreads();
  long long tclk = clock64();
  tiledFactorPairs(jtj + size_t(b) * size_t(NB * (NB + 1) / 2) * 256, L, n, lambda, t, sp, b, tid, tclk, ml);
  const bool badPivot = t.flags[0] != 0;
  if (sp.diagAcc != nullptr) { // the precision estimate's input (as in choleskyFactorResidentKernel)
    const float* Hd = jtj + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
    float w = 0.f;
    for (int i = tid; i < n; i += 256) {
      const float hd = Hd[size_t(tileIndex(i >> 4, i >> 4)) * 256 + (i & 15) * 17] + lambda;
      // (a column the pivot floor DROPPED leaves 1 / l_jj = 0 here: its pivot ratio is at or below kPivotFloor, i.e. w >= 1 --
      // without this an element whose only small pivots were dropped reported ratio 1 and went unmarked; the one-launch solve
      // counts such columns because it reads the unguarded 1 / l_jj)
      w = fmaxf(w, t.invDiag[i] > 0.f ? kPivotFloor * hd * t.invDiag[i] * t.invDiag[i] : 1.f);
    }
    w = waveReduceMaxF(w);
    if ((tid & 63) == 0) {
      atomicMax(reinterpret_cast<int*>(sp.diagAcc + 4 * size_t(b) + 1), __float_as_int(w));
      atomicMax(reinterpret_cast<int*>(sp.diagAcc + 4 * size_t(b) + 3), __float_as_int(w)); // (this iteration's: the finish stage weights it with the step)
    }
  }
  constexpr bool bad = false; // (pivot floor: the factorisation always completes, the step is always taken)
  float* d0 = t.g; // y = L^-1 g, solved in place
  MMX_SCLK(0)
  if (!bad) {
    tiledSweep<false, true>(L, NB, d0, tid, ml.row);
  }
  MMX_SCLK(2)
  if (bad || !sp.refine) {
    applyStepAndBook(pb, P, b, d0, badPivot, errIter, theta, st, sp, tid);
    if (tid == 0) {
      refState[b] = 1;
    }
    return;
  }
  for (int i = tid; i < NP; i += 256) {
    dvec[size_t(b) * NP + i] = d0[i];
  }
  if (tid == 0) {
    refState[b] = 0;
    if (badPivot) { // (the finish stage books the iteration; the floored pivot is reported from here)
      st.status[b] |= 2;
    }
  }
}

constexpr int kResidentLoads = 20; // 16-byte requests a lane of the resident kernels keeps in flight while a matrix comes in

// Substitutions on a factor whose structurally non-zero tiles are resident in LDS (column-compact slots, swizzled tiles:
// choleskyFactorResidentKernel).  maskWords: the 96 words of StepParams::tileMasks in LDS (per-lane lookups), vRowMask /
// vColMask / vColBase: the same words in the lanes of registers (uniform lookups by v_readlane).  invDiag: 1 / l_jj per
// row (LDS) or null: taken from the diagonal tiles.  x: LDS, in place.  kW waves.
template <int kW = 4>
__device__ __forceinline__ void residentSweepBackward(
    const float* tiles, const uint32_t* maskWords, uint32_t vRowMask, uint32_t vColBase, int NB, float* x, const float* invDiag, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lrow = lane & 15;
  auto below = [](int i) { return (1u << i) - 1u; };
  for (int k = NB - 1; k >= 0; --k) {
    const float* Dk = tiles + 256 * __builtin_amdgcn_readlane(int(vColBase), k);
    if (wave == 0) { // x_k = L_kk^-T x_k: sixteen steps, the lane's column of the diagonal tile in registers
      float dgc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        dgc[c] = Dk[tileAddr(c, lrow)]; // L(16k + c, 16k + lrow): zero above the diagonal
      }
      float bi = x[16 * k + lrow];
      float invd; // 1 / l_jj (0 for a dropped column)
      if (invDiag != nullptr) {
        invd = invDiag[16 * k + lrow];
      } else {
        const float dd = Dk[tileAddr(lrow, lrow)];
        invd = dd > 0.f ? 1.f / dd : 0.f;
      }
#pragma unroll
      for (int j = 15; j >= 0; --j) {
        const float xj = readLaneF(bi, j) * readLaneF(invd, j);
        bi = (lrow == j) ? xj : bi - dgc[j] * xj;
      }
      if (lane < 16) {
        x[16 * k + lane] = bi;
      }
    }
    __syncthreads();
    const uint32_t present = uint32_t(__builtin_amdgcn_readlane(int(vRowMask), k)) & below(k); // row block k's tiles left of the diagonal
    // x[c] -= sum_r L(16k + r, c) x_k[r] for the columns c < 16 k whose tile (k, c >> 4) exists: a thread per column
    if (present != 0u) {
      for (int c = tid; c < 16 * k; c += 64 * kW) {
        const int jb = c >> 4;
        if ((present >> jb & 1u) == 0u) {
          continue;
        }
        const float* T = tiles + 256 * (int(maskWords[64 + jb]) + __builtin_popcount(maskWords[32 + jb] & below(k))); // tile (k, jb)
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc += T[tileAddr(r, c & 15)] * x[16 * k + r];
        }
        x[c] -= acc;
      }
      __syncthreads();
    }
  }
}

// The factor stage of the wide route with the whole (tile-sparse) factor RESIDENT in LDS -- for systems whose structurally
// non-zero tiles (mmx::TileMasks) fit half a CU (<= 75 tiles of 1 KB + the vectors: two workgroups per CU; cfg5 has 71).
// Same contract as choleskyFactorTiledKernel.  H's tiles are read from HBM ONCE, straight into the slots the factor's
// tiles will occupy (column-compact: tile (I, k) at colBase[k] + rank of I in column k, so a block column's panel is
// contiguous and tiledPanelFactor works on it in place); the left-looking update takes its operands L(I,j), L(k,j) from
// LDS (~100 cycles instead of a dependent HBM round trip per column), every finished column is stored to the tile-major
// factor in HBM without anybody waiting for it (the finish stage and the trust region read it there), and the backward
// substitution of the first solve runs on the resident tiles.  The block columns are taken a LEVEL of the elimination tree
// at a time (TileMasks::levelSteps: independent columns side by side, each panel on its own waves).
//   kW waves per workgroup: 4 (8 was measured in round 5 and is 1.1 % slower on BASELINE configs[4]: profiles/r05_exp_fused.txt): the matrix comes in, the left-looking updates are
// dealt, the sweep's columns and the factor's stores are spread over eight waves; the panels' rows and the damping's trace stay
// on the first four (same sums in the same order: bit-identical results).
template <int kW>
__global__ void __launch_bounds__(64 * kW, kW == 8 ? 4 : 2) choleskyFactorResidentKernel(
    ProblemDev pb,
    int P,
    const float* __restrict__ jtj,
    const float* __restrict__ jtr,
    float* __restrict__ factor,
    float* __restrict__ dvec, // [B][NP]
    int32_t* __restrict__ refState, // [B]
    const double* __restrict__ errIter,
    float* __restrict__ theta,
    SolveStateDev st,
    StepParams sp,
    int numTiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (st.done[b] != 0) {
    if (tid == 0) {
      refState[b] = 1;
    }
    return;
  }
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 15, lkg = lane >> 4, opOff = lrow * 16 + 4 * lkg;
  const int n = pb.n, NP = (n + 15) & ~15, NB = NP >> 4;
  // mask words and the columns' first slots in the lanes of three registers (lane i: block i's word)
  const uint32_t vRowMask = sp.tileMasks[lane & 31], vColMask = sp.tileMasks[32 + (lane & 31)], vColBase = sp.tileMasks[64 + (lane & 31)];
  auto rowMask = [&](int I) { return uint32_t(__builtin_amdgcn_readlane(int(vRowMask), I)); };
  auto colMask = [&](int kk) { return uint32_t(__builtin_amdgcn_readlane(int(vColMask), kk)); };
  auto colBase = [&](int kk) { return __builtin_amdgcn_readlane(int(vColBase), kk); };
  auto below = [](int i) { return (1u << i) - 1u; }; // bits 0 .. i-1 (i <= 31)
  float* tiles = smem; // [numTiles][256], swizzled (tileAddr)
  float* g = tiles + size_t(numTiles) * 256; // [NP] g, then y, then the step
  float* invDiag = g + NP; // [NP]
  float* floorAll = invDiag + NP; // [NP] the rows' pivot floors: kPivotFloor x the original diagonal (incl. the damping)
  float* red = floorAll + NP; // [4]
  int* flags = reinterpret_cast<int*>(red + 4); // [4]
  uint32_t* maskWords = reinterpret_cast<uint32_t*>(flags + 4); // [96] the same words for per-lane lookups (a v_readlane needs a uniform index)
  auto tileAt = [&](int I, int j) { return tiles + 256 * (colBase(j) + __builtin_popcount(colMask(j) & below(I))); }; // (uniform)
  const float* H = jtj + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
  float* L = factor + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
  float lambda = sp.lambdaPer != nullptr ? sp.lambdaPer[b] : sp.lambda;
  if (tid == 0) {
    flags[0] = 0;
  }
  if (tid < 96) {
    maskWords[tid] = sp.tileMasks[tid];
  }
  for (int i = tid; i < NP; i += 64 * kW) {
    g[i] = i < n ? jtr[size_t(b) * n + i] : 0.f;
  }
  long long tclk = clock64();
  // ---- H into the slots: the wave's tiles (every fourth slot), ALL its requests in flight together -- one HBM round trip
  // for the whole matrix (4 x kResidentLoads >= the tiles that fit the kernel's LDS budget)
  constexpr int kLoads = 4 * kResidentLoads / kW;
  for (int s0 = 0; s0 < numTiles; s0 += kW * kLoads) {
    float4 hv[kLoads];
    // the slots' (I, k) codes of this wave's tiles: ONE load, lane u the code of tile u (read one by one they were twenty
    // dependent global round trips in front of the matrix's own: each a vector load the wave waited for before it could form
    // the next address -- a tenth of the kernel, profiles/r06_exp_fused.txt)
    const uint32_t vCode = sp.tileMasks[96 + min(s0 + kW * min(lane, kLoads - 1) + wave, numTiles - 1)];
#pragma unroll
    for (int u = 0; u < kLoads; ++u) {
      const int code = __builtin_amdgcn_readlane(int(vCode), u); // I | k << 8 (clamped to the last slot: unconditional, independent requests)
      hv[u] = *reinterpret_cast<const float4*>(H + size_t(tileIndex(code & 0xff, code >> 8)) * 256 + opOff);
    }
#pragma unroll
    for (int u = 0; u < kLoads; ++u) {
      const int sl = s0 + kW * u + wave;
      if (sl < numTiles) {
        const int code = __builtin_amdgcn_readlane(int(vCode), u);
        const int tI = code & 0xff, tK = code >> 8;
        const float hq[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w};
        float* T = tiles + 256 * sl;
#pragma unroll
        for (int q = 0; q < 4; ++q) { // padded with the identity, strict upper part of a diagonal tile zero (tiledFactorPairs::loadH)
          const int r = 16 * tI + 4 * lkg + q, cc = 16 * tK + lrow;
          T[tileAddr(4 * lkg + q, lrow)] = (r < n && cc < n) ? (r >= cc ? hq[q] : 0.f) : (r == cc ? 1.f : 0.f);
        }
      }
    }
  }
  __syncthreads();
  MMX_SCLK(3)
  { // damping of the FACTOR: at least kFactorDamping of the mean diagonal (mmx_device.hpp), from the resident diagonal
    // tiles; then the diagonal gets it and every row its pivot floor
    float tr = 0.f;
    for (int i = tid; i < (tid < 256 ? n : 0); i += 256) { // (the first four waves: the sum keeps its order)
      tr += tiles[256 * int(maskWords[64 + (i >> 4)]) + tileAddr(i & 15, i & 15)];
    }
    tr = waveReduceSumF(tr);
    if (lane == 0 && wave < 4) {
      red[wave] = tr;
    }
    __syncthreads();
    const float lambdaFloor = kFactorDamping * ((red[0] + red[1]) + (red[2] + red[3])) / float(n > 0 ? n : 1);
    if (tid == 0 && lambdaFloor > lambda) {
      st.status[b] |= 4; // MMX_SOLVE_DAMPING_FLOORED
    }
    lambda = fmaxf(lambda, lambdaFloor);
    for (int i = tid; i < NP; i += 64 * kW) {
      float* dg = tiles + 256 * int(maskWords[64 + (i >> 4)]) + tileAddr(i & 15, i & 15);
      const float hd = i < n ? *dg + lambda : 1.f;
      *dg = hd;
      floorAll[i] = i < n ? kPivotFloor * hd : 0.f;
    }
  }
  __syncthreads();
  MMX_SCLK(6)
  // ---- left-looking factorisation in LDS, one LEVEL of independent block columns per step (TileMasks::levelSteps behind
  // the slot list: the columns of a step have no tile in each other's rows, e.g. the finger chains of the two hands
  // next to the spine's; cfg5: 17 columns in 11 steps).  Per step: the left-looking updates of all its columns (tiles dealt
  // to the waves round robin), one barrier, then the panels side by side, each on its own waves.
  const uint32_t* sched = sp.tileMasks + 96 + numTiles;
  const int numSteps = int(sched[0]);
  // the schedule's entries (four per step, at most 32 steps) in the lanes of two registers: one round trip for all of them
  // instead of one per step in front of its updates
  const uint32_t vSched0 = lane < 4 * numSteps ? sched[1 + lane] : 0u, vSched1 = 64 + lane < 4 * numSteps ? sched[1 + 64 + lane] : 0u;
  for (int st = 0; st < numSteps; ++st) {
    int ent[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = 4 * st + e; // (uniform)
      ent[e] = idx < 64 ? __builtin_amdgcn_readlane(int(vSched0), idx) : __builtin_amdgcn_readlane(int(vSched1), idx - 64);
    }
    bool anyUpdate = false;
    int rr = 0; // round robin over the step's (column, tile) pairs
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (ent[e] < 0) {
        continue;
      }
      const int k = ent[e] & 0xff;
      const uint32_t cm = colMask(k), rk = rowMask(k) & below(k);
      float* pan = tiles + 256 * colBase(k);
      if (rk == 0u) {
        continue;
      }
      anyUpdate = true;
      uint32_t rem = cm;
      for (int idx = 0; rem != 0u; ++idx) {
        const int I = __builtin_ctz(rem);
        rem &= rem - 1u;
        uint32_t m = rowMask(I) & rk;
        const bool take = (rr & (kW - 1)) == wave;
        ++rr;
        if (!take || m == 0u) {
          continue;
        }
        float* Tc = pan + 256 * idx;
        v4f c0, c1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          c0[q] = Tc[tileAddr(4 * lkg + q, lrow)];
        }
        while (m != 0u) { // four products per trip: their eight operand reads are in flight together
          float4 av[4], bv[4];
          bool use[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            use[u] = m != 0u;
            const int j = use[u] ? __builtin_ctz(m) : 0;
            m = use[u] ? m & (m - 1u) : 0u;
            if (use[u]) {
              av[u] = ldsRow4(tileAt(I, j), lrow, lkg);
              bv[u] = ldsRow4(tileAt(k, j), lrow, lkg);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (use[u]) {
              if (u & 1) { // (two accumulators: consecutive products do not wait for each other)
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].x, bv[u].x, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].y, bv[u].y, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].z, bv[u].z, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].w, bv[u].w, c1, 0, 0, 0);
              } else {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].x, bv[u].x, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].y, bv[u].y, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].z, bv[u].z, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(-av[u].w, bv[u].w, c0, 0, 0, 0);
              }
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          Tc[tileAddr(4 * lkg + q, lrow)] = c0[q] + c1[q];
        }
      }
      // the forward substitution rides along: s_k = g_k - sum_{j<k} L(k,j) y_j (the last wave)
      if (wave == kW - 1) {
        float acc = 0.f;
        uint32_t m = rk;
        while (m != 0u) { // four blocks per trip, reads in flight together
          float4 lv[4], yv[4];
          bool use[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            use[u] = m != 0u;
            const int j = use[u] ? __builtin_ctz(m) : 0;
            m = use[u] ? m & (m - 1u) : 0u;
            if (use[u]) {
              lv[u] = ldsRow4(tileAt(k, j), lrow, lkg);
              yv[u] = *reinterpret_cast<const float4*>(g + 16 * j + 4 * lkg);
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (use[u]) {
              acc = dot4(lv[u], yv[u], acc);
            }
          }
        }
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        if (lane < 16) {
          g[16 * k + lane] -= acc;
        }
      }
    }
    if (anyUpdate) {
      __syncthreads();
    }
    MMX_SCLK(7)
    if (((ent[0] >> 12) & 0xf) == 0xf) { // a panel beyond 208 rows: the whole workgroup, with the tail substitution
      const int k = ent[0] & 0xff;
      tiledPanelFactor(tiles + 256 * colBase(k), __builtin_popcount(colMask(k)), k, g, invDiag, flags, floorAll[16 * k + lrow], tid);
    } else {
      int myK = 0, myRel = 0;
      bool mine = false;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ent[e] >= 0) {
          const int w0 = (ent[e] >> 8) & 0xf, nw = (ent[e] >> 12) & 0xf;
          if (wave >= w0 && wave < w0 + nw) {
            mine = true, myK = ent[e] & 0xff, myRel = wave - w0;
          }
        }
      }
      tiledPanelFactorGroup(tiles + 256 * colBase(myK), __builtin_popcount(colMask(myK)), myK, g, invDiag, flags, floorAll[16 * myK + lrow], lane, myRel, mine);
    }
    MMX_SCLK(1)
  }
  const bool badPivot = flags[0] != 0;
  if (sp.diagAcc != nullptr) { // the precision estimate's input: the largest kPivotFloor (H_jj + mu) / d_jj of this factorisation
    float w = 0.f;
    for (int i = tid; i < n; i += 64 * kW) {
      w = fmaxf(w, invDiag[i] > 0.f ? floorAll[i] * invDiag[i] * invDiag[i] : 1.f); // (a dropped column -- 1 / l_jj = 0 -- counts as a pivot ratio at the floor)
    }
    w = waveReduceMaxF(w);
    if (lane == 0) {
      atomicMax(reinterpret_cast<int*>(sp.diagAcc + 4 * size_t(b) + 1), __float_as_int(w)); // (non-negative floats order like their bits)
      atomicMax(reinterpret_cast<int*>(sp.diagAcc + 4 * size_t(b) + 3), __float_as_int(w)); // (this iteration's: the finish stage weights it with the step)
    }
  }
  float* d0 = g; // y = L^-1 g; solved in place: L^T d = y on the resident tiles
  MMX_SCLK(0)
  residentSweepBackward<kW>(tiles, maskWords, vRowMask, vColBase, NB, d0, invDiag, tid);
  MMX_SCLK(2)
  // the factor goes to its tile-major home in HBM (the finish stage and the trust region read it there) -- only now: a
  // __syncthreads() waits for every outstanding global store of the wave, so stores issued per finished column would put
  // an HBM round trip into each of the barriers above (measured: 4.6 k cycles per panel instead of 2.5 k)
  for (int k = 0; k < NB; ++k) {
    uint32_t rem = colMask(k);
    const float* pan = tiles + 256 * colBase(k);
    for (int idx = 0; rem != 0u; ++idx) {
      const int I = __builtin_ctz(rem);
      rem &= rem - 1u;
      if ((idx & (kW - 1)) == wave) {
        *reinterpret_cast<float4*>(L + size_t(tileIndex(I, k)) * 256 + opOff) = ldsRow4(pan + 256 * idx, lrow, lkg);
      }
    }
  }
  if (!sp.refine) {
    if (tid < 256) {
      applyStepAndBook(pb, P, b, d0, badPivot, errIter, theta, st, sp, tid);
    }
    if (tid == 0) {
      refState[b] = 1;
    }
    return;
  }
  for (int i = tid; i < NP; i += 64 * kW) {
    dvec[size_t(b) * NP + i] = d0[i];
  }
  if (tid == 0) {
    refState[b] = 0;
    if (badPivot) { // (the finish stage books the iteration; the floored pivot is reported from here)
      st.status[b] |= 2;
    }
  }
}

__global__ void __launch_bounds__(256, 3) choleskyFinishTiledKernel(
    ProblemDev pb,
    int P,
    const float* __restrict__ factor,
    float* __restrict__ dvec, // [B][NP] in: d ; out: d + correction when another round follows
    const float* __restrict__ rhoVec, // [B][NP]
    int32_t* __restrict__ refState,
    const double* __restrict__ errIter,
    float* __restrict__ theta,
    SolveStateDev st,
    StepParams sp,
    int round) { // 0 .. sp.refine - 1: the last round applies the step whatever the correction was
  __shared__ __attribute__((aligned(16))) float d0[512 + 16];
  __shared__ __attribute__((aligned(16))) float rho[512 + 16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (refState[b] != 0) {
    return;
  }
  const int n = pb.n;
  const TileMaskLanes ml = loadTileMaskLanes(sp.tileMasks, tid);
  const int NP = (n + 15) & ~15, NB = NP >> 4;
  const float* L = factor + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
  for (int i = tid; i < NP; i += 256) {
    d0[i] = dvec[size_t(b) * NP + i];
    rho[i] = rhoVec[size_t(b) * NP + i];
  }
  __syncthreads();
  long long tclk = clock64();
  tiledSweep<true>(L, NB, rho, tid, ml.col);
  tiledSweep<false>(L, NB, rho, tid, ml.row);
  MMX_SCLK(5)
  float c2 = 0.f, d2 = 0.f;
  for (int i = tid; i < n; i += 256) {
    const float cr = rho[i], dn = d0[i] + cr;
    d0[i] = dn;
    c2 += cr * cr;
    d2 += dn * dn;
  }
  c2 = waveReduceSumF(c2);
  d2 = waveReduceSumF(d2);
  __shared__ float sums[8];
  if (lane == 0) {
    sums[wave] = c2;
    sums[4 + wave] = d2;
  }
  __syncthreads();
  const float corr2 = sums[0] + sums[1] + sums[2] + sums[3], step2 = sums[4] + sums[5] + sums[6] + sums[7];
  bool again = corr2 > kRefineTol2 * step2;
  if (corr2 > kRefineMax2 * step2) { // not a contraction (kRefineMax2): undo this correction, the step is the one before it
    for (int i = tid; i < n; i += 256) {
      d0[i] -= rho[i];
    }
    __syncthreads();
    again = false;
  }
  if (again && round + 1 < sp.refine) {
    for (int i = tid; i < NP; i += 256) {
      dvec[size_t(b) * NP + i] = d0[i];
    }
    return;
  }
  if (sp.diagAcc != nullptr && tid == 0) {
    float* a = sp.diagAcc + 4 * size_t(b);
    if (step2 > 0.f) { // the largest refinement ratio of the solve (squared)
      a[2] = fmaxf(a[2], corr2 / step2);
    }
    // the precision estimate, iteration by iteration (fusedSolveKernel has the rationale): this iteration's largest
    // kPivotFloor (H_jj + mu) / d_jj weighted with the residual's share sqrt(e_it / e_0)
    const float eNow = float(errIter[b]);
    if (sp.iteration == 0) {
      sp.diagErr0[b] = eNow;
    }
    const float e0 = sp.diagErr0[b];
    a[0] = fmaxf(a[0], a[3] * (e0 > 0.f ? sqrtf(eNow / e0) : 1.f));
    a[3] = 0.f;
  }
  applyStepAndBook(pb, P, b, d0, false, errIter, theta, st, sp, tid);
  if (tid == 0) {
    refState[b] = 1;
  }
}

// =============================================================================================
// Kernel 4: the parameter update of an iteration when it needs trial evaluations of the error --
// GaussNewtonSolverT::updateParameters with doLineSearch (momentum/solver/gauss_newton_solver.cpp:
// 283-313: Armijo backtracking, c1 = 1e-3, tau = 0.5, <= 10 trial steps, the last trial stays) or the
// LM gain-ratio schedule (the lambda form of TrustRegionQRT's radius rule, trust_region_qr.cpp:244-268;
// same arithmetic as fusedSolveKernel phase K and the oracle's stepRule 1).  grid = B, block = 256.
// A trial = SkeletonSolverFunctionT::getError (skeleton_solver_function.cpp:64-83): forward
// kinematics without derivatives + the error of every block, rounded through float (:82).
// =============================================================================================
__global__ void __launch_bounds__(256) stepUpdateKernel(
    RigDev rig,
    ProblemDev pb,
    float* __restrict__ theta, // [B][P] in/out
    const float* __restrict__ jtr, // [B][n] (LM: predicted decrease)
    const double* __restrict__ errIter, // [B] error at the theta the step was computed at
    StepParams sp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ double red[4];
  __shared__ float redF[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  selectInstanceRig(rig, b);
  selectInstanceWeights(pb, b);
  if (sp.stepRule == MMX_STEP_TRUST_REGION) { // only the instances whose step is the one to try (phase 2)
    const int ph = sp.tr.phase[b];
    if (ph != 2) {
      if (tid == 0 && ph != 3) {
        atomicAdd(sp.tr.active, 1);
      }
      return;
    }
  } else {
    const int mark = sp.stepIter[b];
    if (mark != sp.iteration + 1 && mark != -(sp.iteration + 1)) {
      return; // the instance had converged before this iteration
    }
    if (mark < 0) { // H was not positive definite: no step; the schedule raises the damping
      if (sp.stepRule == MMX_STEP_LM_SCHEDULE && tid == 0) {
        const float lam = sp.lambdaPer != nullptr ? sp.lambdaPer[b] : sp.lambda;
        if (sp.stepHistory != nullptr) {
          double* sh = sp.stepHistory + (size_t(b) * sp.maxIterations + sp.iteration) * 2;
          sh[0] = double(lam);
          sh[1] = -1.0; // no step was taken: the schedule treats it as a rejected one
        }
        sp.lambdaPer[b] = fminf(lam * sp.lmUp, sp.lmLambdaMax);
      }
      return;
    }
  }
  float lambda = sp.lambdaPer != nullptr ? sp.lambdaPer[b] : sp.lambda;
  const int P = rig.P, n = pb.n;
  float* thL;
  const SideLds sl = carveSideLds(smem, rig.J, &thL);
  float* thT = thL + ((P + 3) & ~3);
  float* th = theta + size_t(b) * P;
  const float* dl = sp.delta + size_t(b) * n;
  for (int i = tid; i < P; i += 256) {
    thL[i] = th[i];
  }
  const double curError = errIter[b];
  // every thread returns the error at thT
  auto trialError = [&]() -> double {
    sideFk(rig, sl, thT, tid, false);
    double e = 0.0;
    for (int u = tid; u < pb.U; u += 256) {
      e += double(evalUnit(pb, sl.js, b, u).werr);
    }
    for (int g = tid; g < pb.G; g += 256) {
      const JointBlockDev k = jointBlockOf(pb, b, pb.genBlock[g]);
      e += double(evalJointConstraint(k, sl.js, pb.genJoint[g], size_t(b) * size_t(k.count) + size_t(g - k.first)).werr);
    }
    if (pb.wLimit > 0.f) {
      for (int q = tid; q < pb.NE; q += 256) {
        e += double(evalEllipsoid(pb.ellipsoids[q], sl.js, 1e+1f * pb.wLimit).werr);
      }
    }
    if (pb.M > pb.rowsJoint) {
      e += paramRowsError<false>(rig, pb, P, thT, b, tid);
    }
    e = waveReduceSum(e);
    __syncthreads();
    if (lane == 0) {
      red[wave] = e;
    }
    __syncthreads();
    return double(float((red[0] + red[1]) + (red[2] + red[3])));
  };
  auto makeTrial = [&](float scale) {
    __syncthreads();
    for (int i = tid; i < P; i += 256) {
      thT[i] = thL[i];
    }
    __syncthreads();
    for (int c = tid; c < n; c += 256) {
      thT[pb.enabledList[c]] -= scale * dl[c];
    }
    __syncthreads();
  };
  if (sp.stepRule == MMX_STEP_TRUST_REGION) {
    // TrustRegionQRT::doIteration, the trial (trust_region_qr.cpp:240-269; fusedSolveKernel phase K): gain ratio against
    // the quadratic model e - 2 g.p + p^T (J^T J + 1e-20 I) p, where p^T J^T J p = g.p - mu |p|^2 because
    // (J^T J + mu I) p = g; radius x 0.25 / x 2 (cap 10); a step with rho <= 0 is rejected.
    float p0 = 0.f, p1 = 0.f;
    for (int c = tid; c < n; c += 256) {
      p0 += dl[c] * dl[c];
      p1 += dl[c] * jtr[size_t(b) * n + c];
    }
    p0 = waveReduceSumF(p0);
    p1 = waveReduceSumF(p1);
    __shared__ float redT[8];
    if (lane == 0) {
      redT[wave] = p0;
      redT[4 + wave] = p1;
    }
    __syncthreads();
    const float dn2 = (redT[0] + redT[1]) + (redT[2] + redT[3]), dg = (redT[4] + redT[5]) + (redT[6] + redT[7]);
    makeTrial(1.f);
    const double eNew = trialError();
    const float predicted = dg + (lambda - 1e-20f) * dn2; // e - model (lambdaPer holds mu = 1e-20 + (lambda_TR - 1e-10))
    const float rho = float((curError - eNew) / double(predicted));
    float radius = sp.tr.radius[b];
    if (rho < 0.25f) { // :256-262
      radius = 0.25f * radius;
    } else if (rho > 0.75f) {
      radius = fminf(2.f * radius, 10.f);
    }
    if (rho > 0.f) { // :265
      for (int i = tid; i < P; i += 256) {
        th[i] = thT[i];
      }
    }
    if (tid == 0) {
      sp.tr.radius[b] = radius;
      if (rho > 0.f) {
        sp.tr.phase[b] = 3;
      } else {
        const int tried = sp.tr.step[b] + 1;
        sp.tr.step[b] = tried;
        if (tried >= 10) { // :157: every trial rejected, the parameters stay (:268-269)
          sp.tr.phase[b] = 3;
        } else { // the same step against the smaller radius: decide again (a Newton update of lambda follows)
          sp.tr.phase[b] = 1;
          sp.tr.newton[b] = 0;
          atomicAdd(sp.tr.active, 1);
        }
      }
    }
    return;
  }
  if (sp.stepRule == MMX_STEP_LM_SCHEDULE) {
    float part = 0.f;
    for (int c = tid; c < n; c += 256) {
      part += dl[c] * jtr[size_t(b) * n + c] + lambda * dl[c] * dl[c];
    }
    part = waveReduceSumF(part);
    if (lane == 0) {
      redF[wave] = part;
    }
    __syncthreads();
    const float predicted = float((double(redF[0]) + double(redF[1])) + (double(redF[2]) + double(redF[3]))); // d.g + lambda d.d
    makeTrial(1.f);
    const double eNew = trialError();
    const float rho = predicted > 0.f ? float((curError - eNew) / double(predicted)) : -1.f;
    if (rho > 0.f) {
      for (int i = tid; i < P; i += 256) {
        th[i] = thT[i];
      }
    }
    if (tid == 0) {
      if (sp.stepHistory != nullptr) {
        double* sh = sp.stepHistory + (size_t(b) * sp.maxIterations + sp.iteration) * 2;
        sh[0] = double(lambda);
        sh[1] = double(rho);
      }
      if (!(rho >= 0.25f)) {
        lambda = fminf(lambda * sp.lmUp, sp.lmLambdaMax);
      } else if (rho > 0.75f) {
        lambda = fmaxf(lambda * sp.lmDown, sp.lmLambdaMin);
      }
      sp.lambdaPer[b] = lambda;
    }
    return;
  }
  const float scaledError = 1e-3f * float(curError);
  double gd = 0.0; // SubsetGaussNewtonSolverT / GaussNewtonSolverQRT rule: J^T r . delta
  if (sp.doLineSearch == 2) {
    float part = 0.f;
    for (int c = tid; c < n; c += 256) {
      part += dl[c] * jtr[size_t(b) * n + c];
    }
    part = waveReduceSumF(part);
    if (lane == 0) {
      redF[wave] = part;
    }
    __syncthreads();
    gd = (double(redF[0]) + double(redF[1])) + (double(redF[2]) + double(redF[3]));
  }
  float scale = 1.f;
  for (int ls = 0; ls < 10; ++ls) {
    makeTrial(scale);
    const double eNew = trialError();
    if ((curError - eNew) >= (sp.doLineSearch == 2 ? double(1e-4f * scale) * gd : double(scale * scaledError))) {
      break;
    }
    scale *= 0.5f;
  }
  for (int i = tid; i < P; i += 256) {
    th[i] = thT[i];
  }
}

// ---------------------------------------------------------------------------------------------
// TrustRegionQRT on the wide route (momentum/character_solver/trust_region_qr.cpp:52-270): the control flow that
// fusedSolveKernel runs inside one launch, as per-instance state + small kernels between the linear solves.
// ---------------------------------------------------------------------------------------------
__global__ void trustInitKernel(TrustStateDev tr, int B, float radius0) { // initializeSolver (:38-41)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    tr.radius[b] = radius0;
  }
}

// start of an iteration: lambda = 1e-10 ("a tiny lambda just to make sure we don't divide by zero", :86), the first
// trust step; converged instances sit the iteration out
__global__ void trustBeginKernel(TrustStateDev tr, SolveStateDev st, float* lambdaPer, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    const bool idle = st.done[b] != 0;
    tr.lambda[b] = 1e-10f;
    lambdaPer[b] = 1e-20f; // mu = 1e-20 + (lambda - 1e-10): R is seeded with lambda ON its diagonal (online_householder_qr.cpp:133-140)
    tr.phase[b] = idle ? 3 : 0;
    tr.newton[b] = 0;
    tr.step[b] = 0;
    tr.mask[b] = idle ? 1 : 0;
  }
}

// after a linear solve (or after a rejected trial shrank the radius): is the step on the table the one to try (:164,
// :180-181), or does lambda take a Newton update first (Nocedal & Wright eq. 4.44, :191-224: p_l = -(step),
// |q_l|^2 = p_l^T (R^T R)^-1 p_l = |L^-1 p_l|^2 with the factor at hand)?  grid = B, block = 256.
__global__ void __launch_bounds__(256) trustDecideKernel(
    ProblemDev pb, const float* __restrict__ factor, const float* __restrict__ jtr, const double* __restrict__ errIter, SolveStateDev st, StepParams sp) {
  __shared__ __attribute__((aligned(16))) float x[512 + 16];
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ph = sp.tr.phase[b];
  if (st.done[b] != 0 || (ph != 0 && ph != 1)) {
    return;
  }
  const int n = pb.n, NP = (n + 15) & ~15, NB = NP >> 4;
  const TileMaskLanes ml = loadTileMaskLanes(sp.tileMasks, tid);
  const float* dl = sp.delta + size_t(b) * n;
  float p0 = 0.f, p1 = 0.f;
  for (int c = tid; c < NP; c += 256) {
    const float d = c < n ? dl[c] : 0.f;
    x[c] = d;
    p0 += d * d;
    p1 += c < n ? d * jtr[size_t(b) * n + c] : 0.f;
  }
  p0 = waveReduceSumF(p0);
  p1 = waveReduceSumF(p1);
  if (lane == 0) {
    red[wave] = p0;
    red[4 + wave] = p1;
  }
  __syncthreads();
  const float dn2 = (red[0] + red[1]) + (red[2] + red[3]), dg = (red[4] + red[5]) + (red[6] + red[7]);
  const int newton = sp.tr.newton[b];
  int next = 2; // the step is the one to try
  if (newton == 0 && 2.f * dg < FLT_EPSILON * (1.f + float(errIter[b]))) { // :164 (gradientSub_ = 2 J^T r): not worth a step
    next = 3;
  } else if (newton < 3 && sqrtf(dn2) >= 1.05f * sp.tr.radius[b]) { // :180-181
    const float* L = factor + size_t(b) * size_t(NB * (NB + 1) / 2) * 256;
    tiledSweep<true>(L, NB, x, tid, ml.col); // x = L^-1 p_l
    __syncthreads();
    float q = 0.f;
    for (int c = tid; c < n; c += 256) {
      q += x[c] * x[c];
    }
    q = waveReduceSumF(q);
    __syncthreads();
    if (lane == 0) {
      red[wave] = q;
    }
    __syncthreads();
    const float q2 = (red[0] + red[1]) + (red[2] + red[3]);
    if (q2 >= FLT_EPSILON) { // :198
      const float pn = sqrtf(dn2), radius = sp.tr.radius[b];
      const float deltaLambda = (dn2 / q2) * ((pn - radius) / radius);
      if (deltaLambda > 0.f) { // :207: lambda only ever grows
        next = 0; // factor and solve again with the larger damping (the reference appends rows to its QR)
        if (tid == 0) {
          const float lam = sp.tr.lambda[b] + deltaLambda;
          sp.tr.lambda[b] = lam;
          sp.lambdaPer[b] = 1e-20f + (lam - 1e-10f);
          sp.tr.newton[b] = newton + 1;
        }
      }
    }
  }
  if (tid == 0) {
    sp.tr.phase[b] = next;
    sp.tr.mask[b] = next == 0 ? 0 : 1;
  }
}

// end of an iteration: SolverT::solve's bookkeeping (solver.cpp:92-119), once per instance that took part
__global__ void trustEndKernel(SolveStateDev st, StepParams sp, const double* __restrict__ errIter, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || st.done[b] != 0) {
    return;
  }
  const double e = errIter[b];
  const double last = st.lastError[b];
  if (st.errorHistory != nullptr) {
    st.errorHistory[size_t(b) * sp.maxIterations + sp.iteration] = e;
  }
  st.iterations[b] = sp.iteration + 1;
  st.finalError[b] = e;
  const bool converged = fabs(last - e) / (fabs(e) + double(FLT_MIN)) <= double(sp.threshold) * double(FLT_EPSILON);
  if (sp.iteration >= sp.minIterations && converged) {
    st.done[b] = 1;
  }
  st.lastError[b] = e;
}

hipError_t launchTrustInit(const TrustStateDev& tr, int B, float radius0, hipStream_t stream) {
  hipLaunchKernelGGL(trustInitKernel, dim3((B + 255) / 256), dim3(256), 0, stream, tr, B, radius0);
  return hipGetLastError();
}
hipError_t launchTrustBegin(const TrustStateDev& tr, const SolveStateDev& st, float* lambdaPer, int B, hipStream_t stream) {
  hipLaunchKernelGGL(trustBeginKernel, dim3((B + 255) / 256), dim3(256), 0, stream, tr, st, lambdaPer, B);
  return hipGetLastError();
}
hipError_t launchTrustDecide(const ProblemDev& pb, const float* factor, const float* jtr, const double* errIter, const SolveStateDev& st, const StepParams& sp, hipStream_t stream) {
  if (pb.n > 512) {
    return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(trustDecideKernel, dim3(pb.B), dim3(256), 0, stream, pb, factor, jtr, errIter, st, sp);
  return hipGetLastError();
}
hipError_t launchTrustEnd(const SolveStateDev& st, const StepParams& sp, const double* errIter, int B, hipStream_t stream) {
  hipLaunchKernelGGL(trustEndKernel, dim3((B + 255) / 256), dim3(256), 0, stream, st, sp, errIter, B);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// small bookkeeping kernels
// ---------------------------------------------------------------------------------------------
__global__ void solveInitKernel(SolveStateDev st, int B, float* lambdaPer, float lambda0, float* diagAcc) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    if (lambdaPer != nullptr) {
      lambdaPer[b] = lambda0;
    }
    if (diagAcc != nullptr) {
      *reinterpret_cast<float4*>(diagAcc + 4 * size_t(b)) = float4{0.f, 0.f, 0.f, 0.f};
    }
    st.done[b] = 0;
    st.iterations[b] = 0;
    st.status[b] = 0;
    st.lastError[b] = DBL_MAX; // solver.cpp:84-85
    st.finalError[b] = DBL_MAX;
  }
}

// NaN/Inf guard of the batched driver: revert to the initial parameters
// (pymomentum/tensor_ik/tensor_ik.cpp:168-173).  One wavefront per instance.
__global__ void __launch_bounds__(64)
solveFinalizeKernel(float* __restrict__ theta, const float* __restrict__ thetaInit, int P, SolveStateDev st, const float* __restrict__ diagAcc) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float* th = theta + size_t(b) * P;
  const float* ti = thetaInit + size_t(b) * P;
  int bad = 0;
  float th2 = 0.f;
  for (int i = lane; i < P; i += 64) {
    const float v = th[i];
    if (!isfinite(v)) {
      bad = 1;
    }
    th2 += v * v;
  }
  bad = __any(bad);
  if (diagAcc != nullptr && st.diag != nullptr) { // the wide route's precision estimate (fusedSolveKernel's epilogue has the formula)
    th2 = waveReduceSumF(th2);
    if (lane == 0) {
      const float* a = diagAcc + 4 * size_t(b);
      const float ratio = a[1] > 0.f ? kPivotFloorOrOne / a[1] : 1.f;
      const float est = a[0] > 0.f ? kPrecisionGain * FLT_EPSILON * a[0] / kPivotFloorOrOne : kPrecisionGain * FLT_EPSILON / ratio;
      if (!bad && st.precisionBound > 0.f && !(est <= st.precisionBound)) {
        st.status[b] |= 8; // MMX_SOLVE_PRECISION_SUSPECT
      }
      float* dg = st.diag + 4 * size_t(b);
      dg[0] = est;
      dg[1] = ratio;
      dg[2] = sqrtf(a[2]);
      dg[3] = bad ? 0.f : sqrtf(th2);
    }
  }
  if (bad) {
    for (int i = lane; i < P; i += 64) {
      th[i] = ti[i];
    }
    if (lane == 0) {
      st.status[b] = 1; // MMX_SOLVE_NONFINITE
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host-callable launchers (declared in mmx_kernels.hpp)
// ---------------------------------------------------------------------------------------------
// Profiling aid: the store pattern of J-assembly without any kinematics -- one workgroup of `waves`
// wavefronts per instance writes the instance's column-major M x P block column by column, lane u the
// rows 3u..3u+2 (the same 12-byte stores, streaming when the unit count fits one chunk, plain and
// chunk after chunk otherwise, like fkJacobianKernel).  What this reaches is the ceiling the write
// pattern itself sets for the graded kernel on the box at hand (bench.py: roofline.store_pattern_gbs).
template <bool kNt>
__global__ void __launch_bounds__(1024) storePatternKernel(float* __restrict__ jac, int M, int P, int waves) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* jb = jac + size_t(blockIdx.x) * size_t(M) * size_t(P);
  const float v = float(blockIdx.x);
  const int U = M / 3;
  for (int c = wave; c < P; c += waves) {
    for (int u = lane; u < U; u += 64) {
      store3<kNt>(jb + size_t(c) * M + 3 * size_t(u), v, v, v);
    }
  }
}

hipError_t launchStorePattern(float* jac, int B, int M, int P, int waves, hipStream_t stream, hipEvent_t startEvent, hipEvent_t stopEvent) {
  if (waves < 1 || waves > 16) {
    return hipErrorInvalidValue;
  }
  if (M / 3 <= 64) {
    hipExtLaunchKernelGGL((storePatternKernel<true>), dim3(B), dim3(64 * waves), 0, stream, startEvent, stopEvent, 0, jac, M, P, waves);
  } else {
    hipExtLaunchKernelGGL((storePatternKernel<false>), dim3(B), dim3(64 * waves), 0, stream, startEvent, stopEvent, 0, jac, M, P, waves);
  }
  return hipGetLastError();
}

int fkJacobianWavesPerInstance(const RigDev& rig, const ProblemDev& pb, bool withJacobian) {
  // Wavefronts per instance.  J-assembly: several waves share one instance (FK over all threads, the column program dealt to
  // the waves) -- fewer instances are then in flight at a time, and write bandwidth on this part falls with the footprint of
  // the concurrently written regions (scripts/probes/store_k.hip: 6.7 TB/s when a workgroup writes 4 KB and ends, 5.4 TB/s at
  // 96 KB per workgroup).  Round 6 sweep, one box, 20 launches each, two sweeps, fraction of the 8 TB/s peak at one / three /
  // four waves (two and eight lose everywhere): B = 4096: 0.517 / 0.543 / 0.582; 8192: 0.58 / 0.614 / 0.611; 16 384: 0.63 / 0.65 /
  // 0.63; 32 768: 0.66 / 0.70 / 0.67; 40 000: 0.68 / 0.70 / 0.67; 65 536: 0.69 / 0.70 / 0.68 -- four below 12 288 instances, three
  // from there on.  The LDS-bound large rigs (a 300-joint instance needs 25 KB; BASELINE configs[4], 1.09 MB of J per instance):
  // B = 2048: four / eight / sixteen waves 0.516 / 0.534 / 0.515; B = 8192: 0.554-0.571 / 0.574-0.590 / 0.589 (two boxes) -- eight,
  // sixteen from 8192 instances on.  FK only (no J): one wave per instance from 2048 instances on (16 against 23 us at 4096).
  const bool smallRig = fkJacobianLdsBytes(rig.J, rig.P, pb.U) <= 12 * 1024;
  if (!smallRig) {
    return withJacobian ? (pb.B >= 8192 ? 16 : 8) : 4;
  }
  if (pb.B < 2048) {
    return 4;
  }
  return withJacobian ? (pb.B < 12288 ? 4 : 3) : 1;
}

// =============================================================================================
// MMX_LAYOUT_ROW_MAJOR: J[b][i * P + p] from the column-major J[b][p * M + i] the assembly kernel writes
// (the reference's own layout, math/resizeable_matrix.h:18,32-34; row-major is offered for callers that
// hold torch-style [B][M][P] tensors, pymomentum/tensor_ik/tensor_error_function.cpp).  A 32 x 32 tile
// per workgroup through LDS (stride 33: conflict-free both ways), reads and writes both in 128-byte runs.
// =============================================================================================
__global__ void __launch_bounds__(256) transposeJacobianKernel(const float* __restrict__ colMajor, float* __restrict__ rowMajor, int M, int P) {
  __shared__ float tile[32][33];
  const size_t base = size_t(blockIdx.z) * size_t(M) * size_t(P);
  const int i0 = 32 * blockIdx.x, p0 = 32 * blockIdx.y; // rows i, columns p
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) { // column p0 + k, rows i0 + tx (contiguous in the column-major source)
    const int p = p0 + k, i = i0 + tx;
    tile[k][tx] = (p < P && i < M) ? colMajor[base + size_t(p) * M + i] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) { // row i0 + k, columns p0 + tx (contiguous in the row-major destination)
    const int i = i0 + k, p = p0 + tx;
    if (i < M && p < P) {
      rowMajor[base + size_t(i) * P + p] = tile[tx][k];
    }
  }
}

hipError_t launchTransposeJacobian(const float* colMajor, float* rowMajor, int B, int M, int P, hipStream_t stream) {
  if (B <= 0 || M <= 0 || P <= 0) {
    return hipSuccess;
  }
  for (int b0 = 0; b0 < B; b0 += 65535) { // gridDim.z limit
    const int nb = B - b0 < 65535 ? B - b0 : 65535;
    const size_t off = size_t(b0) * size_t(M) * size_t(P);
    hipLaunchKernelGGL(transposeJacobianKernel, dim3((M + 31) / 32, (P + 31) / 32, nb), dim3(256), 0, stream, colMajor + off, rowMajor + off, M, P);
  }
  return hipGetLastError();
}

size_t fkJacobianLdsBytes(int J, int P, int U) {
  return fkJacobianLdsFloats(J, P, U) * sizeof(float);
}

hipError_t launchFkJacobian(
    const RigDev& rig,
    const ProblemDev& pb,
    const float* theta,
    float* jac,
    float* res,
    double* err,
    float* state,
    const int32_t* done,
    hipStream_t stream,
    hipEvent_t startEvent,
    hipEvent_t stopEvent,
    bool accurateFk) {
  size_t lds = fkJacobianLdsBytes(rig.J, rig.P, pb.U);
  if (accurateFk) { // (rigs whose double buffers do not fit the default 64 KB of dynamic LDS keep the single-precision rounds)
    const size_t ldsD = (((fkJacobianLdsFloats(rig.J, rig.P, pb.U) + 1) & ~size_t(1)) + 2 * fkBufFloats(rig.J)) * sizeof(float);
    if (ldsD <= 64 * 1024) {
      lds = ldsD;
    } else {
      accurateFk = false;
    }
  }
  const int wpi = fkJacobianWavesPerInstance(rig, pb, jac != nullptr);
  // non-temporal column stores throughout (measured better at every batch size once they were really emitted: see store3());
  // the structurally zero columns: one wave per instance alternates their position, several waves: those without joints write them first
  const int zeroPhase = (wpi == 1 ? 1 : 0) | (accurateFk ? 0x100 : 0);
  // (A two-kernel form -- FK + units once per instance handed over through HBM, then four adjacent columns per short-lived
  // workgroup -- was built and measured in round 2: 95-150 us against 86 us for this one at B = 4096, because the L2s are
  // written back between the two kernels and every column workgroup's first loads miss.  Removed in round 3)
#define MMX_FKJ(W_, WPI_, S_)                                                                                                  \
  hipExtLaunchKernelGGL(                                                                                                       \
      (fkJacobianKernel<W_, WPI_, S_>), dim3(pb.B), dim3(64 * WPI_), lds, stream, startEvent, stopEvent, 0, rig, pb, theta, jac, res, err, state, done, zeroPhase)
  if (jac != nullptr) {
    if (wpi == 16) {
      MMX_FKJ(true, 16, true);
    } else if (wpi == 8) {
      MMX_FKJ(true, 8, true);
    } else if (wpi == 3) {
      MMX_FKJ(true, 3, true);
    } else if (wpi == 4) {
      MMX_FKJ(true, 4, true);
    } else {
      MMX_FKJ(true, 1, true);
    }
  } else {
    if (wpi == 4) {
      MMX_FKJ(false, 4, false);
    } else {
      MMX_FKJ(false, 1, false);
    }
  }
#undef MMX_FKJ
  if (pb.G + pb.NE > 0 && (jac != nullptr || res != nullptr || err != nullptr)) {
    const size_t jl = jointBlocksLdsBytes(rig.J, rig.P, pb.G + pb.NE);
    if (jl > 64 * 1024) {
      hipError_t rc = hipFuncSetAttribute(
          jac != nullptr ? reinterpret_cast<const void*>(jointBlocksKernel<true>) : reinterpret_cast<const void*>(jointBlocksKernel<false>),
          hipFuncAttributeMaxDynamicSharedMemorySize,
          int(jl));
      if (rc != hipSuccess) {
        return rc;
      }
    }
    if (jac != nullptr) {
      hipLaunchKernelGGL(jointBlocksKernel<true>, dim3(pb.B), dim3(256), jl, stream, rig, pb, theta, jac, res, err, done);
    } else {
      hipLaunchKernelGGL(jointBlocksKernel<false>, dim3(pb.B), dim3(256), jl, stream, rig, pb, theta, jac, res, err, done);
    }
  }
  if (pb.M > pb.rowsJoint && (jac != nullptr || res != nullptr || err != nullptr)) {
    hipLaunchKernelGGL(
        parameterRowsKernel, dim3(pb.B), dim3(256), parameterRowsLdsBytes(rig.P, pb.NL), stream, rig, pb, rig.P, theta, jac, res, err, done);
  }
  return hipGetLastError();
}

int normalEquationsChunkRows(int n) { // 32 rows of J per staged chunk while they fit the LDS, else 16
  const int nT = (n + 3) >> 2;
  return size_t(kNeChunk) * size_t(4 * nT + 5) * sizeof(float) <= 160 * 1024 - 64 ? kNeChunk : kNeChunk / 2;
}
size_t normalEquationsLdsBytes(int n) {
  const int nT = (n + 3) >> 2, chunk = normalEquationsChunkRows(n);
  return size_t(chunk) * size_t(4 * nT + 4) * sizeof(float) + chunk * sizeof(float);
}

hipError_t launchNormalEquations(
    const ProblemDev& pb,
    int P,
    const float* jac,
    const float* res,
    float* jtj,
    float* jtr,
    const int32_t* done,
    bool lowerOnly,
    hipStream_t stream) {
  // wide systems go to the matrix cores; the staging loop of that kernel covers n <= 384, g <= 512 columns
  if (pb.n >= 32 && pb.n <= 384) {
    const size_t lds = normalEquationsMfmaLdsBytes(pb.n);
    const int NB = (pb.n + 15) >> 4, T = NB * (NB + 1) / 2;
#define MMX_NE_LAUNCH_V(TPW_, V_)                                                                                       \
  do {                                                                                                                  \
    static LdsLimitCache ldsLimit;                                                                                      \
    if (lds > 64 * 1024) {                                                                                              \
      hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(normalEquationsMfmaKernel<TPW_, V_>), 160 * 1024 - 64); \
      if (rc != hipSuccess) {                                                                                           \
        return rc;                                                                                                      \
      }                                                                                                                 \
    }                                                                                                                   \
    hipLaunchKernelGGL((normalEquationsMfmaKernel<TPW_, V_>), dim3(pb.B), dim3(256), lds, stream, pb, P, jac, res, jtj, jtr, done, lowerOnly ? 0 : 1); \
  } while (0)
#define MMX_NE_LAUNCH(TPW_)        \
  do {                             \
    if (vec) {                     \
      MMX_NE_LAUNCH_V(TPW_, true); \
    } else {                       \
      MMX_NE_LAUNCH_V(TPW_, false);\
    }                              \
  } while (0)
    const bool vec = (pb.M & 3) == 0 && (reinterpret_cast<uintptr_t>(jac) & 15) == 0; // 16-byte column pieces
    const int need = (T + 3) / 4; // tiles per wave for a single pass
    if (need <= 4) {
      MMX_NE_LAUNCH(4);
    } else if (need <= 8) {
      MMX_NE_LAUNCH(8);
    } else if (need <= 12) {
      MMX_NE_LAUNCH(12);
    } else if (need <= 16) {
      MMX_NE_LAUNCH(16);
    } else if (need <= 24) {
      MMX_NE_LAUNCH(24);
    } else if (need <= 32) {
      MMX_NE_LAUNCH(32);
    } else if (need <= 40) {
      MMX_NE_LAUNCH(40);
    } else {
      MMX_NE_LAUNCH(48);
    }
#undef MMX_NE_LAUNCH
#undef MMX_NE_LAUNCH_V
    return hipGetLastError();
  }
  if (pb.n > kMaxSolved) {
    return hipErrorInvalidValue;
  }
  {
    static LdsLimitCache ldsLimit;
    const size_t lds = normalEquationsLdsBytes(pb.n);
    if (lds > 64 * 1024) {
      hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(normalEquationsKernel), 160 * 1024 - 64);
      if (rc != hipSuccess) {
        return rc;
      }
    }
    hipLaunchKernelGGL(normalEquationsKernel, dim3(pb.B), dim3(256), lds, stream, pb, P, jac, res, jtj, jtr, done, normalEquationsChunkRows(pb.n));
  }
  return hipGetLastError();
}

size_t choleskyStepLdsBytes(int n, int M) {
  return (size_t(n) * size_t((n + 1) | 1) + 3 * size_t(n) + size_t(M) + 12) * sizeof(float);
}

hipError_t launchCholeskyStep(
    const ProblemDev& pb,
    int P,
    const float* jac,
    const float* res,
    const float* jtj,
    const float* jtr,
    const double* errIter,
    float* theta,
    const SolveStateDev& st,
    const StepParams& sp,
    float* factor,
    hipStream_t stream) {
  size_t lds = choleskyStepLdsBytes(pb.n, pb.M);
  if (lds > 160 * 1024 && factor != nullptr) { // large system: left-looking factor in its own tile-major scratch
    // rows of J per refinement chunk: 32 (every column contributes one full 128-byte line per chunk) while the
    // chunk's loads fit the prefetch registers, else 16
    int chunkRows = size_t(pb.n) * 8 <= 256 * size_t(kChunkLoads) ? 32 : 16;
    lds = tiledLdsFloats(pb.n, chunkRows, nullptr, nullptr) * sizeof(float);
    if (lds > 160 * 1024 - 64) { // (beyond ~1900 solved parameters the factor's panel leaves room for eight rows only)
      chunkRows = 8;
      lds = tiledLdsFloats(pb.n, chunkRows, nullptr, nullptr) * sizeof(float);
    }
    if (pb.n > kMaxSolved || lds > 160 * 1024 - 64) {
      return hipErrorInvalidValue;
    }
#define MMX_TILED_STEP(KM_)                                                                                          \
  do {                                                                                                               \
    if (lds > 64 * 1024) {                                                                                           \
      static LdsLimitCache ldsLimit;                                                                                 \
      hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(choleskyStepTiledKernel<KM_>), 160 * 1024 - 64); \
      if (rc != hipSuccess) {                                                                                        \
        return rc;                                                                                                   \
      }                                                                                                              \
    }                                                                                                                \
    hipLaunchKernelGGL(                                                                                              \
        choleskyStepTiledKernel<KM_>, dim3(pb.B), dim3(256), lds, stream, pb, P, jac, res, jtj, jtr, factor, errIter, theta, st, sp, chunkRows); \
  } while (0)
    if (pb.n <= 512) {
      MMX_TILED_STEP(2);
    } else {
      MMX_TILED_STEP(kNeCols);
    }
#undef MMX_TILED_STEP
    return hipGetLastError();
  }
  if (lds > 160 * 1024) {
    return hipErrorInvalidValue; // (large systems need the factor scratch)
  }
  if (lds > 64 * 1024) {
    hipError_t rc = hipFuncSetAttribute(
        reinterpret_cast<const void*>(choleskyStepKernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (rc != hipSuccess) {
      return rc;
    }
  }
  hipLaunchKernelGGL(
      choleskyStepKernel, dim3(pb.B), dim3(256), lds, stream, pb, P, jac, res, jtj, jtr, errIter, theta, st, sp);
  return hipGetLastError();
}

hipError_t launchCholeskyFactorTiled(
    const ProblemDev& pb,
    int P,
    const float* jtj,
    const float* jtr,
    float* factor,
    float* dvec,
    int32_t* refState,
    const double* errIter,
    float* theta,
    const SolveStateDev& st,
    const StepParams& sp,
    hipStream_t stream) {
  if (pb.n > 512) {
    return hipErrorInvalidValue;
  }
  // the factor resident in LDS when its structurally non-zero tiles + the vectors fit half a CU (two workgroups per CU)
  {
    const size_t NP = (size_t(pb.n) + 15) & ~size_t(15);
    const size_t resident = (size_t(sp.numTiles) * 256 + 3 * NP + 8 + 96) * sizeof(float);
    const bool useResident = sp.numTiles > 0 && resident <= 80 * 1024 - 64;
    if (useResident) {
      constexpr int kW = 4;
      static LdsLimitCache ldsLimit;
      hipError_t rc = ldsLimit.ensure(reinterpret_cast<const void*>(choleskyFactorResidentKernel<kW>), resident);
      if (rc != hipSuccess) {
        return rc;
      }
      hipLaunchKernelGGL(choleskyFactorResidentKernel<kW>, dim3(pb.B), dim3(64 * kW), resident, stream, pb, P, jtj, jtr, factor, dvec, refState, errIter, theta, st, sp, int(sp.numTiles));
      return hipGetLastError();
    }
  }
  const size_t lds = tiledLdsFloats(pb.n, 0, nullptr, nullptr) * sizeof(float);
  if (lds > 64 * 1024) { // two panels of a 450-512-parameter system: beyond the default dynamic LDS limit
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(choleskyFactorTiledKernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (rc != hipSuccess) {
      return rc;
    }
  }
  hipLaunchKernelGGL(choleskyFactorTiledKernel, dim3(pb.B), dim3(256), lds, stream, pb, P, jtj, jtr, factor, dvec, refState, errIter, theta, st, sp);
  return hipGetLastError();
}

hipError_t launchCholeskyFinishTiled(
    const ProblemDev& pb,
    int P,
    const float* factor,
    float* dvec,
    const float* rhoVec,
    int32_t* refState,
    const double* errIter,
    float* theta,
    const SolveStateDev& st,
    const StepParams& sp,
    int round,
    hipStream_t stream) {
  // (the sweeps on a factor brought into LDS in one round trip -- the form choleskyFactorResidentKernel's first solve uses --
  // were built for this stage too and measured slower: 0.61 against 0.53 ms on cfg5; two workgroups per CU instead of four)
  hipLaunchKernelGGL(choleskyFinishTiledKernel, dim3(pb.B), dim3(256), 0, stream, pb, P, factor, dvec, rhoVec, refState, errIter, theta, st, sp, round);
  return hipGetLastError();
}

static __global__ void __launch_bounds__(256) zeroFillKernel(uint32_t* __restrict__ p, size_t words) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < words; i += size_t(gridDim.x) * 256) {
    p[i] = 0u;
  }
}
hipError_t zeroAsync(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0 || p == nullptr) {
    return hipSuccess;
  }
  if ((bytes & 3) != 0 || (reinterpret_cast<uintptr_t>(p) & 3) != 0) {
    return hipErrorInvalidValue;
  }
  const size_t words = bytes / 4;
  const unsigned grid = unsigned(std::min<size_t>((words + 255) / 256, 8192));
  hipLaunchKernelGGL(zeroFillKernel, dim3(grid), dim3(256), 0, stream, static_cast<uint32_t*>(p), words);
  return hipGetLastError();
}

hipError_t launchSolveInit(const SolveStateDev& st, int B, float* lambdaPer, float lambda0, hipStream_t stream, float* diagAcc) {
  hipLaunchKernelGGL(solveInitKernel, dim3((B + 255) / 256), dim3(256), 0, stream, st, B, lambdaPer, lambda0, diagAcc);
  return hipGetLastError();
}

hipError_t launchStepUpdate(
    const RigDev& rig,
    const ProblemDev& pb,
    float* theta,
    const float* jtr,
    const double* errIter,
    const StepParams& sp,
    hipStream_t stream) {
  const size_t lds = (sideFkLdsFloats(rig.J) + 2 * size_t((rig.P + 3) & ~3)) * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(stepUpdateKernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (rc != hipSuccess) {
      return rc;
    }
  }
  hipLaunchKernelGGL(stepUpdateKernel, dim3(pb.B), dim3(256), lds, stream, rig, pb, theta, jtr, errIter, sp);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) paramHistoryFinalizeKernel(float* __restrict__ hist, const int32_t* __restrict__ iterations, int maxIterations, int P) {
  const int b = blockIdx.x;
  const size_t rows = size_t(maxIterations) * size_t(P);
  for (size_t i = size_t(iterations[b]) * size_t(P) + threadIdx.x; i < rows; i += 256) {
    hist[size_t(b) * rows + i] = 0.f;
  }
}

hipError_t launchParamHistoryFinalize(float* paramHistory, const int32_t* iterations, int B, int maxIterations, int P, hipStream_t stream) {
  if (paramHistory == nullptr || B <= 0 || maxIterations <= 0) {
    return hipSuccess;
  }
  hipLaunchKernelGGL(paramHistoryFinalizeKernel, dim3(B), dim3(256), 0, stream, paramHistory, iterations, maxIterations, P);
  return hipGetLastError();
}

hipError_t launchSolveFinalize(float* theta, const float* thetaInit, int P, const SolveStateDev& st, int B, hipStream_t stream, const float* diagAcc) {
  hipLaunchKernelGGL(solveFinalizeKernel, dim3(B), dim3(64), 0, stream, theta, thetaInit, P, st, diagAcc);
  return hipGetLastError();
}

} // namespace mmx
