// mmx_oracle_capi.cpp -- C entry points of the CPU ORACLE for ctypes (test infrastructure only;
// see mmx_oracle.hpp for the pinning statement).  Built into oracle/libmmx_oracle.so by
// oracle/Makefile.  Uses the rig / constraint descriptors of include/mmx.h so that tests feed the
// oracle and the HIP path from the same buffers.
#include "mmx_oracle.hpp"

#include "../include/mmx.h"

#include <atomic>
#include <thread>

using namespace mmx_oracle;

namespace {

Rig makeRig(const mmx_rig_desc* d) {
  Rig r;
  r.J = d->num_joints;
  r.P = d->num_params;
  const int R = kParametersPerJoint * r.J;
  r.parent.assign(d->parent, d->parent + r.J);
  r.preRot.assign(d->pre_rotation, d->pre_rotation + 4 * r.J);
  r.offset.assign(d->translation_offset, d->translation_offset + 3 * r.J);
  r.outer.assign(d->pt_outer, d->pt_outer + R + 1);
  const int nnz = r.outer[R];
  r.inner.assign(d->pt_inner, d->pt_inner + nnz);
  r.value.assign(d->pt_value, d->pt_value + nnz);
  if (d->pt_offsets != nullptr) {
    r.ptOffsets.assign(d->pt_offsets, d->pt_offsets + R);
  } else {
    r.ptOffsets.assign(R, 0.f);
  }
  return r;
}

template <class T>
Constraints<T> makeConstraints(
    int Kp,
    const int32_t* posParent,
    int Ko,
    const int32_t* oriParent,
    const mmx_constraint_data* c,
    size_t b,
    int P) {
  Constraints<T> cs;
  cs.Kp = Kp;
  cs.Ko = Ko;
  cs.posParent = posParent;
  cs.oriParent = oriParent;
  cs.posOffset = c->pos_offset ? c->pos_offset + b * Kp * 3 : nullptr;
  cs.posTarget = c->pos_target ? c->pos_target + b * Kp * 3 : nullptr;
  cs.posWeight = c->pos_weight ? c->pos_weight + b * Kp : nullptr;
  cs.oriOffset = c->ori_offset ? c->ori_offset + b * Ko * 4 : nullptr;
  cs.oriTarget = c->ori_target ? c->ori_target + b * Ko * 4 : nullptr;
  cs.oriWeight = c->ori_weight ? c->ori_weight + b * Ko : nullptr;
  // per-element error-function weights: buildMomentumErrorFunctions gives every error function of element iBatch
  // setWeight(errorFunctionWeights[iBatch][weightsMap[iErr]]) (pymomentum/tensor_ik/tensor_ik_utility.cpp:162-177)
  auto fnw = [&](int col) {
    return (c->function_weights != nullptr && col < c->num_function_weights) ? c->function_weights[b * size_t(c->num_function_weights) + size_t(col)] : 1.f;
  };
  cs.posFunctionWeight = c->pos_function_weight * fnw(0);
  cs.oriFunctionWeight = c->ori_function_weight * fnw(1);
  if (c->pos_loss_c > 0.f) {
    cs.posLoss = Loss<T>(c->pos_loss_alpha == MMX_LOSS_WELSCH ? std::numeric_limits<T>::lowest() : T(c->pos_loss_alpha), T(c->pos_loss_c));
  }
  if (c->ori_loss_c > 0.f) {
    cs.oriLoss = Loss<T>(c->ori_loss_alpha == MMX_LOSS_WELSCH ? std::numeric_limits<T>::lowest() : T(c->ori_loss_alpha), T(c->ori_loss_c));
  }
  for (int32_t i = 0; i < c->num_joint_blocks; ++i) {
    const mmx_joint_constraint_block& jb = c->joint_blocks[i];
    JointBlock<T> blk;
    blk.type = jb.type;
    blk.count = jb.count;
    blk.parent = jb.parent;
    const size_t o = b * size_t(jb.count);
    blk.localPoint = jb.local_point ? jb.local_point + 3 * o : nullptr;
    blk.localDir = jb.local_dir ? jb.local_dir + 3 * o : nullptr;
    blk.global = jb.global ? jb.global + 3 * o : nullptr;
    blk.planeD = jb.plane_d ? jb.plane_d + o : nullptr;
    blk.weight = jb.weight + o;
    blk.functionWeight = jb.function_weight * fnw(4 + i);
    if (jb.loss_c > 0.f) {
      blk.loss = Loss<T>(jb.loss_alpha == MMX_LOSS_WELSCH ? std::numeric_limits<T>::lowest() : T(jb.loss_alpha), T(jb.loss_c));
    }
    cs.blocks.push_back(blk);
  }
  cs.NE = c->num_ellipsoid_limits;
  cs.ellipsoids = c->ellipsoid_limits;
  cs.P = P;
  cs.NL = c->num_limits;
  cs.limits = c->limits;
  cs.limFunctionWeight = c->limit_function_weight * fnw(2);
  cs.mpTarget = c->model_target ? c->model_target + b * size_t(P) : nullptr;
  cs.mpWeights = c->model_weights ? c->model_weights + b * size_t(P) : nullptr;
  cs.mpFunctionWeight = c->model_function_weight * fnw(3);
  return cs;
}

Options makeOptions(const mmx_gn_options* o, int useBlockJtJ) {
  Options r;
  r.minIterations = o->min_iterations;
  r.maxIterations = o->max_iterations;
  r.threshold = o->threshold;
  r.regularization = o->regularization;
  r.doLineSearch = o->do_line_search;
  r.useBlockJtJ = (useBlockJtJ & 1) != 0; // bit 0: SolverFunctionT::getJtJR block accumulation; bit 1: GaussNewtonSolverQRT
  r.useQR = (useBlockJtJ & 2) != 0;
  r.stepRule = o->step_rule;
  r.lmLambdaMin = o->lm_lambda_min;
  r.lmLambdaMax = o->lm_lambda_max;
  r.lmUp = o->lm_up;
  r.lmDown = o->lm_down;
  r.trustRegionRadius = o->trust_region_radius > 0.f ? o->trust_region_radius : 1.f;
  return r;
}

template <class T>
int skeletonState(
    const mmx_rig_desc* d,
    const T* theta,
    T* world, // [J][8]
    T* local, // [J][8]
    T* transAxis, // [J][3][3] row-major
    T* rotAxis, // [J][3][3] row-major
    T* jointParams) { // [7J]
  const Rig rig = makeRig(d);
  std::vector<T> jp(size_t(kParametersPerJoint) * rig.J);
  applyParameterTransform<T>(rig, theta, jp.data());
  std::vector<JointState<T>> st;
  setSkeletonState<T>(rig, jp.data(), st);
  for (int j = 0; j < rig.J; ++j) {
    auto put = [&](T* o, const Xf<T>& x) {
      o[0] = x.t.x, o[1] = x.t.y, o[2] = x.t.z;
      o[3] = x.q.x, o[4] = x.q.y, o[5] = x.q.z, o[6] = x.q.w;
      o[7] = x.s;
    };
    if (world) {
      put(world + 8 * j, st[j].world);
    }
    if (local) {
      put(local + 8 * j, st[j].local);
    }
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) {
        if (transAxis) {
          transAxis[9 * j + 3 * r + c] = st[j].translationAxis.m[r][c];
        }
        if (rotAxis) {
          rotAxis[9 * j + 3 * r + c] = st[j].rotationAxis.m[r][c];
        }
      }
    }
  }
  if (jointParams) {
    std::copy(jp.begin(), jp.end(), jointParams);
  }
  return 0;
}

template <class T>
int evalJacobian(
    const mmx_rig_desc* d,
    int Kp,
    const int32_t* posParent,
    int Ko,
    const int32_t* oriParent,
    const mmx_constraint_data* c,
    const uint8_t* enabled,
    const T* theta,
    T* jac,
    T* res,
    double* err) {
  const Rig rig = makeRig(d);
  SolverFunction<T> fn(rig, makeConstraints<T>(Kp, posParent, Ko, oriParent, c, 0, d->num_params));
  if (enabled) {
    fn.setEnabledParameters(enabled);
  }
  const double e = fn.getJacobian(theta, jac, res);
  if (err) {
    *err = e;
  }
  return 0;
}

template <class T>
int getError(
    const mmx_rig_desc* d,
    int Kp,
    const int32_t* posParent,
    int Ko,
    const int32_t* oriParent,
    const mmx_constraint_data* c,
    const T* theta,
    double* err) {
  const Rig rig = makeRig(d);
  SolverFunction<T> fn(rig, makeConstraints<T>(Kp, posParent, Ko, oriParent, c, 0, d->num_params));
  *err = fn.getError(theta);
  return 0;
}

template <class T>
int solveOne(
    const Rig& rig,
    const Constraints<T>& cs,
    const uint8_t* enabled,
    const Options& opt,
    T* theta,
    double* err,
    int32_t* iters,
    int32_t* status,
    double* hist,
    T* jtj,
    T* jtr,
    double* lamHist = nullptr, // [maxIterations] LM schedule: damping of iteration i (0 beyond the run / other step rules)
    double* rhoHist = nullptr) { // [maxIterations] LM schedule: gain ratio of iteration i
  SolverFunction<T> fn(rig, cs);
  if (enabled) {
    fn.setEnabledParameters(enabled);
  }
  std::vector<T> init(theta, theta + rig.P);
  SolveResult<T> r = solveGaussNewton<T>(fn, opt, theta);
  int st = r.notPD ? MMX_SOLVE_NOT_PD : MMX_SOLVE_OK;
  // NaN/Inf guard of the batched driver (pymomentum/tensor_ik/tensor_ik.cpp:168-173)
  bool bad = false;
  for (int i = 0; i < rig.P; ++i) {
    if (!std::isfinite(theta[i])) {
      bad = true;
    }
  }
  if (bad) {
    std::copy(init.begin(), init.end(), theta);
    st = MMX_SOLVE_NONFINITE;
  }
  if (err) {
    *err = r.error;
  }
  if (iters) {
    *iters = r.iterations;
  }
  if (status) {
    *status = st;
  }
  if (hist) {
    for (int i = 0; i < opt.maxIterations; ++i) {
      hist[i] = i < r.iterations ? r.errorHistory[i] : 0.0;
    }
  }
  if (jtj) {
    std::copy(r.lastJtJ.begin(), r.lastJtJ.end(), jtj);
  }
  if (jtr) {
    std::copy(r.lastJtr.begin(), r.lastJtr.end(), jtr);
  }
  for (int i = 0; i < opt.maxIterations; ++i) {
    if (lamHist) {
      lamHist[i] = size_t(i) < r.lambdaHistory.size() ? r.lambdaHistory[i] : 0.0;
    }
    if (rhoHist) {
      rhoHist[i] = size_t(i) < r.gainRatioHistory.size() ? r.gainRatioHistory[i] : 0.0;
    }
  }
  return 0;
}

template <class T>
int solveBatch(
    const mmx_rig_desc* d,
    int B,
    int Kp,
    const int32_t* posParent,
    int Ko,
    const int32_t* oriParent,
    const mmx_constraint_data* c,
    const uint8_t* enabled,
    const mmx_gn_options* o,
    int useBlockJtJ,
    T* theta, // [B][P]
    double* err,
    int32_t* iters,
    int32_t* status,
    double* hist, // [B][maxIter] or null
    int nthreads,
    double* lamHist = nullptr, // [B][maxIter] or null (LM schedule)
    double* rhoHist = nullptr) {
  const Rig rig = makeRig(d);
  const Options opt = makeOptions(o, useBlockJtJ);
  // one independent solver + function per task, like dispenso::parallel_for(0, nBatch, ...) in
  // pymomentum/tensor_ik/tensor_ik.cpp:127-177
  std::atomic<int> next{0};
  auto work = [&]() {
    for (;;) {
      const int b = next.fetch_add(1);
      if (b >= B) {
        break;
      }
      solveOne<T>(
          rig,
          makeConstraints<T>(Kp, posParent, Ko, oriParent, c, size_t(b), d->num_params),
          enabled,
          opt,
          theta + size_t(b) * rig.P,
          err ? err + b : nullptr,
          iters ? iters + b : nullptr,
          status ? status + b : nullptr,
          hist ? hist + size_t(b) * opt.maxIterations : nullptr,
          nullptr,
          nullptr,
          lamHist ? lamHist + size_t(b) * opt.maxIterations : nullptr,
          rhoHist ? rhoHist + size_t(b) * opt.maxIterations : nullptr);
    }
  };
  if (nthreads <= 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
      th.emplace_back(work);
    }
    for (auto& t : th) {
      t.join();
    }
  }
  return 0;
}

template <class T>
int mockSolve(int P, const uint8_t* enabled, const mmx_gn_options* o, int useBlockJtJ, T* theta, double* err, int32_t* iters, double* hist) {
  MockSolverFunction<T> fn(P);
  if (enabled) {
    fn.enabled.assign(enabled, enabled + P);
  }
  const Options opt = makeOptions(o, useBlockJtJ);
  SolveResult<T> r = solveGaussNewton<T>(fn, opt, theta);
  if (err) {
    *err = r.error;
  }
  if (iters) {
    *iters = r.iterations;
  }
  if (hist) {
    for (int i = 0; i < opt.maxIterations; ++i) {
      hist[i] = i < r.iterations ? r.errorHistory[i] : 0.0;
    }
  }
  return 0;
}

} // namespace

extern "C" {

#define ORC_INSTANTIATE(SUF, T)                                                                          \
  int orc_skeleton_state_##SUF(                                                                          \
      const mmx_rig_desc* d, const T* theta, T* world, T* local, T* transAxis, T* rotAxis, T* jp) {      \
    return skeletonState<T>(d, theta, world, local, transAxis, rotAxis, jp);                             \
  }                                                                                                      \
  int orc_eval_jacobian_##SUF(                                                                           \
      const mmx_rig_desc* d,                                                                             \
      int Kp,                                                                                            \
      const int32_t* pp,                                                                                 \
      int Ko,                                                                                            \
      const int32_t* op,                                                                                 \
      const mmx_constraint_data* c,                                                                      \
      const uint8_t* en,                                                                                 \
      const T* theta,                                                                                    \
      T* jac,                                                                                            \
      T* res,                                                                                            \
      double* err) {                                                                                     \
    return evalJacobian<T>(d, Kp, pp, Ko, op, c, en, theta, jac, res, err);                              \
  }                                                                                                      \
  int orc_get_error_##SUF(                                                                               \
      const mmx_rig_desc* d,                                                                             \
      int Kp,                                                                                            \
      const int32_t* pp,                                                                                 \
      int Ko,                                                                                            \
      const int32_t* op,                                                                                 \
      const mmx_constraint_data* c,                                                                      \
      const T* theta,                                                                                    \
      double* err) {                                                                                     \
    return getError<T>(d, Kp, pp, Ko, op, c, theta, err);                                                \
  }                                                                                                      \
  int orc_solve_##SUF(                                                                                   \
      const mmx_rig_desc* d,                                                                             \
      int Kp,                                                                                            \
      const int32_t* pp,                                                                                 \
      int Ko,                                                                                            \
      const int32_t* op,                                                                                 \
      const mmx_constraint_data* c,                                                                      \
      const uint8_t* en,                                                                                 \
      const mmx_gn_options* o,                                                                           \
      int useBlockJtJ,                                                                                   \
      T* theta,                                                                                          \
      double* err,                                                                                       \
      int32_t* iters,                                                                                    \
      int32_t* status,                                                                                   \
      double* hist,                                                                                      \
      T* jtj,                                                                                            \
      T* jtr) {                                                                                          \
    const Rig rig = makeRig(d);                                                                          \
    return solveOne<T>(                                                                                  \
        rig,                                                                                             \
        makeConstraints<T>(Kp, pp, Ko, op, c, 0, d->num_params),                                                        \
        en,                                                                                              \
        makeOptions(o, useBlockJtJ),                                                                     \
        theta,                                                                                           \
        err,                                                                                             \
        iters,                                                                                           \
        status,                                                                                          \
        hist,                                                                                            \
        jtj,                                                                                             \
        jtr);                                                                                            \
  }                                                                                                      \
  int orc_solve_batch_##SUF(                                                                             \
      const mmx_rig_desc* d,                                                                             \
      int B,                                                                                             \
      int Kp,                                                                                            \
      const int32_t* pp,                                                                                 \
      int Ko,                                                                                            \
      const int32_t* op,                                                                                 \
      const mmx_constraint_data* c,                                                                      \
      const uint8_t* en,                                                                                 \
      const mmx_gn_options* o,                                                                           \
      int useBlockJtJ,                                                                                   \
      T* theta,                                                                                          \
      double* err,                                                                                       \
      int32_t* iters,                                                                                    \
      int32_t* status,                                                                                   \
      double* hist,                                                                                      \
      int nthreads) {                                                                                    \
    return solveBatch<T>(d, B, Kp, pp, Ko, op, c, en, o, useBlockJtJ, theta, err, iters, status, hist, nthreads); \
  }                                                                                                      \
  /* orc_solve_batch + the LM schedule's per-iteration damping and gain ratio ([B][maxIterations] doubles) */ \
  int orc_solve_batch_steps_##SUF(                                                                       \
      const mmx_rig_desc* d,                                                                             \
      int B,                                                                                             \
      int Kp,                                                                                            \
      const int32_t* pp,                                                                                 \
      int Ko,                                                                                            \
      const int32_t* op,                                                                                 \
      const mmx_constraint_data* c,                                                                      \
      const uint8_t* en,                                                                                 \
      const mmx_gn_options* o,                                                                           \
      int useBlockJtJ,                                                                                   \
      T* theta,                                                                                          \
      double* err,                                                                                       \
      int32_t* iters,                                                                                    \
      int32_t* status,                                                                                   \
      double* hist,                                                                                      \
      int nthreads,                                                                                      \
      double* lamHist,                                                                                   \
      double* rhoHist) {                                                                                 \
    return solveBatch<T>(d, B, Kp, pp, Ko, op, c, en, o, useBlockJtJ, theta, err, iters, status, hist, nthreads, lamHist, rhoHist); \
  }                                                                                                      \
  int orc_mock_solve_##SUF(                                                                              \
      int P, const uint8_t* en, const mmx_gn_options* o, int useBlockJtJ, T* theta, double* err, int32_t* iters, double* hist) { \
    return mockSolve<T>(P, en, o, useBlockJtJ, theta, err, iters, hist);                                 \
  }

ORC_INSTANTIATE(f32, float)
ORC_INSTANTIATE(f64, double)

// parent-chasing ancestor test, the reference's own bookkeeping (joint_error_function-inl.h:228,293):
// out[J*J], out[a*J + j] = 1 iff a is j or an ancestor of j.
int orc_ancestor_matrix(const mmx_rig_desc* d, uint8_t* out) {
  const int J = d->num_joints;
  std::fill(out, out + size_t(J) * J, uint8_t(0));
  for (int j = 0; j < J; ++j) {
    int a = j;
    while (a >= 0) {
      out[size_t(a) * J + j] = 1;
      a = d->parent[a];
    }
  }
  return 0;
}

int orc_active_joint_params(const mmx_rig_desc* d, const uint8_t* enabled, uint8_t* active) {
  const Rig rig = makeRig(d);
  computeActiveJointParams(rig, enabled, active);
  return 0;
}

#ifdef ORC_PHASE_TIMERS
// cycles per phase accumulated by THIS thread's solves since the last call (single-threaded diagnostic runs); clears them
int orc_phase_cycles(unsigned long long out[8]) {
  unsigned long long* c = orcPhaseCycles();
  for (int i = 0; i < 8; ++i) {
    out[i] = c[i];
    c[i] = 0;
  }
  return 0;
}
#endif

int orc_hardware_threads(void) {
  return int(std::thread::hardware_concurrency());
}

} // extern "C"
