// tensor_ik_mmx_adapter.cpp -- the reference-side binding of INTEGRATION.md section 2 as a translation unit: what a
// momentum maintainer adds next to solveTensorIKProblem (pymomentum/tensor_ik/tensor_ik.cpp:95-188) to send a batch whose
// characters share one topology through include/mmx.h.  Built in this repository's tests against integration/momentum_stub.h
// (tests/test_cpp_shell.py), so that a signature drift in mmx.h breaks a test instead of a document; built against momentum
// with -DMMX_ADAPTER_WITH_MOMENTUM.
#include "tensor_ik_mmx_adapter.h"

#include <cstring>

namespace mmx_adapter {

// momentum::Character -> mmx_rig_desc over host arrays the RigArrays object owns (mmx_rig_create copies them)
RigArrays makeRigDesc(const momentum::Character& c) {
  RigArrays a;
  const auto& joints = c.skeleton.joints;
  const auto& pt = c.parameterTransform; // SparseRowMatrix<float> == CSR: the three arrays go over as they are
  a.parent.resize(joints.size());
  a.pre.resize(4 * joints.size());
  a.off.resize(3 * joints.size());
  for (size_t j = 0; j < joints.size(); ++j) {
    a.parent[j] = joints[j].parent == momentum::kInvalidIndex ? MMX_INVALID_PARENT : int32_t(joints[j].parent);
    std::memcpy(&a.pre[4 * j], joints[j].preRotation.coeffs().data(), 4 * sizeof(float)); // (x, y, z, w)
    std::memcpy(&a.off[3 * j], joints[j].translationOffset.data(), 3 * sizeof(float));
  }
  a.desc = mmx_rig_desc{};
  a.desc.num_joints = int32_t(joints.size());
  a.desc.num_params = int32_t(pt.numAllModelParameters());
  a.desc.parent = a.parent.data();
  a.desc.pre_rotation = a.pre.data();
  a.desc.translation_offset = a.off.data();
  a.desc.pt_outer = pt.transform.outerIndexPtr();
  a.desc.pt_inner = pt.transform.innerIndexPtr();
  a.desc.pt_value = pt.transform.valuePtr();
  a.desc.pt_offsets = pt.offsets.size() > 0 ? pt.offsets.data() : nullptr;
  return a;
}

mmx_rig* makeRig(const momentum::Character& c, int device) {
  MT_THROW_IF(mmx_abi_version() != MMX_ABI_VERSION, "libmmx_hip.so was built for another version of mmx.h");
  const RigArrays a = makeRigDesc(c);
  mmx_rig* rig = nullptr;
  MT_THROW_IF(mmx_rig_create(&a.desc, device, &rig) != MMX_OK, mmx_last_error());
  return rig;
}

namespace {

// problem handle + constraints + options of one batch (what both scalar types share)
struct Prepared {
  mmx_problem* pb = nullptr;
  mmx_gn_options o{};
  std::vector<float> fw;
  ~Prepared() { mmx_problem_destroy(pb); }
};

void prepare(Prepared& p, mmx_rig* rig, const momentum::ParameterSet& activeParams, const BatchTensors& t, const pymomentum::SolverOptions& options) {
  MT_THROW_IF(mmx_problem_create(rig, int32_t(t.nBatch), t.numPositions, t.positionParents, t.numOrientations, t.orientationParents, &p.pb) != MMX_OK, mmx_last_error());
  mmx_problem* pb = p.pb;
  const int32_t P = mmx_rig_num_params(rig);
  std::vector<uint8_t> enabled(size_t(P), 0);
  for (int32_t q = 0; q < P; ++q) {
    enabled[size_t(q)] = activeParams.test(size_t(q)) ? 1 : 0;
  }
  MT_THROW_IF(mmx_problem_set_enabled(pb, enabled.data()) != MMX_OK, mmx_last_error());
  if (t.perElementTranslationOffsets != nullptr || t.perElementPreRotations != nullptr) { // characters[iBatch] (:129,140)
    MT_THROW_IF(mmx_problem_set_instance_rig(pb, t.perElementTranslationOffsets, t.perElementPreRotations, MMX_MEM_HOST, nullptr) != MMX_OK, mmx_last_error());
  }
  if (t.perElementPositionParents != nullptr || t.perElementOrientationParents != nullptr) { // parents per element
    MT_THROW_IF(mmx_problem_set_instance_parents(pb, t.perElementPositionParents, t.perElementOrientationParents, MMX_MEM_HOST, nullptr) != MMX_OK, mmx_last_error());
  }
  mmx_constraint_data cd{}; // zero = every optional block absent
  cd.pos_offset = t.positionOffsets, cd.pos_target = t.positionTargets, cd.pos_weight = t.positionWeights;
  cd.ori_offset = t.orientationOffsets, cd.ori_target = t.orientationTargets, cd.ori_weight = t.orientationWeights;
  cd.pos_function_weight = cd.ori_function_weight = 1.f; // the per-element weights below carry setWeight()
  cd.memory = MMX_MEM_HOST;
  // errorFunctionWeights [nBatch][numWeightColumns] + weightsMap (tensor_ik.cpp:100-101): one column per block in the ABI's
  // order (position, orientation); weightsMap[iErr] < 0 means weight 0 (tensor_ik_utility.cpp:176)
  p.fw.assign(size_t(t.nBatch) * 2, 1.f);
  if (t.errorFunctionWeights != nullptr) {
    for (int64_t b = 0; b < t.nBatch; ++b) {
      for (int k = 0; k < 2; ++k) {
        const int col = t.weightsMap[k];
        p.fw[size_t(2 * b + k)] = col < 0 ? 0.f : t.errorFunctionWeights[size_t(b) * size_t(t.numWeightColumns) + size_t(col)];
      }
    }
    cd.function_weights = p.fw.data();
    cd.num_function_weights = 2;
  }
  MT_THROW_IF(mmx_problem_set_constraints_sized(pb, &cd, sizeof(cd), nullptr) != MMX_OK, mmx_last_error());
  mmx_gn_options_default(&p.o);
  p.o.min_iterations = int32_t(options.minIter);
  p.o.max_iterations = int32_t(options.maxIter);
  p.o.threshold = options.threshold;
  p.o.regularization = options.levmar_lambda;
  // SubsetGaussNewtonSolver / GaussNewtonSolverQR share one backtracking rule (tensor_ik.cpp:142-158)
  p.o.do_line_search = options.lineSearch ? MMX_LINE_SEARCH_DIRECTIONAL : MMX_LINE_SEARCH_NONE;
  if (options.linearSolverType == pymomentum::LinearSolverType::TrustRegionQR) {
    p.o.step_rule = MMX_STEP_TRUST_REGION;
  }
}

SolveReport report(const BatchTensors& t, const std::vector<int32_t>& status) {
  SolveReport r;
  for (int64_t b = 0; b < t.nBatch; ++b) {
    const int32_t s = status[size_t(b)];
    r.precisionSuspect += (s & MMX_SOLVE_PRECISION_SUSPECT) != 0;
    r.mixed += (s & MMX_SOLVE_MIXED) != 0;
    r.escalatedF64 += (s & MMX_SOLVE_ESCALATED_F64) != 0;
    r.dampingFloored += (s & MMX_SOLVE_DAMPING_FLOORED) != 0;
    r.failed += MMX_SOLVE_FAILED(s) ? 1 : 0;
    if (t.dampingFloored != nullptr) { // informational: elements whose damping sat below the single-precision factor's floor
      t.dampingFloored[b] = (s & MMX_SOLVE_DAMPING_FLOORED) != 0;
    }
    if (t.status != nullptr) {
      t.status[b] = s;
    }
  }
  return r;
}

} // namespace

// the body of solveTensorIKProblem<float> when the GPU path applies: what the dispenso::parallel_for over the batch
// elements (tensor_ik.cpp:127-177) does, for all elements at once.  modelParameters: [nBatch][P], initial values in,
// solution out (an element whose result is not finite keeps its initial values, :168-173).
SolveReport solveBatch(
    mmx_rig* rig,
    const momentum::ParameterSet& activeParams,
    const BatchTensors& t,
    const pymomentum::SolverOptions& options,
    float* modelParameters,
    int32_t precision) {
  Prepared p;
  prepare(p, rig, activeParams, t, options);
  p.o.precision = precision; // (ABI 10 / 11; precision_bound keeps its default: north_star's 1e-5)
  std::vector<int32_t> status(size_t(t.nBatch), 0);
  MT_THROW_IF(mmx_solve_host(p.pb, &p.o, modelParameters, nullptr, nullptr, status.data()) != MMX_OK, mmx_last_error());
  return report(t, status);
}

// solveTensorIKProblem<double>: the double instantiation on double parameters
SolveReport solveBatch(
    mmx_rig* rig,
    const momentum::ParameterSet& activeParams,
    const BatchTensors& t,
    const pymomentum::SolverOptions& options,
    double* modelParameters) {
  Prepared p;
  prepare(p, rig, activeParams, t, options);
  std::vector<int32_t> status(size_t(t.nBatch), 0);
  MT_THROW_IF(mmx_solve_f64_host(p.pb, &p.o, modelParameters, nullptr, nullptr, status.data()) != MMX_OK, mmx_last_error());
  return report(t, status);
}

} // namespace mmx_adapter
