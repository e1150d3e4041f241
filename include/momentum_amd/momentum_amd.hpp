// momentum_amd.hpp -- header-only C++17 shell over the C ABI (include/mmx.h) that keeps momentum's
// Character / Skeleton / SkeletonSolverFunction / GaussNewtonSolver surface for the batched IK
// path, so that code written against momentum's classes switches to the MI355X path by changing
// the namespace and the solver type.  Host-side glue only; all compute is in libmmx_hip.so.
//
// Mirrors (reference file:line):
//   Joint / Skeleton                 momentum/character/joint.h:18-36, skeleton.h:22-25
//   ParameterTransform               momentum/character/parameter_transform.h:62-95 (CSR = SparseRowMatrix)
//   Character                        momentum/character/character.h:32-125 (skeleton + parameterTransform only)
//   PositionData / OrientationData   momentum/character_solver/position_error_function.h:16-29,
//                                    orientation_error_function.h:16-36
//   SolverOptions                    momentum/solver/solver.h:19-34
//   GaussNewtonSolverOptions         momentum/solver/gauss_newton_solver.h:17-59
//   SkeletonSolverFunction           momentum/character_solver/skeleton_solver_function.h:21-95
//   GaussNewtonSolver::solve         momentum/solver/solver.cpp:50-128 (one per batch element, as
//                                    pymomentum/tensor_ik/tensor_ik.cpp:127-177 does)
// Errors: every non-zero status becomes std::runtime_error with the library's message, the
// OSS behaviour of MT_CHECK / MT_THROW (momentum/common/checks.h:36-45, exception.h:24-66).
#pragma once

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../mmx.h"

namespace momentum_amd {

inline constexpr size_t kInvalidIndex = std::numeric_limits<size_t>::max(); // character/types.h:182
inline constexpr size_t kParametersPerJoint = 7; // character/types.h:21

using Vector3f = std::array<float, 3>;
using Quaternionf = std::array<float, 4>; // (x, y, z, w), Eigen storage order
using ParameterSet = std::vector<bool>; // momentum: std::bitset<kMaxModelParams>

inline void check(int32_t rc) {
  if (rc != MMX_OK) {
    throw std::runtime_error(std::string("momentum_amd: ") + mmx_last_error());
  }
}

struct Joint {
  std::string name;
  size_t parent = kInvalidIndex;
  Quaternionf preRotation{0.f, 0.f, 0.f, 1.f};
  Vector3f translationOffset{0.f, 0.f, 0.f};
};

struct Skeleton {
  std::vector<Joint> joints;
};

// Row-major sparse 7J x P transform + offsets.  `addEntry` keeps rows sorted by column.
struct ParameterTransform {
  std::vector<std::string> name; // model parameter names; size = numAllModelParameters()
  std::vector<int32_t> outer{0}, inner;
  std::vector<float> value;
  std::vector<float> offsets;

  size_t numAllModelParameters() const {
    return name.size();
  }
  // build from (row, col, value) triplets like Eigen's setFromTriplets
  struct Triplet {
    int32_t row, col;
    float value;
  };
  void setFromTriplets(size_t numJoints, const std::vector<Triplet>& triplets) {
    const size_t rows = kParametersPerJoint * numJoints;
    std::vector<std::vector<std::pair<int32_t, float>>> r(rows);
    for (const auto& t : triplets) {
      if (t.row < 0 || size_t(t.row) >= rows || t.col < 0 || size_t(t.col) >= name.size()) {
        throw std::runtime_error("ParameterTransform triplet out of range");
      }
      bool merged = false;
      for (auto& e : r[t.row]) {
        if (e.first == t.col) {
          e.second += t.value;
          merged = true;
        }
      }
      if (!merged) {
        r[t.row].push_back({t.col, t.value});
      }
    }
    outer.assign(1, 0);
    inner.clear();
    value.clear();
    for (auto& row : r) {
      for (size_t i = 1; i < row.size(); ++i) { // insertion sort by column
        for (size_t j = i; j > 0 && row[j].first < row[j - 1].first; --j) {
          std::swap(row[j], row[j - 1]);
        }
      }
      for (const auto& e : row) {
        inner.push_back(e.first);
        value.push_back(e.second);
      }
      outer.push_back(int32_t(inner.size()));
    }
    offsets.assign(rows, 0.f);
  }
};

struct Character {
  Skeleton skeleton;
  ParameterTransform parameterTransform;
};

struct PositionData {
  Vector3f offset{0.f, 0.f, 0.f};
  Vector3f target{0.f, 0.f, 0.f};
  size_t parent = 0;
  float weight = 1.f;
};

// momentum::ParameterLimit restricted to the limit types that act on model parameters
// (momentum/character/parameter_limits.h:20-31,33-99,125-136)
enum LimitType {
  MinMax = MMX_LIMIT_MINMAX,
  MinMaxJoint = MMX_LIMIT_MINMAX_JOINT, // index0 = 7 * jointIndex + jointParameter
  Linear = MMX_LIMIT_LINEAR,
  LinearJoint = MMX_LIMIT_LINEAR_JOINT,
  HalfPlane = MMX_LIMIT_HALFPLANE
};
struct ParameterLimit {
  LimitType type = MinMax;
  float weight = 1.f;
  // MinMax: parameterIndex, limits = {min, max}
  // Linear: referenceIndex, targetIndex, {scale, offset, rangeMin, rangeMax}
  // HalfPlane: param1, param2, {normal[0], normal[1], offset}
  size_t index0 = 0, index1 = 0;
  std::array<float, 4> values{0.f, 0.f, 0.f, 0.f};
  static ParameterLimit minMax(size_t parameterIndex, float lo, float hi, float weight = 1.f) {
    return ParameterLimit{MinMax, weight, parameterIndex, 0, {lo, hi, 0.f, 0.f}};
  }
  static ParameterLimit
  linear(size_t referenceIndex, size_t targetIndex, float scale, float offset, float rangeMin = 0.f, float rangeMax = 0.f, float weight = 1.f) {
    return ParameterLimit{Linear, weight, referenceIndex, targetIndex, {scale, offset, rangeMin, rangeMax}};
  }
  static ParameterLimit halfPlane(size_t param1, size_t param2, float n0, float n1, float offset, float weight = 1.f) {
    return ParameterLimit{HalfPlane, weight, param1, param2, {n0, n1, offset, 0.f}};
  }
};
using ParameterLimits = std::vector<ParameterLimit>;

struct OrientationData {
  Quaternionf offset{0.f, 0.f, 0.f, 1.f};
  Quaternionf target{0.f, 0.f, 0.f, 1.f};
  size_t parent = 0;
  float weight = 1.f;
};

struct SolverOptions {
  size_t minIterations = 1;
  size_t maxIterations = 2;
  float threshold = 1.0f;
  bool verbose = false;
  virtual ~SolverOptions() = default;
};

// gauss_newton_solver.h:17-33
struct GaussNewtonSolverBaseOptions : SolverOptions {
  float regularization = 0.05f;
  bool doLineSearch = false;
  GaussNewtonSolverBaseOptions() = default;
  /* implicit */ GaussNewtonSolverBaseOptions(const SolverOptions& base) : SolverOptions(base) {}
};

struct GaussNewtonSolverOptions : GaussNewtonSolverBaseOptions {
  bool useBlockJtJ = false; // accepted for source compatibility: both settings build the same system
  GaussNewtonSolverOptions() = default;
  explicit GaussNewtonSolverOptions(const SolverOptions& base) : GaussNewtonSolverBaseOptions(base) {}
};

// subset_gauss_newton_solver.h:19-26 and gauss_newton_solver_qr.h:20-25: the two solvers the batched
// driver builds (pymomentum/tensor_ik/tensor_ik.cpp:142-158).  Same normal equations as
// GaussNewtonSolverT; their line search tests against the directional derivative
// (MMX_LINE_SEARCH_DIRECTIONAL).
struct SubsetGaussNewtonSolverOptions : GaussNewtonSolverBaseOptions {
  SubsetGaussNewtonSolverOptions() = default;
  /* implicit */ SubsetGaussNewtonSolverOptions(const SolverOptions& base) : GaussNewtonSolverBaseOptions(base) {}
};
struct GaussNewtonSolverQROptions : GaussNewtonSolverBaseOptions {
  GaussNewtonSolverQROptions() = default;
  /* implicit */ GaussNewtonSolverQROptions(const SolverOptions& base) : GaussNewtonSolverBaseOptions(base) {}
};
// trust_region_qr.h:22-33: the third solver the batched driver builds (tensor_ik.cpp:150-152)
struct TrustRegionQROptions : SolverOptions {
  float trustRegionRadius_ = 1.0f;
  TrustRegionQROptions() = default;
  /* implicit */ TrustRegionQROptions(const SolverOptions& base) : SolverOptions(base) {}
};

// Device-resident Skeleton + ParameterTransform.
class DeviceCharacter {
 public:
  explicit DeviceCharacter(const Character& c, int device = 0) : numJoints_(c.skeleton.joints.size()) {
    const size_t J = numJoints_;
    std::vector<int32_t> parent(J);
    std::vector<float> pre(4 * J), off(3 * J);
    for (size_t j = 0; j < J; ++j) {
      const Joint& jt = c.skeleton.joints[j];
      parent[j] = jt.parent == kInvalidIndex ? MMX_INVALID_PARENT : int32_t(jt.parent);
      for (int k = 0; k < 4; ++k) {
        pre[4 * j + k] = jt.preRotation[k];
      }
      for (int k = 0; k < 3; ++k) {
        off[3 * j + k] = jt.translationOffset[k];
      }
    }
    const ParameterTransform& pt = c.parameterTransform;
    if (pt.outer.size() != kParametersPerJoint * J + 1) {
      throw std::runtime_error("momentum_amd: parameter transform has the wrong number of rows");
    }
    mmx_rig_desc d{};
    d.num_joints = int32_t(J);
    d.num_params = int32_t(pt.numAllModelParameters());
    d.parent = parent.data();
    d.pre_rotation = pre.data();
    d.translation_offset = off.data();
    d.pt_outer = pt.outer.data();
    d.pt_inner = pt.inner.data();
    d.pt_value = pt.value.data();
    d.pt_offsets = pt.offsets.empty() ? nullptr : pt.offsets.data();
    numParams_ = pt.numAllModelParameters();
    parents_ = parent;
    mmx_rig* h = nullptr;
    check(mmx_rig_create(&d, device, &h));
    handle_.reset(h, [](mmx_rig* p) { mmx_rig_destroy(p); });
  }
  mmx_rig* handle() const {
    return handle_.get();
  }
  size_t numParameters() const {
    return numParams_;
  }
  size_t numJoints() const {
    return numJoints_;
  }
  int32_t parentOf(size_t joint) const { // MMX_INVALID_PARENT for a root
    return parents_[joint];
  }

 private:
  std::shared_ptr<mmx_rig> handle_;
  size_t numJoints_ = 0, numParams_ = 0;
  std::vector<int32_t> parents_;
};

// Constraint data of the further JointErrorFunction specialisations; vectors are normalised where
// the reference's constructors do (plane_error_function.h:16-31, aim_error_function.h:17-37,
// fixed_axis_error_function.h:17-31, normal_error_function.h:17-36).
struct PlaneData {
  Vector3f offset{0.f, 0.f, 0.f};
  Vector3f normal{0.f, 1.f, 0.f};
  float d = 0.f;
  size_t parent = 0;
  float weight = 1.f;
};
struct AimData {
  Vector3f localPoint{0.f, 0.f, 0.f};
  Vector3f localDir{0.f, 0.f, 1.f};
  Vector3f globalTarget{0.f, 0.f, 0.f};
  size_t parent = 0;
  float weight = 1.f;
};
struct FixedAxisData {
  Vector3f localAxis{0.f, 0.f, 1.f};
  Vector3f globalAxis{0.f, 0.f, 1.f};
  size_t parent = 0;
  float weight = 1.f;
};
struct NormalData {
  Vector3f localPoint{0.f, 0.f, 0.f};
  Vector3f localNormal{0.f, 0.f, 1.f};
  Vector3f globalPoint{0.f, 0.f, 0.f};
  size_t parent = 0;
  float weight = 1.f;
};
enum class JointErrorFunctionType {
  Plane = MMX_JC_PLANE, // PlaneErrorFunction(above = false)
  HalfPlane = MMX_JC_HALF_PLANE, // PlaneErrorFunction(above = true)
  AimDist = MMX_JC_AIM_DIST,
  AimDir = MMX_JC_AIM_DIR,
  FixedAxisDiff = MMX_JC_FIXED_AXIS_DIFF,
  FixedAxisCos = MMX_JC_FIXED_AXIS_COS,
  FixedAxisAngle = MMX_JC_FIXED_AXIS_ANGLE,
  Normal = MMX_JC_NORMAL
};

// One SkeletonSolverFunction + PositionErrorFunction + OrientationErrorFunction per batch element.
// The parent lists of the constructor are the default of every element; an element whose constraints name
// other parents (ConstraintData::parent) gets its own list, and setCharacters() gives every element its own
// Character of the same topology -- what solveTensorIKProblem does with characters[iBatch]
// (pymomentum/tensor_ik/tensor_ik.cpp:129-141).
class BatchedSkeletonSolverFunction {
 public:
  BatchedSkeletonSolverFunction(
      const DeviceCharacter& character,
      size_t batch,
      const std::vector<size_t>& positionParents,
      const std::vector<size_t>& orientationParents)
      : character_(character), batch_(batch), kp_(positionParents.size()), ko_(orientationParents.size()) {
    std::vector<int32_t> pp(positionParents.begin(), positionParents.end());
    std::vector<int32_t> op(orientationParents.begin(), orientationParents.end());
    mmx_problem* h = nullptr;
    check(mmx_problem_create(
        character.handle(), int32_t(batch), int32_t(kp_), pp.data(), int32_t(ko_), op.data(), &h));
    handle_.reset(h, [](mmx_problem* p) { mmx_problem_destroy(p); });
    posOffset_.assign(batch * kp_ * 3, 0.f);
    posTarget_.assign(batch * kp_ * 3, 0.f);
    posWeight_.assign(batch * kp_, 1.f);
    oriOffset_.assign(batch * ko_ * 4, 0.f);
    oriTarget_.assign(batch * ko_ * 4, 0.f);
    for (size_t i = 0; i < batch * ko_; ++i) {
      oriOffset_[4 * i + 3] = 1.f;
      oriTarget_[4 * i + 3] = 1.f;
    }
    oriWeight_.assign(batch * ko_, 1.f);
    posParent_.resize(batch * kp_);
    oriParent_.resize(batch * ko_);
    for (size_t b = 0; b < batch; ++b) {
      std::copy(pp.begin(), pp.end(), posParent_.begin() + b * kp_);
      std::copy(op.begin(), op.end(), oriParent_.begin() + b * ko_);
    }
    sharedPos_ = pp;
    sharedOri_ = op;
  }
  BatchedSkeletonSolverFunction(const BatchedSkeletonSolverFunction&) = delete; // like the reference (:29-32)
  BatchedSkeletonSolverFunction& operator=(const BatchedSkeletonSolverFunction&) = delete;

  size_t getNumParameters() const {
    return character_.numParameters();
  }
  size_t batchSize() const {
    return batch_;
  }
  // One Character per batch element, all of the topology of the DeviceCharacter (same parents and
  // parameter transform; translationOffset / preRotation differ, e.g. per-subject bone lengths).  An
  // empty list goes back to the shared character.
  void setCharacters(const std::vector<const Character*>& characters) {
    if (characters.empty()) {
      check(mmx_problem_set_instance_rig(handle_.get(), nullptr, nullptr, MMX_MEM_HOST, nullptr));
      return;
    }
    const size_t J = character_.numJoints();
    if (characters.size() != batch_) {
      throw std::runtime_error("momentum_amd: one character per batch element expected");
    }
    std::vector<float> off(batch_ * J * 3), pre(batch_ * J * 4);
    for (size_t b = 0; b < batch_; ++b) {
      const Character& c = *characters[b];
      if (c.skeleton.joints.size() != J || c.parameterTransform.numAllModelParameters() != character_.numParameters()) {
        throw std::runtime_error("momentum_amd: per-element characters must share the topology of the device character");
      }
      for (size_t j = 0; j < J; ++j) {
        const Joint& jt = c.skeleton.joints[j];
        const int32_t par = jt.parent == kInvalidIndex ? MMX_INVALID_PARENT : int32_t(jt.parent);
        if (par != character_.parentOf(j)) {
          throw std::runtime_error("momentum_amd: per-element characters must share the topology of the device character");
        }
        for (int k = 0; k < 3; ++k) {
          off[(b * J + j) * 3 + k] = jt.translationOffset[k];
        }
        for (int k = 0; k < 4; ++k) {
          pre[(b * J + j) * 4 + k] = jt.preRotation[k];
        }
      }
    }
    check(mmx_problem_set_instance_rig(handle_.get(), off.data(), pre.data(), MMX_MEM_HOST, nullptr));
  }
  // PositionErrorFunction::setConstraints of batch element b (ConstraintData::parent per constraint)
  void setPositionConstraints(size_t b, const std::vector<PositionData>& c) {
    if (b >= batch_ || c.size() != kp_) {
      throw std::runtime_error("momentum_amd: position constraint count / batch index mismatch");
    }
    for (size_t i = 0; i < kp_; ++i) {
      posParent_[b * kp_ + i] = int32_t(c[i].parent);
      for (int k = 0; k < 3; ++k) {
        posOffset_[(b * kp_ + i) * 3 + k] = c[i].offset[k];
        posTarget_[(b * kp_ + i) * 3 + k] = c[i].target[k];
      }
      posWeight_[b * kp_ + i] = c[i].weight;
    }
    dirty_ = true;
  }
  void setOrientationConstraints(size_t b, const std::vector<OrientationData>& c) {
    if (b >= batch_ || c.size() != ko_) {
      throw std::runtime_error("momentum_amd: orientation constraint count / batch index mismatch");
    }
    for (size_t i = 0; i < ko_; ++i) {
      oriParent_[b * ko_ + i] = int32_t(c[i].parent);
      for (int k = 0; k < 4; ++k) {
        oriOffset_[(b * ko_ + i) * 4 + k] = c[i].offset[k];
        oriTarget_[(b * ko_ + i) * 4 + k] = c[i].target[k];
      }
      oriWeight_[b * ko_ + i] = c[i].weight;
    }
    dirty_ = true;
  }
  void setWeights(float positionWeight, float orientationWeight) { // SkeletonErrorFunction::setWeight
    wPos_ = positionWeight;
    wOri_ = orientationWeight;
    dirty_ = true;
  }
  // Per-element error-function weights: errorFunctionWeights[iBatch][weightsMap[iErr]] of solveTensorIKProblem
  // (pymomentum/tensor_ik/tensor_ik.cpp:100-101,137-138; every error function of element iBatch gets setWeight(entry),
  // tensor_ik_utility.cpp:162-177).  weights: [batch][columns] row-major, columns = position, orientation, limits,
  // model parameters, joint block 0, 1, ... (mmx_constraint_data::function_weights); an empty vector removes them.
  void setErrorFunctionWeights(const std::vector<float>& weights, size_t columns) {
    if (!weights.empty() && (columns == 0 || weights.size() != batch_ * columns)) {
      throw std::runtime_error("momentum_amd: error-function weights must be [batch][columns]");
    }
    fnWeights_ = weights;
    fnCols_ = weights.empty() ? 0 : columns;
    dirty_ = true;
  }
  // LimitErrorFunctionT::setLimits (limit_error_function.h:88) for every element, with its weight_
  void setLimits(const ParameterLimits& limits, float weight = 1.f) {
    limits_.clear();
    for (const ParameterLimit& l : limits) {
      mmx_parameter_limit m{};
      m.type = int32_t(l.type);
      m.index0 = int32_t(l.index0);
      m.index1 = int32_t(l.index1);
      m.weight = l.weight;
      for (int k = 0; k < 4; ++k) {
        m.v[k] = l.values[size_t(k)];
      }
      limits_.push_back(m);
    }
    wLimit_ = weight;
    dirty_ = true;
  }
  // ModelParametersErrorFunctionT::setTargetParameters (model_parameters_error_function.h:48-51) of element b
  void setTargetParameters(size_t b, const std::vector<float>& params, const std::vector<float>& weights, float weight = 1.f) {
    const size_t P = getNumParameters();
    if (b >= batch_ || params.size() != P || weights.size() != P) {
      throw std::runtime_error("momentum_amd: target parameter count / batch index mismatch");
    }
    if (mpTarget_.empty()) {
      mpTarget_.assign(batch_ * P, 0.f);
      mpWeights_.assign(batch_ * P, 0.f);
    }
    std::copy(params.begin(), params.end(), mpTarget_.begin() + b * P);
    std::copy(weights.begin(), weights.end(), mpWeights_.begin() + b * P);
    wModel_ = weight;
    dirty_ = true;
  }
  // GeneralizedLoss(alpha, c) of the position / orientation blocks: the lossAlpha / lossC constructor
  // arguments of PositionErrorFunctionT / OrientationErrorFunctionT (position_error_function.h:41-48)
  void setLoss(float positionAlpha, float positionC, float orientationAlpha, float orientationC) {
    lossPos_[0] = positionAlpha, lossPos_[1] = positionC;
    lossOri_[0] = orientationAlpha, lossOri_[1] = orientationC;
    dirty_ = true;
  }
  // SkeletonSolverFunction::addErrorFunction for one of the further joint error functions
  // (Plane / Aim / FixedAxis / Normal); parents are shared by the batch.  Returns the block's index.
  size_t addJointErrorFunction(JointErrorFunctionType type, const std::vector<size_t>& parents, float lossAlpha = 2.f, float lossC = 1.f) {
    if (blocks_.size() >= MMX_MAX_JOINT_BLOCKS) {
      throw std::runtime_error("momentum_amd: too many joint error functions");
    }
    Block k;
    k.type = int32_t(type);
    k.parent.assign(parents.begin(), parents.end());
    const size_t cnt = batch_ * parents.size();
    k.localPoint.assign(3 * cnt, 0.f);
    k.localDir.assign(3 * cnt, 0.f);
    k.global.assign(3 * cnt, 0.f);
    k.planeD.assign(cnt, 0.f);
    k.weight.assign(cnt, 0.f); // constraints not set yet carry weight 0 (skipped, joint_error_function-inl.h:197-199)
    k.loss[0] = lossAlpha, k.loss[1] = lossC;
    blocks_.push_back(std::move(k));
    dirty_ = true;
    return blocks_.size() - 1;
  }
  void setWeight(size_t block, float weight) { // SkeletonErrorFunction::setWeight of that error function
    blockAt(block).fw = weight;
    dirty_ = true;
  }
  void setConstraints(size_t block, size_t b, const std::vector<PlaneData>& c) {
    Block& k = expect(block, b, c.size(), MMX_JC_PLANE, MMX_JC_HALF_PLANE);
    for (size_t i = 0; i < c.size(); ++i) {
      put(k, b, i, c[i].offset.data(), nullptr, c[i].normal.data(), c[i].d, c[i].weight);
    }
  }
  void setConstraints(size_t block, size_t b, const std::vector<AimData>& c) {
    Block& k = expect(block, b, c.size(), MMX_JC_AIM_DIST, MMX_JC_AIM_DIR);
    for (size_t i = 0; i < c.size(); ++i) {
      put(k, b, i, c[i].localPoint.data(), c[i].localDir.data(), c[i].globalTarget.data(), 0.f, c[i].weight);
    }
  }
  void setConstraints(size_t block, size_t b, const std::vector<FixedAxisData>& c) {
    Block& k = expect(block, b, c.size(), MMX_JC_FIXED_AXIS_DIFF, MMX_JC_FIXED_AXIS_ANGLE);
    for (size_t i = 0; i < c.size(); ++i) {
      put(k, b, i, nullptr, c[i].localAxis.data(), c[i].globalAxis.data(), 0.f, c[i].weight);
    }
  }
  void setConstraints(size_t block, size_t b, const std::vector<NormalData>& c) {
    Block& k = expect(block, b, c.size(), MMX_JC_NORMAL, MMX_JC_NORMAL);
    for (size_t i = 0; i < c.size(); ++i) {
      put(k, b, i, c[i].localPoint.data(), c[i].localNormal.data(), c[i].globalPoint.data(), 0.f, c[i].weight);
    }
  }
  void setEnabledParameters(const ParameterSet& ps) {
    std::vector<uint8_t> e(character_.numParameters(), 0);
    for (size_t i = 0; i < e.size() && i < ps.size(); ++i) {
      e[i] = ps[i] ? 1 : 0;
    }
    check(mmx_problem_set_enabled(handle_.get(), e.data()));
  }
  // uploads the constraint payload if it changed
  void sync() {
    if (!dirty_) {
      return;
    }
    mmx_constraint_data d{};
    d.pos_offset = posOffset_.data();
    d.pos_target = posTarget_.data();
    d.pos_weight = posWeight_.data();
    d.ori_offset = oriOffset_.data();
    d.ori_target = oriTarget_.data();
    d.ori_weight = oriWeight_.data();
    d.pos_function_weight = wPos_;
    d.ori_function_weight = wOri_;
    d.memory = MMX_MEM_HOST;
    d.num_limits = int32_t(limits_.size());
    d.limits = limits_.empty() ? nullptr : limits_.data();
    d.limit_function_weight = wLimit_;
    d.model_target = mpTarget_.empty() ? nullptr : mpTarget_.data();
    d.model_weights = mpWeights_.empty() ? nullptr : mpWeights_.data();
    d.model_function_weight = wModel_;
    d.pos_loss_alpha = lossPos_[0], d.pos_loss_c = lossPos_[1];
    d.ori_loss_alpha = lossOri_[0], d.ori_loss_c = lossOri_[1];
    std::vector<mmx_joint_constraint_block> jb(blocks_.size());
    for (size_t i = 0; i < blocks_.size(); ++i) {
      const Block& k = blocks_[i];
      jb[i].type = k.type;
      jb[i].count = int32_t(k.parent.size());
      jb[i].parent = k.parent.data();
      jb[i].local_point = k.localPoint.data();
      jb[i].local_dir = k.localDir.data();
      jb[i].global = k.global.data();
      jb[i].plane_d = k.planeD.data();
      jb[i].weight = k.weight.data();
      jb[i].function_weight = k.fw;
      jb[i].loss_alpha = k.loss[0], jb[i].loss_c = k.loss[1];
    }
    d.num_joint_blocks = int32_t(jb.size());
    d.joint_blocks = jb.empty() ? nullptr : jb.data();
    d.function_weights = fnWeights_.empty() ? nullptr : fnWeights_.data();
    d.num_function_weights = int32_t(fnCols_);
    check(mmx_problem_set_constraints_sized(handle_.get(), &d, sizeof(d), nullptr));
    // per-element parent lists only when some element departs from the constructor's lists
    auto departs = [&](const std::vector<int32_t>& all, const std::vector<int32_t>& shared) {
      for (size_t b = 0; b < batch_; ++b) {
        if (!std::equal(shared.begin(), shared.end(), all.begin() + b * shared.size())) {
          return true;
        }
      }
      return false;
    };
    const bool ip = departs(posParent_, sharedPos_), io = departs(oriParent_, sharedOri_);
    if (ip || io || instanceParents_) {
      check(mmx_problem_set_instance_parents(handle_.get(), ip ? posParent_.data() : nullptr, io ? oriParent_.data() : nullptr, MMX_MEM_HOST, nullptr));
      instanceParents_ = ip || io;
    }
    dirty_ = false;
  }
  // SolverFunctionT::getJacobian for every element: column-major M x P per element, residual M
  void getJacobian(const std::vector<float>& parameters, std::vector<float>& jacobian, std::vector<float>& residual, std::vector<double>& error) {
    sync();
    const size_t M = size_t(mmx_problem_num_rows(handle_.get())), P = getNumParameters();
    if (parameters.size() != batch_ * P) {
      throw std::runtime_error("momentum_amd: parameters.size() != batch * numParameters"); // solver.cpp:77
    }
    jacobian.resize(batch_ * M * P);
    residual.resize(batch_ * M);
    error.resize(batch_);
    check(mmx_eval_jacobian_host(handle_.get(), parameters.data(), jacobian.data(), residual.data(), error.data(), MMX_LAYOUT_COL_MAJOR));
  }
  mmx_problem* handle() const {
    return handle_.get();
  }
  const DeviceCharacter& character() const {
    return character_;
  }

 private:
  struct Block {
    int32_t type = 0;
    std::vector<int32_t> parent;
    std::vector<float> localPoint, localDir, global, planeD, weight;
    float fw = 1.f;
    float loss[2] = {2.f, 1.f};
  };
  Block& blockAt(size_t block) {
    if (block >= blocks_.size()) {
      throw std::runtime_error("momentum_amd: joint error function index out of range");
    }
    return blocks_[block];
  }
  Block& expect(size_t block, size_t b, size_t count, int32_t typeLo, int32_t typeHi) {
    Block& k = blockAt(block);
    if (b >= batch_ || count != k.parent.size() || k.type < typeLo || k.type > typeHi) {
      throw std::runtime_error("momentum_amd: constraint data does not match the joint error function (type / count / batch index)");
    }
    dirty_ = true;
    return k;
  }
  void put(Block& k, size_t b, size_t i, const float* lp, const float* ld, const float* gl, float d, float w) {
    const size_t c = b * k.parent.size() + i;
    for (int q = 0; q < 3; ++q) {
      k.localPoint[3 * c + q] = lp != nullptr ? lp[q] : 0.f;
      k.localDir[3 * c + q] = ld != nullptr ? ld[q] : 0.f;
      k.global[3 * c + q] = gl[q];
    }
    k.planeD[c] = d;
    k.weight[c] = w;
  }
  std::vector<Block> blocks_;
  const DeviceCharacter& character_; // non-owning, like the reference (:87-88)
  size_t batch_, kp_, ko_;
  std::shared_ptr<mmx_problem> handle_;
  std::vector<float> posOffset_, posTarget_, posWeight_, oriOffset_, oriTarget_, oriWeight_;
  std::vector<int32_t> posParent_, oriParent_, sharedPos_, sharedOri_; // [batch * K] per element ; the constructor's lists
  bool instanceParents_ = false;
  std::vector<mmx_parameter_limit> limits_;
  std::vector<float> mpTarget_, mpWeights_;
  float wPos_ = 1.f, wOri_ = 1.f, wLimit_ = 1.f, wModel_ = 1.f;
  std::vector<float> fnWeights_; // [batch][fnCols_] per-element error-function weights, or empty
  size_t fnCols_ = 0;
  float lossPos_[2] = {2.f, 1.f}, lossOri_[2] = {2.f, 1.f};
  bool dirty_ = true;
};

// GaussNewtonSolverT<float> for every element of the batch at once.
// TransformT<float> (momentum/math/transform.h:36-42): p -> translation + rotation * (scale * p)
struct Transform {
  Vector3f translation{0.f, 0.f, 0.f};
  Quaternionf rotation{0.f, 0.f, 0.f, 1.f}; // (x, y, z, w)
  float scale = 1.f;
  Vector3f transformPoint(const Vector3f& p) const { // transform.h:193 (Eigen's quaternion * vector: v + w uv + qv x uv, uv = 2 qv x v)
    const float x = scale * p[0], y = scale * p[1], z = scale * p[2];
    const float qx = rotation[0], qy = rotation[1], qz = rotation[2], qw = rotation[3];
    const float ux = 2.f * (qy * z - qz * y), uy = 2.f * (qz * x - qx * z), uz = 2.f * (qx * y - qy * x);
    return Vector3f{
        translation[0] + x + qw * ux + (qy * uz - qz * uy),
        translation[1] + y + qw * uy + (qz * ux - qx * uz),
        translation[2] + z + qw * uz + (qx * uy - qy * ux)};
  }
};
// the world-space part of JointStateT<float> (momentum/character/joint_state.h:50-74)
struct JointState {
  Transform transform;
  const Vector3f& translation() const {
    return transform.translation;
  }
  const Quaternionf& rotation() const {
    return transform.rotation;
  }
  float scale() const {
    return transform.scale;
  }
};
// SkeletonStateT<float> (momentum/character/skeleton_state.h:45): jointState[j] of one character
struct SkeletonState {
  std::vector<JointState> jointState;
};
// SkeletonStateT<float>(parameterTransform.apply(modelParameters), skeleton) for every batch element
// (skeleton_state.cpp:22-28,87-121; the forward pass only, call stack SURVEY 3.3): element b is evaluated on ITS
// character when the function carries per-element characters.  modelParameters: [batch][numParameters].
class BatchedSkeletonState {
 public:
  BatchedSkeletonState() = default;
  BatchedSkeletonState(BatchedSkeletonSolverFunction& function, const std::vector<float>& modelParameters) {
    set(function, modelParameters);
  }
  void set(BatchedSkeletonSolverFunction& function, const std::vector<float>& modelParameters) {
    function.sync();
    const size_t B = function.batchSize(), P = function.getNumParameters();
    if (modelParameters.size() != B * P) {
      throw std::runtime_error("momentum_amd: parameters.size() != batch * numParameters"); // parameter_transform.cpp:112-121
    }
    const size_t J = size_t(mmx_rig_num_joints(function.character().handle()));
    std::vector<float> st(B * J * 8);
    check(mmx_eval_skeleton_state_host(function.handle(), modelParameters.data(), st.data()));
    states_.assign(B, SkeletonState{});
    for (size_t b = 0; b < B; ++b) {
      states_[b].jointState.resize(J);
      for (size_t j = 0; j < J; ++j) {
        const float* w = st.data() + (b * J + j) * 8;
        Transform& t = states_[b].jointState[j].transform;
        t.translation = Vector3f{w[0], w[1], w[2]};
        t.rotation = Quaternionf{w[3], w[4], w[5], w[6]};
        t.scale = w[7];
      }
    }
  }
  size_t batchSize() const {
    return states_.size();
  }
  const SkeletonState& operator[](size_t b) const {
    return states_.at(b);
  }

 private:
  std::vector<SkeletonState> states_;
};

class BatchedGaussNewtonSolver {
 public:
  BatchedGaussNewtonSolver(const SolverOptions& options, BatchedSkeletonSolverFunction* function) : fn_(function) {
    setOptions(options);
  }
  virtual ~BatchedGaussNewtonSolver() = default;
  virtual std::string getName() const {
    return "GaussNewton";
  }
  void setOptions(const SolverOptions& options) { // gauss_newton_solver.cpp:37-46
    mmx_gn_options_default(&opt_);
    opt_.min_iterations = int32_t(options.minIterations);
    opt_.max_iterations = int32_t(options.maxIterations);
    opt_.threshold = options.threshold;
    if (const auto* d = dynamic_cast<const GaussNewtonSolverBaseOptions*>(&options)) {
      opt_.regularization = d->regularization;
      opt_.do_line_search = d->doLineSearch ? lineSearchRule() : MMX_LINE_SEARCH_NONE;
    }
    applyDerivedOptions(options, opt_);
    opt_.precision = precision_;
    opt_.precision_bound = precisionBound_;
  }
  void setEnabledParameters(const ParameterSet& ps) {
    fn_->setEnabledParameters(ps);
  }
  // mmx_gn_options::precision (ABI 10) of solve(std::vector<float>&): MMX_PRECISION_F32 (default), MMX_PRECISION_F64 (every
  // element by the double instantiation, float parameters in and out), MMX_PRECISION_AUTO (the elements the single-precision
  // solve marks MMX_SOLVE_PRECISION_SUSPECT are solved again in double); bound <= 0 keeps the default (1e-5).  Kept across
  // setOptions().
  void setPrecision(int32_t precision, float bound = 0.f) {
    precision_ = precision;
    precisionBound_ = bound;
    opt_.precision = precision;
    opt_.precision_bound = bound;
  }
  // parameters [batch * P] in/out; returns the value SolverT::solve returns, per element
  std::vector<double> solve(std::vector<float>& parameters) {
    fn_->sync();
    const size_t B = fn_->batchSize(), P = fn_->getNumParameters();
    if (parameters.size() != B * P) {
      throw std::runtime_error("momentum_amd: parameters.size() != batch * numParameters"); // solver.cpp:77
    }
    std::vector<double> err(B);
    iterations_.assign(B, 0);
    status_.assign(B, 0);
    check(mmx_solve_host(fn_->handle(), &opt_, parameters.data(), err.data(), iterations_.data(), status_.data()));
    return err;
  }
  // SolverT<double>::solve with GaussNewtonSolverT<double> for every element (the reference instantiates its
  // solvers for float and double, gauss_newton_solver.cpp:315-316): parameters [batch * P] in double
  std::vector<double> solve(std::vector<double>& parameters) {
    fn_->sync();
    const size_t B = fn_->batchSize(), P = fn_->getNumParameters();
    if (parameters.size() != B * P) {
      throw std::runtime_error("momentum_amd: parameters.size() != batch * numParameters"); // solver.cpp:77
    }
    std::vector<double> err(B);
    iterations_.assign(B, 0);
    status_.assign(B, 0);
    check(mmx_solve_f64_host(fn_->handle(), &opt_, parameters.data(), err.data(), iterations_.data(), status_.data()));
    return err;
  }
  const std::vector<int32_t>& getIterations() const {
    return iterations_;
  }
  const std::vector<int32_t>& getStatus() const {
    return status_;
  }

 protected:
  virtual int32_t lineSearchRule() const { // gauss_newton_solver.cpp:283-313
    return MMX_LINE_SEARCH_GAUSS_NEWTON;
  }
  virtual void applyDerivedOptions(const SolverOptions&, mmx_gn_options&) const {}

 private:
  BatchedSkeletonSolverFunction* fn_; // raw pointer like SolverT::solverFunction_ (solver.h:106)
  mmx_gn_options opt_{};
  int32_t precision_ = MMX_PRECISION_F32;
  float precisionBound_ = 0.f;
  std::vector<int32_t> iterations_, status_;
};

// SubsetGaussNewtonSolverT<float> / GaussNewtonSolverQRT<float> for every element of the batch: the
// regularised normal equations of GaussNewtonSolverT (the QR solver factors [J; sqrt(lambda) I],
// gauss_newton_solver_qr.cpp:75-77, which is the same least-squares problem), with the directional
// line search both share (subset_gauss_newton_solver.cpp:117-142, gauss_newton_solver_qr.cpp:126-149).
class BatchedSubsetGaussNewtonSolver : public BatchedGaussNewtonSolver {
 public:
  BatchedSubsetGaussNewtonSolver(const SolverOptions& options, BatchedSkeletonSolverFunction* function)
      : BatchedGaussNewtonSolver(SolverOptions(options), function) {
    setOptions(options); // the base constructor ran with the base class's rule
  }
  std::string getName() const override {
    return "SubsetGaussNewton";
  }

 protected:
  int32_t lineSearchRule() const override {
    return MMX_LINE_SEARCH_DIRECTIONAL;
  }
};
class BatchedGaussNewtonSolverQR : public BatchedSubsetGaussNewtonSolver {
 public:
  using BatchedSubsetGaussNewtonSolver::BatchedSubsetGaussNewtonSolver;
  std::string getName() const override {
    return "GaussNewtonQR";
  }
};

// TrustRegionQRT<float> (and, through solve(std::vector<double>&), <double>) for every element of the batch
// (trust_region_qr.cpp:52-270): the step rule
// MMX_STEP_TRUST_REGION -- radius, up to ten trust steps per iteration, Newton updates of the damping, gain-ratio
// radius update -- inside the one-launch solve where the problem fits it, on the wide route otherwise
// (larger systems, further joint error functions, ellipsoid limits).
class BatchedTrustRegionQR : public BatchedGaussNewtonSolver {
 public:
  BatchedTrustRegionQR(const SolverOptions& options, BatchedSkeletonSolverFunction* function)
      : BatchedGaussNewtonSolver(SolverOptions(options), function) {
    setOptions(options); // the base constructor ran without this class's hook
  }
  std::string getName() const override {
    return "TrustRegionQR";
  }

 protected:
  void applyDerivedOptions(const SolverOptions& options, mmx_gn_options& o) const override {
    o.step_rule = MMX_STEP_TRUST_REGION;
    o.do_line_search = MMX_LINE_SEARCH_NONE;
    if (const auto* d = dynamic_cast<const TrustRegionQROptions*>(&options)) {
      o.trust_region_radius = d->trustRegionRadius_;
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// Single-instance forms under momentum's own names and signatures -- what a per-frame caller holds
// (momentum/marker_tracking/marker_tracker.cpp:905-913: one SkeletonSolverFunction, one solver, solve(parameters) per
// frame): a batch of one element behind momentum/character_solver/skeleton_solver_function.h:21-95 and
// momentum/solver/solver.h:41-106.  (A GPU launch per frame is latency, not throughput: 0.5 ms per solve whatever the
// batch up to ~768 elements; callers with many frames at hand should batch them.)
// ---------------------------------------------------------------------------------------------------------------------
class SkeletonSolverFunction : public BatchedSkeletonSolverFunction {
 public:
  SkeletonSolverFunction(const DeviceCharacter& character, const std::vector<size_t>& positionParents, const std::vector<size_t>& orientationParents)
      : BatchedSkeletonSolverFunction(character, 1, positionParents, orientationParents) {}
  using BatchedSkeletonSolverFunction::setConstraints;
  void setPositionConstraints(const std::vector<PositionData>& c) {
    BatchedSkeletonSolverFunction::setPositionConstraints(0, c);
  }
  void setOrientationConstraints(const std::vector<OrientationData>& c) {
    BatchedSkeletonSolverFunction::setOrientationConstraints(0, c);
  }
  void setTargetParameters(const std::vector<float>& params, const std::vector<float>& weights, float weight = 1.f) {
    BatchedSkeletonSolverFunction::setTargetParameters(0, params, weights, weight);
  }
  template <class Data>
  void setConstraints(size_t block, const std::vector<Data>& c) {
    BatchedSkeletonSolverFunction::setConstraints(block, 0, c);
  }
  // SolverFunctionT::getJacobian (momentum/solver/solver_function.h): Jacobian [rows x P] column-major, residual, returns the error
  double getJacobian(const std::vector<float>& parameters, std::vector<float>& jacobian, std::vector<float>& residual) {
    std::vector<double> e;
    BatchedSkeletonSolverFunction::getJacobian(parameters, jacobian, residual, e);
    return e.at(0);
  }
  // SolverFunctionT::getError: the same evaluation without keeping the Jacobian
  double getError(const std::vector<float>& parameters) {
    std::vector<float> j, r;
    return getJacobian(parameters, j, r);
  }
};

// SolverT::solve(Eigen::VectorX<T>& params) -> double (momentum/solver/solver.h:62) over the batched solvers: one element
template <class BatchedSolver>
class SingleInstanceSolverT : public BatchedSolver {
 public:
  SingleInstanceSolverT(const SolverOptions& options, SkeletonSolverFunction* function) : BatchedSolver(options, function) {}
  double solve(std::vector<float>& parameters) {
    return BatchedSolver::solve(parameters).at(0);
  }
  double solve(std::vector<double>& parameters) {
    return BatchedSolver::solve(parameters).at(0);
  }
  int32_t iterations() const {
    return this->getIterations().at(0);
  }
  int32_t status() const {
    return this->getStatus().at(0);
  }
};
using GaussNewtonSolver = SingleInstanceSolverT<BatchedGaussNewtonSolver>;
using SubsetGaussNewtonSolver = SingleInstanceSolverT<BatchedSubsetGaussNewtonSolver>;
using GaussNewtonSolverQR = SingleInstanceSolverT<BatchedGaussNewtonSolverQR>;
using TrustRegionQR = SingleInstanceSolverT<BatchedTrustRegionQR>;

} // namespace momentum_amd
