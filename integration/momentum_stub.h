// momentum_stub.h -- the few reference types integration/tensor_ik_mmx_adapter.cpp touches, restated at the size the
// adapter needs them, so that the adapter is COMPILED against include/mmx.h in this repository's tests (the reference's
// own headers need Eigen / fmt / gsl, none of which exist here).  A momentum maintainer builds the adapter with
// -DMMX_ADAPTER_WITH_MOMENTUM against the real headers instead; member names and signatures below are the reference's:
//   momentum::JointT / Skeleton          momentum/character/joint.h:18-36, skeleton.h:22-25, types.h:182 (kInvalidIndex)
//   momentum::ParameterTransform         momentum/character/parameter_transform.h:62-95 (transform: Eigen::SparseMatrix<float, RowMajor>)
//   momentum::Character                  momentum/character/character.h:32-125
//   momentum::ParameterSet               momentum/character/types.h (std::bitset<kMaxModelParams>), math/types.h:426-429
//   pymomentum::SolverOptions            pymomentum/tensor_ik/solver_options.h:28-37
//   MT_THROW_IF                          momentum/common/exception.h:60-66
#pragma once

#include <bitset>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace Eigen { // the two accessors the adapter uses of each Eigen type
struct Vector3f {
  float v[3] = {0.f, 0.f, 0.f};
  const float* data() const { return v; }
};
struct Vector4f {
  float v[4] = {0.f, 0.f, 0.f, 1.f};
  const float* data() const { return v; }
};
struct Quaternionf {
  Vector4f c; // (x, y, z, w): Eigen's storage order
  const Vector4f& coeffs() const { return c; }
};
struct VectorXf {
  std::vector<float> v;
  const float* data() const { return v.data(); }
  long size() const { return long(v.size()); }
};
struct SparseRowMatrixf { // Eigen::SparseMatrix<float, Eigen::RowMajor>, compressed
  std::vector<int> outer{0}, inner;
  std::vector<float> values;
  const int* outerIndexPtr() const { return outer.data(); }
  const int* innerIndexPtr() const { return inner.data(); }
  const float* valuePtr() const { return values.data(); }
  long rows() const { return long(outer.size()) - 1; }
  long nonZeros() const { return long(inner.size()); }
};
} // namespace Eigen

namespace momentum {
inline constexpr size_t kInvalidIndex = std::numeric_limits<size_t>::max();
inline constexpr size_t kParametersPerJoint = 7;
inline constexpr size_t kMaxModelParams = 2048;
using ParameterSet = std::bitset<kMaxModelParams>;
struct Joint {
  std::string name;
  size_t parent = kInvalidIndex;
  Eigen::Quaternionf preRotation;
  Eigen::Vector3f translationOffset;
};
struct Skeleton {
  std::vector<Joint> joints;
};
struct ParameterTransform {
  std::vector<std::string> name;
  Eigen::SparseRowMatrixf transform;
  Eigen::VectorXf offsets;
  size_t numAllModelParameters() const { return name.size(); }
};
struct Character {
  Skeleton skeleton;
  ParameterTransform parameterTransform;
};
} // namespace momentum

namespace pymomentum {
enum class LinearSolverType { Cholesky, QR, TrustRegionQR };
struct SolverOptions {
  LinearSolverType linearSolverType = LinearSolverType::QR;
  float levmar_lambda = 0.01f;
  size_t minIter = 4;
  size_t maxIter = 50;
  float threshold = 10.0f;
  bool lineSearch = true;
};
} // namespace pymomentum

#define MT_THROW_IF(cond, msg) \
  do {                         \
    if (cond) {                \
      throw std::runtime_error(std::string(msg)); \
    }                          \
  } while (0)
