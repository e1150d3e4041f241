// tensor_ik_mmx_adapter.h -- see tensor_ik_mmx_adapter.cpp
#pragma once

#ifdef MMX_ADAPTER_WITH_MOMENTUM
#include <momentum/character/character.h>
#include <pymomentum/tensor_ik/solver_options.h>
#else
#include "momentum_stub.h"
#endif

#include <mmx.h>

#include <cstdint>
#include <vector>

namespace mmx_adapter {

struct RigArrays {
  std::vector<int32_t> parent;
  std::vector<float> pre, off;
  mmx_rig_desc desc; // points into the vectors above and into the character's parameter transform
};
RigArrays makeRigDesc(const momentum::Character& c);
mmx_rig* makeRig(const momentum::Character& c, int device);

// the tensors solveTensorIKProblem receives, as the host pointers of their data (float32 / int32, contiguous)
struct BatchTensors {
  int64_t nBatch = 0;
  int32_t numPositions = 0, numOrientations = 0;
  const int32_t* positionParents = nullptr; // [numPositions]
  const int32_t* orientationParents = nullptr; // [numOrientations]
  const float *positionOffsets = nullptr, *positionTargets = nullptr, *positionWeights = nullptr; // [B][K][3], [B][K][3], [B][K]
  const float *orientationOffsets = nullptr, *orientationTargets = nullptr, *orientationWeights = nullptr; // [B][K][4] (x,y,z,w), .., [B][K]
  const float* errorFunctionWeights = nullptr; // [B][numWeightColumns] or null
  int numWeightColumns = 0;
  int weightsMap[2] = {0, 1}; // column of the position / orientation error function, < 0: switched off
  const float* perElementTranslationOffsets = nullptr; // [B][J][3] or null: characters[iBatch] of one topology
  const float* perElementPreRotations = nullptr; // [B][J][4] or null
  const int32_t* perElementPositionParents = nullptr; // [B][numPositions] or null
  const int32_t* perElementOrientationParents = nullptr; // [B][numOrientations] or null
  bool* dampingFloored = nullptr; // [B] out, or null
  int32_t* status = nullptr; // [B] out, or null: the elements' MMX_SOLVE_* bit sets
};
// what a batch's precision policy did (ABI 11): how many elements the single-precision solve marked, how many the
// mixed-precision instantiation and the double kernel solved again, how many failed (MMX_SOLVE_FAILED: non-finite -> reverted)
struct SolveReport {
  int64_t precisionSuspect = 0, mixed = 0, escalatedF64 = 0, failed = 0, dampingFloored = 0;
};
// solveTensorIKProblem<float>: float parameters in and out.  precision: MMX_PRECISION_AUTO by default -- single precision where its
// own estimate holds north_star's 1e-5 against the double instantiation, the mixed-precision instantiation (double theta / forward
// kinematics / residuals / g around the single-precision factor) on the elements it marks, the double kernel on what that cannot
// converge; MMX_PRECISION_F32 is BASELINE's metric, MMX_PRECISION_MIXED / F64 follow GaussNewtonSolverT<double> everywhere.
SolveReport solveBatch(mmx_rig* rig, const momentum::ParameterSet& activeParams, const BatchTensors& t, const pymomentum::SolverOptions& options, float* modelParameters,
                       int32_t precision = MMX_PRECISION_AUTO);
// solveTensorIKProblem<double> (tensor_ik.cpp is templated on T; gauss_newton_solver.cpp:315-316 instantiates both): double
// parameters in and out through mmx_solve_f64_host, every element by the double instantiation
SolveReport solveBatch(mmx_rig* rig, const momentum::ParameterSet& activeParams, const BatchTensors& t, const pymomentum::SolverOptions& options, double* modelParameters);

} // namespace mmx_adapter
