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
};
void solveBatch(mmx_rig* rig, const momentum::ParameterSet& activeParams, const BatchTensors& t, const pymomentum::SolverOptions& options, float* modelParameters);

} // namespace mmx_adapter
