// multi_gpu.hpp -- the batched solver over several GPUs of one node, in C++ over the C ABI (include/mmx.h).
//
// The reference solves a batch as one independent task per element (pymomentum/tensor_ik/tensor_ik.cpp:
// 127-177).  Here the batch is cut into contiguous shards, one per device; each device gets its own
// DeviceCharacter / BatchedSkeletonSolverFunction / solver (momentum_amd.hpp) and its own host thread, and
// NO data crosses devices during a solve.  The one exchange is the per-batch residual norms -- (sum of final
// errors, sum of iterations, number of failed elements) -- all-reduced once per solve by RCCL over xGMI
// (mmx_comm_*), so that every shard's owner sees the batch totals.
#pragma once

#include <array>
#include <exception>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "momentum_amd.hpp"

namespace momentum_amd {

// contiguous shard [begin, end) of rank `rank` out of `world`: ceil(total / world) elements per rank, the
// last ranks may be short or empty
inline std::pair<size_t, size_t> shardRange(size_t total, size_t rank, size_t world) {
  const size_t per = (total + world - 1) / world;
  const size_t begin = std::min(rank * per, total);
  return {begin, std::min(begin + per, total)};
}

// The exchange of the multi-GPU solver: one communicator per shard, created together, that sums three doubles over
// the shards.  The default is RCCL through the C ABI (mmx_comm_*); the policy exists so that the shard routing, the
// ragged / empty shards and the failure protocol of BatchedMultiGpuSolverT can be exercised on a host without several
// GPUs (tests/cpp/test_multi_gpu_stub.cpp: RCCL cannot take one device twice).
struct RcclNormsComm {
  using Handle = std::shared_ptr<mmx_comm>;
  static std::vector<Handle> createAll(const std::vector<int>& devices) {
    std::vector<int32_t> devs(devices.begin(), devices.end());
    std::vector<mmx_comm*> comms(devices.size(), nullptr);
    std::vector<Handle> out;
    out.reserve(comms.size()); // before any communicator exists: a failing allocation then has nothing to leak
    check(mmx_comm_create_all(int32_t(devs.size()), devs.data(), comms.data()));
    size_t wrapped = 0;
    try {
      for (; wrapped < comms.size(); ++wrapped) { // (shared_ptr's constructor runs the deleter itself when it throws)
        out.emplace_back(comms[wrapped], [](mmx_comm* p) { mmx_comm_destroy(p); });
      }
    } catch (...) {
      for (size_t i = wrapped + 1; i < comms.size(); ++i) { // the ones no handle owns yet
        mmx_comm_destroy(comms[i]);
      }
      throw;
    }
    return out;
  }
  static size_t worldSize(const Handle& h) {
    return size_t(mmx_comm_world_size(h.get()));
  }
  // in-place sum over the shards; false + message on failure
  static bool allReduce(const Handle& h, double v[3], std::string& error) {
    if (mmx_comm_all_reduce_norms_host(h.get(), v) != MMX_OK) {
      error = mmx_last_error();
      return false;
    }
    return true;
  }
};

// SolverT: BatchedGaussNewtonSolver or one of its subclasses (the line-search rule / step rule differs) -- anything with
// SolverT(options, FunctionT*), solve(std::vector<float>&) -> std::vector<double>, getIterations(), getStatus().
template <class SolverT = BatchedGaussNewtonSolver, class FunctionT = BatchedSkeletonSolverFunction, class CharacterT = DeviceCharacter, class CommT = RcclNormsComm>
class BatchedMultiGpuSolverT {
 public:
  // One shard per entry of `devices` (HIP device indices).
  template <class OptionsT>
  BatchedMultiGpuSolverT(
      const Character& character,
      const std::vector<int>& devices,
      size_t batch,
      const std::vector<size_t>& positionParents,
      const std::vector<size_t>& orientationParents,
      const OptionsT& options)
      : batch_(batch), numParams_(character.parameterTransform.numAllModelParameters()) {
    if (devices.empty()) {
      throw std::runtime_error("momentum_amd: no device given");
    }
    for (size_t i = 0; i < devices.size(); ++i) {
      const auto range = shardRange(batch, i, devices.size());
      Shard sh;
      sh.begin = range.first, sh.end = range.second, sh.device = devices[i];
      sh.character = std::make_unique<CharacterT>(character, devices[i]);
      if (sh.end > sh.begin) {
        sh.function = std::make_unique<FunctionT>(*sh.character, sh.end - sh.begin, positionParents, orientationParents);
        sh.solver = std::make_unique<SolverT>(options, sh.function.get());
      }
      shards_.push_back(std::move(sh));
    }
    std::vector<typename CommT::Handle> comms = CommT::createAll(devices);
    if (comms.size() != shards_.size()) {
      throw std::runtime_error("momentum_amd: one communicator per shard expected");
    }
    for (size_t i = 0; i < comms.size(); ++i) {
      shards_[i].comm = comms[i];
    }
  }

  size_t numShards() const {
    return shards_.size();
  }
  size_t batchSize() const {
    return batch_;
  }
  // the shard that owns batch element b, and b's index inside it
  std::pair<size_t, size_t> locate(size_t b) const {
    for (size_t i = 0; i < shards_.size(); ++i) {
      if (b >= shards_[i].begin && b < shards_[i].end) {
        return {i, b - shards_[i].begin};
      }
    }
    throw std::runtime_error("momentum_amd: batch index out of range");
  }
  FunctionT& function(size_t shard) {
    return *shards_.at(shard).function;
  }
  SolverT& solver(size_t shard) {
    return *shards_.at(shard).solver;
  }
  std::pair<size_t, size_t> shardBounds(size_t shard) const {
    return {shards_.at(shard).begin, shards_.at(shard).end};
  }
  // ranks the exchange sees (RCCL: ncclCommCount): must equal numShards()
  size_t commWorldSize() const {
    return CommT::worldSize(shards_.front().comm);
  }
  void setPositionConstraints(size_t b, const std::vector<PositionData>& c) {
    const auto at = locate(b);
    shards_[at.first].function->setPositionConstraints(at.second, c);
  }
  void setOrientationConstraints(size_t b, const std::vector<OrientationData>& c) {
    const auto at = locate(b);
    shards_[at.first].function->setOrientationConstraints(at.second, c);
  }

  // parameters [batch * P] in/out.  Returns SolverT::solve's value per element.  Afterwards norms() holds the
  // batch totals (identical on every shard: they come out of the all-reduce).
  std::vector<double> solve(std::vector<float>& parameters) {
    if (parameters.size() != batch_ * numParams_) {
      throw std::runtime_error("momentum_amd: parameters.size() != batch * numParameters"); // solver.cpp:77
    }
    std::vector<double> errors(batch_, 0.0);
    std::vector<std::exception_ptr> failures(shards_.size());
    std::vector<std::array<double, 3>> reduced(shards_.size());
    std::vector<std::thread> threads;
    for (size_t i = 0; i < shards_.size(); ++i) {
      threads.emplace_back([&, i]() {
        Shard& sh = shards_[i];
        std::array<double, 3> local{0.0, 0.0, 0.0};
        try {
          if (sh.end > sh.begin) {
            std::vector<float> slice(parameters.begin() + sh.begin * numParams_, parameters.begin() + sh.end * numParams_);
            const std::vector<double> err = sh.solver->solve(slice);
            std::copy(slice.begin(), slice.end(), parameters.begin() + sh.begin * numParams_);
            std::copy(err.begin(), err.end(), errors.begin() + sh.begin);
            for (size_t k = 0; k < err.size(); ++k) {
              local[0] += err[k];
              local[1] += double(sh.solver->getIterations()[k]);
              local[2] += (sh.solver->getStatus()[k] & MMX_SOLVE_ERROR_MASK) != 0 ? 1.0 : 0.0;
            }
          }
        } catch (...) {
          failures[i] = std::current_exception(); // rethrown after the join, like the reference (tensor_ik.cpp:179-186)
        }
        // every rank joins the collective, also after a failure: a missing rank would hang the others
        std::string commError;
        if (!CommT::allReduce(sh.comm, local.data(), commError) && !failures[i]) {
          failures[i] = std::make_exception_ptr(std::runtime_error("momentum_amd: " + commError));
        }
        reduced[i] = local;
      });
    }
    for (std::thread& t : threads) {
      t.join();
    }
    for (const std::exception_ptr& f : failures) {
      if (f) {
        std::rethrow_exception(f);
      }
    }
    norms_ = reduced[0];
    for (const auto& r : reduced) {
      if (r != norms_) {
        throw std::runtime_error("momentum_amd: ranks disagree on the reduced residual norms");
      }
    }
    return errors;
  }
  // (sum of the returned errors, sum of iterations, number of failed elements) over the whole batch
  const std::array<double, 3>& norms() const {
    return norms_;
  }

 private:
  struct Shard {
    size_t begin = 0, end = 0;
    int device = 0;
    std::unique_ptr<CharacterT> character;
    std::unique_ptr<FunctionT> function;
    std::unique_ptr<SolverT> solver;
    typename CommT::Handle comm;
  };
  size_t batch_, numParams_;
  std::vector<Shard> shards_;
  std::array<double, 3> norms_{0.0, 0.0, 0.0};
};

using BatchedMultiGpuSolver = BatchedMultiGpuSolverT<>; // Gauss-Newton, fixed lambda, no line search: the BASELINE metric

} // namespace momentum_amd
