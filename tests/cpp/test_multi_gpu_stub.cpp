// BatchedMultiGpuSolverT (include/momentum_amd/multi_gpu.hpp) with stubbed device classes and an in-process exchange:
// the host-side logic of the N > 1 path -- shard routing of per-element calls, ragged and empty shards, every rank
// joining the collective also after a failure, the failure being rethrown after the join, the reduced norms being the
// batch totals on every rank -- on a host without several GPUs (RCCL cannot take one device twice, so the 1-GPU box can
// only ever run it with one rank; tests/cpp/test_multi_gpu.cpp is the same class with the real policies).
// The batched driver this mirrors: pymomentum/tensor_ik/tensor_ik.cpp:127-186 (one task per element, exceptions
// collected and rethrown after the parallel_for).
#include <condition_variable>
#include <cstdio>
#include <mutex>

#include "momentum_amd/multi_gpu.hpp"

using namespace momentum_amd;

namespace {

struct FakeCharacter {
  FakeCharacter(const Character& c, int device) : numParams(c.parameterTransform.numAllModelParameters()), device(device) {}
  size_t numParams;
  int device;
};

struct FakeFunction {
  FakeFunction(const FakeCharacter& c, size_t batch, const std::vector<size_t>& pp, const std::vector<size_t>&) : character(c), batch(batch), kp(pp.size()), target(batch, 0.f) {}
  void setPositionConstraints(size_t b, const std::vector<PositionData>& c) {
    if (b >= batch || c.size() != kp) {
      throw std::runtime_error("stub: bad element index");
    }
    target[b] = c[0].target[0];
  }
  void setOrientationConstraints(size_t, const std::vector<OrientationData>&) {}
  const FakeCharacter& character;
  size_t batch, kp;
  std::vector<float> target; // per local element: what the "solve" writes into every parameter
};

struct FakeOptions {
  int failOnDevice = -1; // the shard on this device throws in solve()
};

struct FakeSolver {
  FakeSolver(const FakeOptions& o, FakeFunction* f) : options(o), fn(f) {}
  std::vector<double> solve(std::vector<float>& parameters) {
    if (fn->character.device == options.failOnDevice) {
      throw std::runtime_error("stub: device lost");
    }
    const size_t P = fn->character.numParams;
    if (parameters.size() != fn->batch * P) {
      throw std::runtime_error("stub: slice size");
    }
    std::vector<double> err(fn->batch);
    iterations.assign(fn->batch, 0);
    status.assign(fn->batch, 0);
    for (size_t b = 0; b < fn->batch; ++b) {
      for (size_t p = 0; p < P; ++p) {
        parameters[b * P + p] += fn->target[b]; // element-local: a wrong routing shows in the result
      }
      err[b] = double(fn->target[b]);
      iterations[b] = 3;
      status[b] = fn->target[b] < 0.f ? 2 : 0;
    }
    return err;
  }
  const std::vector<int32_t>& getIterations() const {
    return iterations;
  }
  const std::vector<int32_t>& getStatus() const {
    return status;
  }
  FakeOptions options;
  FakeFunction* fn;
  std::vector<int32_t> iterations, status;
};

// an all-reduce among the threads of one process: every rank adds its three doubles, the last one to arrive releases all
struct ThreadComm {
  struct Shared {
    std::mutex m;
    std::condition_variable cv;
    size_t world = 0, arrived = 0, generation = 0;
    double sum[3] = {0, 0, 0}, result[3] = {0, 0, 0};
  };
  using Handle = std::shared_ptr<Shared>;
  static std::vector<Handle> createAll(const std::vector<int>& devices) {
    auto sh = std::make_shared<Shared>();
    sh->world = devices.size();
    return std::vector<Handle>(devices.size(), sh);
  }
  static size_t worldSize(const Handle& h) {
    return h->world;
  }
  static bool allReduce(const Handle& h, double v[3], std::string&) {
    std::unique_lock<std::mutex> lk(h->m);
    const size_t gen = h->generation;
    for (int i = 0; i < 3; ++i) {
      h->sum[i] += v[i];
    }
    if (++h->arrived == h->world) {
      for (int i = 0; i < 3; ++i) {
        h->result[i] = h->sum[i];
        h->sum[i] = 0;
      }
      h->arrived = 0;
      ++h->generation;
      h->cv.notify_all();
    } else {
      h->cv.wait(lk, [&] { return h->generation != gen; });
    }
    for (int i = 0; i < 3; ++i) {
      v[i] = h->result[i];
    }
    return true;
  }
};

using StubSolver = BatchedMultiGpuSolverT<FakeSolver, FakeFunction, FakeCharacter, ThreadComm>;

Character tinyCharacter(size_t P) {
  Character c;
  Joint j;
  j.name = "root";
  j.parent = kInvalidIndex;
  c.skeleton.joints.push_back(j);
  for (size_t p = 0; p < P; ++p) {
    c.parameterTransform.name.push_back("p" + std::to_string(p));
  }
  c.parameterTransform.setFromTriplets(1, {});
  return c;
}

int fail(const char* what) {
  std::printf("FAIL: %s\n", what);
  return 1;
}

} // namespace

int main() {
  const size_t P = 5;
  const Character character = tinyCharacter(P);
  // (world, batch): even, ragged, fewer elements than ranks (empty shards), one rank
  const size_t cases[][2] = {{8, 64}, {8, 61}, {4, 10}, {8, 3}, {1, 7}, {2, 2}};
  for (const auto& cs : cases) {
    const size_t world = cs[0], B = cs[1];
    std::vector<int> devices(world);
    for (size_t i = 0; i < world; ++i) {
      devices[i] = int(i);
    }
    StubSolver solver(character, devices, B, {0}, {}, FakeOptions{});
    if (solver.numShards() != world || solver.commWorldSize() != world) {
      return fail("one shard and one rank per device");
    }
    // shards tile the batch, contiguous, in order
    size_t next = 0;
    for (size_t i = 0; i < world; ++i) {
      const auto r = solver.shardBounds(i);
      if (r.first != next || r.second < r.first || r != shardRange(B, i, world)) {
        return fail("shard bounds");
      }
      next = r.second;
    }
    if (next != B) {
      return fail("shards do not cover the batch");
    }
    std::vector<float> theta(B * P);
    double expectErr = 0.0, expectFailed = 0.0;
    for (size_t b = 0; b < B; ++b) {
      PositionData pc;
      pc.parent = 0;
      pc.target = {float(b) - (b % 7 == 3 ? 100.f : 0.f), 0.f, 0.f}; // a few negative ones: status 2
      solver.setPositionConstraints(b, {pc});
      for (size_t p = 0; p < P; ++p) {
        theta[b * P + p] = 1000.f * float(p);
      }
      expectErr += double(pc.target[0]);
      expectFailed += pc.target[0] < 0.f ? 1.0 : 0.0;
    }
    const std::vector<double> err = solver.solve(theta);
    for (size_t b = 0; b < B; ++b) {
      const float t = float(b) - (b % 7 == 3 ? 100.f : 0.f);
      if (err[b] != double(t)) {
        return fail("per-element return value landed on the wrong element");
      }
      for (size_t p = 0; p < P; ++p) {
        if (theta[b * P + p] != 1000.f * float(p) + t) {
          return fail("parameters of an element were solved by the wrong shard slot");
        }
      }
    }
    const auto& n = solver.norms();
    if (n[0] != expectErr || n[1] != 3.0 * double(B) || n[2] != expectFailed) {
      return fail("reduced norms are not the batch totals");
    }
    if (B == 3 && world == 8) { // out-of-range element
      try {
        solver.setPositionConstraints(B, {PositionData{}});
        return fail("batch index out of range accepted");
      } catch (const std::runtime_error&) {
      }
    }
  }
  // a failing rank: the others finish, every rank joins the collective (no hang), the failure is rethrown after the join
  {
    std::vector<int> devices = {0, 1, 2, 3};
    FakeOptions o;
    o.failOnDevice = 2;
    StubSolver solver(character, devices, 12, {0}, {}, o);
    for (size_t b = 0; b < 12; ++b) {
      PositionData pc;
      pc.parent = 0;
      pc.target = {1.f, 0.f, 0.f};
      solver.setPositionConstraints(b, {pc});
    }
    std::vector<float> theta(12 * P, 0.f);
    bool thrown = false;
    try {
      solver.solve(theta);
    } catch (const std::runtime_error& e) {
      thrown = std::string(e.what()).find("device lost") != std::string::npos;
    }
    if (!thrown) {
      return fail("a shard's exception was not rethrown after the join");
    }
    for (size_t b = 0; b < 12; ++b) { // the healthy shards' elements were solved, the failing shard's untouched
      const bool failing = b >= 6 && b < 9;
      if (theta[b * P] != (failing ? 0.f : 1.f)) {
        return fail("healthy shards must complete when one fails");
      }
    }
    // wrong parameter vector size: solver.cpp:77
    std::vector<float> wrong(5);
    try {
      solver.solve(wrong);
      return fail("size mismatch accepted");
    } catch (const std::runtime_error&) {
    }
  }
  std::printf("OK\n");
  return 0;
}
