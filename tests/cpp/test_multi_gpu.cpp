// BatchedMultiGpuSolver (include/momentum_amd/multi_gpu.hpp) on every visible GPU: the golden fixture's
// elements (tests/golden/cfg2_humanoid72.npz), replicated to 3 * devices + 1 elements so that the shards are
// ragged, one host thread and one RCCL rank per device.  Every element must land on its golden pose (1e-5),
// and the all-reduced residual norms must be the batch totals on every rank.
#include <cmath>
#include <cstdio>

#include "momentum_amd/multi_gpu.hpp"

#include "golden_cfg2.inc"

using namespace momentum_amd;

static Character goldenCharacter() {
  Character c;
  for (int j = 0; j < kJ; ++j) {
    Joint jt;
    jt.name = "j" + std::to_string(j);
    jt.parent = kParent[j] < 0 ? kInvalidIndex : size_t(kParent[j]);
    jt.preRotation = {kPreRotation[4 * j], kPreRotation[4 * j + 1], kPreRotation[4 * j + 2], kPreRotation[4 * j + 3]};
    jt.translationOffset = {kTranslationOffset[3 * j], kTranslationOffset[3 * j + 1], kTranslationOffset[3 * j + 2]};
    c.skeleton.joints.push_back(jt);
  }
  for (int p = 0; p < kP; ++p) {
    c.parameterTransform.name.push_back("p" + std::to_string(p));
  }
  std::vector<ParameterTransform::Triplet> t;
  for (int k = 0; k < kNnz; ++k) {
    t.push_back({kPtRow[k], kPtCol[k], kPtValue[k]});
  }
  c.parameterTransform.setFromTriplets(size_t(kJ), t);
  return c;
}

int main() {
  // shard arithmetic (ragged and empty shards)
  if (shardRange(10, 3, 4) != std::pair<size_t, size_t>(9, 10) || shardRange(2, 3, 4) != std::pair<size_t, size_t>(2, 2) ||
      shardRange(7, 0, 1) != std::pair<size_t, size_t>(0, 7)) {
    std::printf("FAIL: shardRange\n");
    return 1;
  }
  const int ndev = mmx_device_count();
  if (ndev <= 0) {
    std::printf("FAIL: no device\n");
    return 1;
  }
  std::vector<int> devices;
  for (int d = 0; d < ndev; ++d) {
    devices.push_back(d);
  }
  const size_t B = 3 * size_t(ndev) + 1;
  const Character character = goldenCharacter();
  std::vector<size_t> pp(k_pos_parent, k_pos_parent + kKp), op(k_ori_parent, k_ori_parent + kKo);
  GaussNewtonSolverOptions options;
  options.minIterations = options.maxIterations = size_t(kIterations);
  options.threshold = 1.f;
  options.regularization = 0.05f;
  BatchedMultiGpuSolver solver(character, devices, B, pp, op, options);
  std::vector<float> theta(B * kP);
  for (size_t b = 0; b < B; ++b) {
    const int g = int(b % size_t(kB));
    std::vector<PositionData> pc(kKp);
    for (int i = 0; i < kKp; ++i) {
      const int e = g * kKp + i;
      pc[i].parent = size_t(k_pos_parent[i]);
      pc[i].offset = {k_pos_offset[3 * e], k_pos_offset[3 * e + 1], k_pos_offset[3 * e + 2]};
      pc[i].target = {k_pos_target[3 * e], k_pos_target[3 * e + 1], k_pos_target[3 * e + 2]};
      pc[i].weight = k_pos_weight[e];
    }
    solver.setPositionConstraints(b, pc);
    std::vector<OrientationData> oc(kKo);
    for (int i = 0; i < kKo; ++i) {
      const int e = g * kKo + i;
      oc[i].parent = size_t(k_ori_parent[i]);
      oc[i].offset = {k_ori_offset[4 * e], k_ori_offset[4 * e + 1], k_ori_offset[4 * e + 2], k_ori_offset[4 * e + 3]};
      oc[i].target = {k_ori_target[4 * e], k_ori_target[4 * e + 1], k_ori_target[4 * e + 2], k_ori_target[4 * e + 3]};
      oc[i].weight = k_ori_weight[e];
    }
    solver.setOrientationConstraints(b, oc);
    for (int p = 0; p < kP; ++p) {
      theta[b * kP + p] = k_theta0[g * kP + p];
    }
  }
  const std::vector<double> err = solver.solve(theta);
  double sumErr = 0.0;
  for (size_t b = 0; b < B; ++b) {
    const int g = int(b % size_t(kB));
    double num = 0.0, den = 0.0;
    for (int p = 0; p < kP; ++p) {
      const double d = double(theta[b * kP + p]) - k_theta_final[g * kP + p];
      num += d * d;
      den += k_theta_final[g * kP + p] * k_theta_final[g * kP + p];
    }
    if (!(std::sqrt(num / den) <= 1e-5)) {
      std::printf("FAIL: element %zu differs from the golden pose by %.3e\n", b, std::sqrt(num / den));
      return 1;
    }
    sumErr += err[b];
  }
  const auto& n = solver.norms();
  if (std::fabs(n[0] - sumErr) > 1e-9 * std::fmax(1.0, std::fabs(sumErr)) || n[1] != double(B) * kIterations || n[2] != 0.0) {
    std::printf("FAIL: reduced norms (%.9g, %g, %g), expected (%.9g, %g, 0)\n", n[0], n[1], n[2], sumErr, double(B) * kIterations);
    return 1;
  }
  if (solver.commWorldSize() != solver.numShards()) {
    std::printf("FAIL: RCCL counts %zu ranks for %zu shards\n", solver.commWorldSize(), solver.numShards());
    return 1;
  }
  // the same batch through the driver's default solver type (GaussNewtonSolverQR: directional line search) on every device
  {
    GaussNewtonSolverQROptions qo;
    qo.minIterations = qo.maxIterations = size_t(kIterations);
    qo.threshold = 1.f;
    qo.regularization = 0.05f;
    qo.doLineSearch = true;
    BatchedMultiGpuSolverT<BatchedGaussNewtonSolverQR> qr(character, devices, B, pp, op, qo);
    std::vector<float> th2(B * kP, 0.f);
    for (size_t b = 0; b < B; ++b) {
      const int g = int(b % size_t(kB));
      std::vector<PositionData> pc(kKp);
      for (int i = 0; i < kKp; ++i) {
        const int e = g * kKp + i;
        pc[i].parent = size_t(k_pos_parent[i]);
        pc[i].offset = {k_pos_offset[3 * e], k_pos_offset[3 * e + 1], k_pos_offset[3 * e + 2]};
        pc[i].target = {k_pos_target[3 * e], k_pos_target[3 * e + 1], k_pos_target[3 * e + 2]};
        pc[i].weight = k_pos_weight[e];
      }
      qr.setPositionConstraints(b, pc);
      std::vector<OrientationData> oc(kKo);
      for (int i = 0; i < kKo; ++i) {
        const int e = g * kKo + i;
        oc[i].parent = size_t(k_ori_parent[i]);
        oc[i].offset = {k_ori_offset[4 * e], k_ori_offset[4 * e + 1], k_ori_offset[4 * e + 2], k_ori_offset[4 * e + 3]};
        oc[i].target = {k_ori_target[4 * e], k_ori_target[4 * e + 1], k_ori_target[4 * e + 2], k_ori_target[4 * e + 3]};
        oc[i].weight = k_ori_weight[e];
      }
      qr.setOrientationConstraints(b, oc);
      for (int p = 0; p < kP; ++p) {
        th2[b * kP + p] = k_theta0[g * kP + p];
      }
    }
    const std::vector<double> e2 = qr.solve(th2);
    for (size_t b = 0; b < B; ++b) { // a line search only ever lowers the error of the plain step's run
      if (!(e2[b] <= err[b] * 1.001 + 1e-6) || !std::isfinite(double(th2[b * kP]))) {
        std::printf("FAIL: line-search solver on %zu shards, element %zu: error %.6g against %.6g\n", qr.numShards(), b, e2[b], err[b]);
        return 1;
      }
    }
  }
  std::printf("%d device(s), %zu elements in %zu shards, RCCL ranks: %zu\nOK\n", ndev, B, solver.numShards(), solver.commWorldSize());
  return 0;
}
