// Parity of the C++ shell (include/momentum_amd/momentum_amd.hpp) against committed golden numbers: the
// inputs of tests/golden/cfg2_humanoid72.npz (BASELINE configs[1]: 72-joint humanoid, position +
// orientation constraints on 16 landmark joints, 10 Gauss-Newton iterations, lambda = 0.05) go through
// Character -> DeviceCharacter -> BatchedSkeletonSolverFunction -> BatchedGaussNewtonSolver, and the pose
// parameters must agree with the CPU oracle's double-precision solve stored in the fixture to 1e-5
// relative (north_star), the iteration counts exactly, the returned errors to 1e-4.  Then the same numbers
// once more through the per-element API: every element on its own Character object and its own parent
// list (all equal to the shared ones here, so the answers must be bit-identical to the shared solve), and
// elements with scaled bones must differ.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "momentum_amd/momentum_amd.hpp"

#include "golden_cfg2.inc"

using namespace momentum_amd;

static Character goldenCharacter(float boneScale) {
  Character c;
  for (int j = 0; j < kJ; ++j) {
    Joint jt;
    jt.name = "j" + std::to_string(j);
    jt.parent = kParent[j] < 0 ? kInvalidIndex : size_t(kParent[j]);
    jt.preRotation = {kPreRotation[4 * j], kPreRotation[4 * j + 1], kPreRotation[4 * j + 2], kPreRotation[4 * j + 3]};
    jt.translationOffset = {boneScale * kTranslationOffset[3 * j], boneScale * kTranslationOffset[3 * j + 1], boneScale * kTranslationOffset[3 * j + 2]};
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

static void fill(BatchedSkeletonSolverFunction& fn) {
  for (int b = 0; b < kB; ++b) {
    std::vector<PositionData> pc(kKp);
    for (int i = 0; i < kKp; ++i) {
      const int e = b * kKp + i;
      pc[i].parent = size_t(k_pos_parent[i]);
      pc[i].offset = {k_pos_offset[3 * e], k_pos_offset[3 * e + 1], k_pos_offset[3 * e + 2]};
      pc[i].target = {k_pos_target[3 * e], k_pos_target[3 * e + 1], k_pos_target[3 * e + 2]};
      pc[i].weight = k_pos_weight[e];
    }
    fn.setPositionConstraints(size_t(b), pc);
    std::vector<OrientationData> oc(kKo);
    for (int i = 0; i < kKo; ++i) {
      const int e = b * kKo + i;
      oc[i].parent = size_t(k_ori_parent[i]);
      oc[i].offset = {k_ori_offset[4 * e], k_ori_offset[4 * e + 1], k_ori_offset[4 * e + 2], k_ori_offset[4 * e + 3]};
      oc[i].target = {k_ori_target[4 * e], k_ori_target[4 * e + 1], k_ori_target[4 * e + 2], k_ori_target[4 * e + 3]};
      oc[i].weight = k_ori_weight[e];
    }
    fn.setOrientationConstraints(size_t(b), oc);
  }
}

static double relativeDifference(const float* a, const double* ref, int n) {
  double num = 0.0, den = 0.0;
  for (int i = 0; i < n; ++i) {
    num += (double(a[i]) - ref[i]) * (double(a[i]) - ref[i]);
    den += ref[i] * ref[i];
  }
  return std::sqrt(num / den);
}

int main() {
  const Character character = goldenCharacter(1.f);
  DeviceCharacter dev(character, 0);
  std::vector<size_t> pp(k_pos_parent, k_pos_parent + kKp), op(k_ori_parent, k_ori_parent + kKo);
  GaussNewtonSolverOptions options;
  options.minIterations = size_t(kIterations);
  options.maxIterations = size_t(kIterations);
  options.threshold = 1.f;
  options.regularization = 0.05f;
  options.doLineSearch = false;

  // ---- shared character, shared parent lists
  BatchedSkeletonSolverFunction fn(dev, size_t(kB), pp, op);
  fill(fn);
  BatchedGaussNewtonSolver solver(options, &fn);
  std::vector<float> theta(k_theta0, k_theta0 + kB * kP);
  const std::vector<double> err = solver.solve(theta);
  double worst = 0.0;
  for (int b = 0; b < kB; ++b) {
    const double rel = relativeDifference(theta.data() + b * kP, k_theta_final + b * kP, kP);
    worst = rel > worst ? rel : worst;
    if (!(rel <= 1e-5)) {
      std::printf("FAIL: element %d pose parameters differ from the golden solve by %.3e (> 1e-5)\n", b, rel);
      return 1;
    }
    if (solver.getIterations()[size_t(b)] != k_iterations[b] || solver.getStatus()[size_t(b)] != 0) {
      std::printf("FAIL: element %d iterations %d (golden %d), status %d\n", b, solver.getIterations()[size_t(b)], k_iterations[b], solver.getStatus()[size_t(b)]);
      return 1;
    }
    if (!(std::fabs(err[size_t(b)] - k_final_error[b]) <= 1e-4 * std::fmax(1e-3, std::fabs(k_final_error[b])))) {
      std::printf("FAIL: element %d returned error %.9g, golden %.9g\n", b, err[size_t(b)], k_final_error[b]);
      return 1;
    }
  }
  std::printf("shared character: max relative pose-parameter difference vs golden %.3e\n", worst);

  // ---- the double instantiation through the same classes: 1e-10 against the golden double solve
  {
    std::vector<double> thd(size_t(kB) * kP);
    for (size_t i = 0; i < thd.size(); ++i) {
      thd[i] = double(k_theta0[i]);
    }
    solver.solve(thd);
    for (int b = 0; b < kB; ++b) {
      double num = 0.0, den = 0.0;
      for (int p = 0; p < kP; ++p) {
        const double d = thd[size_t(b) * kP + p] - k_theta_final[b * kP + p];
        num += d * d, den += k_theta_final[b * kP + p] * k_theta_final[b * kP + p];
      }
      if (!(std::sqrt(num / den) <= 1e-10)) {
        std::printf("FAIL: double solve, element %d differs from the golden double solve by %.3e (> 1e-10)\n", b, std::sqrt(num / den));
        return 1;
      }
    }
    std::printf("double instantiation: within 1e-10 of the golden double solve\n");
  }

  // ---- the driver's solver classes on the same problem: SubsetGaussNewton / GaussNewtonQR with the line
  // search on must still land on the golden pose (the full step passes the Armijo test on this fixture)
  {
    SubsetGaussNewtonSolverOptions so;
    so.minIterations = so.maxIterations = size_t(kIterations);
    so.threshold = 1.f;
    so.regularization = 0.05f;
    so.doLineSearch = true;
    BatchedGaussNewtonSolverQR qr(so, &fn);
    std::vector<float> th2(k_theta0, k_theta0 + kB * kP);
    qr.solve(th2);
    for (int b = 0; b < kB; ++b) {
      const double rel = relativeDifference(th2.data() + b * kP, k_theta_final + b * kP, kP);
      if (!(rel <= 1e-5)) {
        std::printf("FAIL: %s element %d differs by %.3e\n", qr.getName().c_str(), b, rel);
        return 1;
      }
    }
  }

  // ---- one Character object and one parent list per element (all equal): bit-identical answers
  std::vector<Character> own(size_t(kB), character);
  std::vector<const Character*> ptrs;
  for (const Character& c : own) {
    ptrs.push_back(&c);
  }
  BatchedSkeletonSolverFunction fn2(dev, size_t(kB), std::vector<size_t>(size_t(kKp), 0), std::vector<size_t>(size_t(kKo), 0)); // default lists differ: every element carries its own
  fill(fn2);
  fn2.setCharacters(ptrs);
  BatchedGaussNewtonSolver solver2(options, &fn2);
  std::vector<float> theta2(k_theta0, k_theta0 + kB * kP);
  solver2.solve(theta2);
  if (std::memcmp(theta.data(), theta2.data(), theta.size() * sizeof(float)) != 0) {
    std::printf("FAIL: per-element characters / parents (all equal) do not reproduce the shared solve bit for bit\n");
    return 1;
  }
  // ---- a subject with 10 %% longer bones in element 1: that element moves, the others stay bit-identical
  own[1] = goldenCharacter(1.1f);
  fn2.setCharacters(ptrs);
  std::vector<float> theta3(k_theta0, k_theta0 + kB * kP);
  solver2.solve(theta3);
  for (int b = 0; b < kB; ++b) {
    const bool same = std::memcmp(theta.data() + b * kP, theta3.data() + b * kP, kP * sizeof(float)) == 0;
    if (same != (b != 1)) {
      std::printf("FAIL: element %d %s after element 1 got its own bone lengths\n", b, same ? "did not change" : "changed");
      return 1;
    }
  }
  // a character of another topology is refused like an MT_CHECK
  bool threw = false;
  try {
    Character other = character;
    other.skeleton.joints[5].parent = 0;
    std::vector<const Character*> bad(ptrs);
    bad[2] = &other;
    fn2.setCharacters(bad);
  } catch (const std::runtime_error&) {
    threw = true;
  }
  if (!threw) {
    std::printf("FAIL: a character of another topology was accepted\n");
    return 1;
  }
  std::printf("OK\n");
  return 0;
}
