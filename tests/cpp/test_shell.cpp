// GPU smoke test of the C++ shell (include/momentum_amd/momentum_amd.hpp): builds momentum's test
// fixture createTestCharacter(24) (momentum/test/character/character_helpers.cpp:38-55,106-149),
// solves a batch of 3-position-constraint IK problems (BASELINE configs[0] shape) on the GPU and
// checks that the error collapses and that bad input throws std::runtime_error like MT_CHECK.
#include <cmath>
#include <cstdio>
#include <random>

#include "momentum_amd/momentum_amd.hpp"

using namespace momentum_amd;

static Character createTestCharacter(size_t n) {
  Character c;
  Joint j;
  j.name = "root";
  c.skeleton.joints.push_back(j);
  for (size_t i = 1; i < n; ++i) {
    j.name = "joint" + std::to_string(i);
    j.parent = i - 1;
    j.translationOffset = {0.f, 1.f, 0.f};
    c.skeleton.joints.push_back(j);
  }
  auto& pt = c.parameterTransform;
  pt.name = {"root_tx", "root_ty", "root_tz", "root_rx", "root_ry", "root_rz", "scale_global", "joint1_rx", "shared_rz"};
  const int rxStart = int(pt.name.size());
  for (size_t i = 2; i < n; ++i) {
    pt.name.push_back("joint" + std::to_string(i) + "_rx");
  }
  std::vector<ParameterTransform::Triplet> t;
  for (int d = 0; d < 7; ++d) {
    t.push_back({d, d, 1.f});
  }
  t.push_back({1 * 7 + 3, 7, 1.f});
  t.push_back({1 * 7 + 5, 8, 0.5f});
  t.push_back({2 * 7 + 5, 8, 0.5f});
  for (size_t i = 2; i < n; ++i) {
    t.push_back({int(i * 7 + 3), rxStart + int(i) - 2, 1.f});
  }
  pt.setFromTriplets(n, t);
  return c;
}

int main() {
  const size_t n = 24, B = 8;
  const Character character = createTestCharacter(n);
  if (character.parameterTransform.numAllModelParameters() != 31) {
    std::printf("FAIL: expected 31 parameters\n");
    return 1;
  }
  DeviceCharacter dev(character, 0);
  BatchedSkeletonSolverFunction fn(dev, B, {23, 12, 5}, {});
  std::mt19937 rng(12345);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (size_t b = 0; b < B; ++b) {
    std::vector<PositionData> cons(3);
    const size_t parents[3] = {23, 12, 5};
    for (int i = 0; i < 3; ++i) {
      cons[i].parent = parents[i];
      cons[i].offset = {0.f, 0.f, 0.f};
      cons[i].target = {0.5f * U(rng), float(parents[i]) * 0.9f + 0.3f * U(rng), 0.5f * U(rng)}; // near the rest pose
      cons[i].weight = 1.f;
    }
    fn.setPositionConstraints(b, cons);
  }
  GaussNewtonSolverOptions opt;
  opt.minIterations = 10;
  opt.maxIterations = 10;
  opt.regularization = 0.05f;
  BatchedGaussNewtonSolver solver(opt, &fn);
  const size_t P = fn.getNumParameters();
  std::vector<float> theta(B * P, 0.f);
  std::vector<float> jac, res;
  std::vector<double> e0;
  fn.getJacobian(theta, jac, res, e0);
  const std::vector<double> e = solver.solve(theta);
  int bad = 0;
  for (size_t b = 0; b < B; ++b) {
    std::printf("instance %zu: error %.6g -> %.3g, iterations %d, status %d\n", b, e0[b], e[b], solver.getIterations()[b], solver.getStatus()[b]);
    if (!(e[b] < 1e-3 * e0[b]) || solver.getIterations()[b] != 10 || (solver.getStatus()[b] & MMX_SOLVE_ERROR_MASK) != 0) {
      ++bad;
    }
  }
  // size mismatch must throw like MT_CHECK(params.size() == numParameters_) (solver.cpp:77)
  bool threw = false;
  try {
    std::vector<float> wrong(P);
    solver.solve(wrong);
  } catch (const std::runtime_error&) {
    threw = true;
  }
  if (!threw) {
    ++bad;
  }
  // parameter limits + model-parameter prior (LimitErrorFunction / ModelParametersErrorFunction):
  // a MinMax limit on parameter 3 must pull the solution into (or close to) its range
  {
    fn.setLimits({ParameterLimit::minMax(3, -0.01f, 0.01f, 100.f), ParameterLimit::linear(4, 5, 1.f, 0.f)}, 1.f);
    for (size_t b = 0; b < B; ++b) {
      fn.setTargetParameters(b, std::vector<float>(P, 0.f), std::vector<float>(P, 0.1f), 1.f);
    }
    std::vector<float> th2(B * P, 0.f);
    std::vector<double> e1;
    fn.getJacobian(th2, jac, res, e1);
    if (jac.size() != B * (9 + 2 + P) * P) {
      ++bad;
    }
    solver.solve(th2);
    for (size_t b = 0; b < B; ++b) {
      std::printf("instance %zu with limits: theta[3] %.4f -> %.4f, theta[4]-theta[5] %.4f\n", b, theta[b * P + 3], th2[b * P + 3], th2[b * P + 4] - th2[b * P + 5]);
      if (std::fabs(th2[b * P + 3]) > std::fabs(theta[b * P + 3]) + 1e-6f || std::fabs(th2[b * P + 3]) > 0.05f) {
        ++bad;
      }
    }
  }
  // a further joint error function: half-plane "floor" constraints y >= 2 on joints 3 and 10 of the
  // chain (PlaneErrorFunction(above = true)) + a fixed-axis constraint; the solve must lift both joints
  // to the plane and the error of the block must vanish
  {
    BatchedSkeletonSolverFunction fn2(dev, B, {23}, {});
    const size_t floor = fn2.addJointErrorFunction(JointErrorFunctionType::HalfPlane, {3, 10});
    const size_t axis = fn2.addJointErrorFunction(JointErrorFunctionType::FixedAxisDiff, {12});
    fn2.setWeight(axis, 0.5f);
    for (size_t b = 0; b < B; ++b) {
      std::vector<PositionData> pc(1);
      pc[0].parent = 23;
      pc[0].target = {0.f, 25.f, 0.f};
      pc[0].weight = 1e-3f;
      fn2.setPositionConstraints(b, pc);
      std::vector<PlaneData> fl(2);
      fl[0].parent = 3, fl[1].parent = 10;
      fl[0].normal = fl[1].normal = {0.f, 2.f, 0.f}; // normalised on ingest
      fl[0].d = 5.f, fl[1].d = 12.f;
      fn2.setConstraints(floor, b, fl);
      std::vector<FixedAxisData> fa(1);
      fa[0].parent = 12;
      fa[0].localAxis = {0.f, 1.f, 0.f};
      fa[0].globalAxis = {0.f, 1.f, 0.f};
      fn2.setConstraints(axis, b, fa);
    }
    BatchedGaussNewtonSolver solver2(opt, &fn2);
    std::vector<float> th3(B * P, 0.f);
    std::vector<double> e2;
    fn2.getJacobian(th3, jac, res, e2);
    if (jac.size() != B * (3 + 2 + 3) * P) {
      ++bad;
    }
    const std::vector<double> e3 = solver2.solve(th3);
    std::printf("floor constraints: error %.4g -> %.4g\n", e2[0], e3[0]);
    if (!(e2[0] > 1.0) || !(e3[0] < 1e-2 * e2[0])) {
      ++bad;
    }
    bool threw2 = false;
    try {
      fn2.setConstraints(floor, 0, std::vector<AimData>(2)); // wrong data type for the block
    } catch (const std::runtime_error&) {
      threw2 = true;
    }
    if (!threw2) {
      ++bad;
    }
  }
  // the two solvers the batched driver builds (tensor_ik.cpp:142-158), with their line search
  {
    SubsetGaussNewtonSolverOptions so(opt);
    so.doLineSearch = true;
    BatchedSubsetGaussNewtonSolver subset(so, &fn);
    GaussNewtonSolverQROptions qo(opt);
    qo.doLineSearch = true;
    BatchedGaussNewtonSolverQR qr(qo, &fn);
    std::vector<float> ta(B * P, 0.f), tb(B * P, 0.f);
    const std::vector<double> ea = subset.solve(ta), eb = qr.solve(tb);
    std::printf("%s / %s with line search: error %.3g / %.3g\n", subset.getName().c_str(), qr.getName().c_str(), ea[0], eb[0]);
    if (subset.getName() != "SubsetGaussNewton" || qr.getName() != "GaussNewtonQR" || ta != tb || !(ea[0] < 1e-3 * e0[0])) {
      ++bad;
    }
  }
  // the third one: TrustRegionQR (same function: position constraints + limits + prior).  What the step rule computes
  // is pinned against the oracle in tests/test_gpu_trust_region.py; here: the class selects it (a result that is not
  // Gauss-Newton's, every element far down from the start) and trustRegionRadius_ reaches the kernel (three iterations
  // of radius 0.05 from the rest pose cannot get as far as three of radius 1)
  {
    GaussNewtonSolverOptions go(opt);
    go.minIterations = go.maxIterations = 20;
    BatchedGaussNewtonSolver gn(go, &fn);
    TrustRegionQROptions to(go);
    to.trustRegionRadius_ = 1.0f;
    BatchedTrustRegionQR tr(to, &fn);
    std::vector<float> ta(B * P, 0.f), tb(B * P, 0.f);
    std::vector<double> es;
    fn.getJacobian(ta, jac, res, es);
    const std::vector<double> eg = gn.solve(ta), et = tr.solve(tb);
    std::printf("%s: error %.4g -> %.3g (GaussNewton %.3g)\n", tr.getName().c_str(), es[0], et[0], eg[0]);
    bool differs = false;
    for (size_t b = 0; b < B; ++b) {
      if (!(et[b] < 0.1 * es[b]) || (tr.getStatus()[b] & MMX_SOLVE_ERROR_MASK) != 0) { // (MMX_SOLVE_DAMPING_FLOORED is set: the rule starts from 1e-10)
        std::printf("  instance %zu: %.4g vs %.4g (start %.4g), status %d\n", b, et[b], eg[b], es[b], tr.getStatus()[b]);
        ++bad;
      }
      differs = differs || et[b] != eg[b];
    }
    TrustRegionQROptions small(to), wide(to);
    small.minIterations = small.maxIterations = wide.minIterations = wide.maxIterations = 3;
    small.trustRegionRadius_ = 0.05f;
    std::vector<float> tc(B * P, 0.f), td(B * P, 0.f);
    tr.setOptions(small);
    const std::vector<double> e_small = tr.solve(tc);
    tr.setOptions(wide);
    const std::vector<double> e_wide = tr.solve(td);
    std::printf("  three iterations: radius 0.05 %.4g, radius 1 %.4g\n", e_small[0], e_wide[0]);
    if (!differs || !(e_small[0] > e_wide[0]) || tr.getName() != "TrustRegionQR") {
      ++bad;
    }
    // TrustRegionQRT<double>: the same solver object on double parameters (mmx_solve_f64 runs the rule in double).  The
    // fixture is under-determined, so the two precisions part ways at some trial decision: the double run has to be as
    // good a fit as the float one, not the same one (parity of the double rule with the oracle: tests/test_gpu_f64.py)
    tr.setOptions(to);
    std::vector<double> tdbl(B * P, 0.0);
    const std::vector<double> etd = tr.solve(tdbl);
    std::printf("  double instantiation: error %.6g (float %.6g)\n", etd[0], et[0]);
    for (size_t b = 0; b < B; ++b) {
      if (!(etd[b] < 0.1 * es[b]) || (tr.getStatus()[b] & MMX_SOLVE_ERROR_MASK) != 0 || !(etd[b] <= 1.5 * et[b] + 1e-3)) {
        std::printf("  instance %zu: double %.6g vs float %.6g, status %d\n", b, etd[b], et[b], tr.getStatus()[b]);
        ++bad;
      }
    }
  }
  // the single-instance forms under momentum's own names (solver.h:41-106, skeleton_solver_function.h:21-95): a batch of one
  // gives the batched solve's element 0 bit for bit; MMX_PRECISION_F64 / AUTO through the shell's setPrecision
  {
    BatchedSkeletonSolverFunction fnB(dev, B, {23, 12, 5}, {}); // (a fresh function: `fn` carries limits and a prior by now)
    {
      std::mt19937 rngB(12345);
      std::uniform_real_distribution<float> UB(-1.f, 1.f);
      for (size_t b = 0; b < B; ++b) {
        std::vector<PositionData> cons(3);
        const size_t parents[3] = {23, 12, 5};
        for (int i = 0; i < 3; ++i) {
          cons[i].parent = parents[i];
          cons[i].offset = {0.f, 0.f, 0.f};
          cons[i].target = {0.5f * UB(rngB), float(parents[i]) * 0.9f + 0.3f * UB(rngB), 0.5f * UB(rngB)};
          cons[i].weight = 1.f;
        }
        fnB.setPositionConstraints(b, cons);
      }
    }
    BatchedGaussNewtonSolver solverB(opt, &fnB);
    std::vector<float> thB(B * P, 0.f);
    const std::vector<double> eB = solverB.solve(thB);
    SkeletonSolverFunction one(dev, {23, 12, 5}, {});
    {
      std::mt19937 rng1(12345); // element 0's constraints again
      std::uniform_real_distribution<float> U1(-1.f, 1.f);
      std::vector<PositionData> cons(3);
      const size_t parents[3] = {23, 12, 5};
      for (int i = 0; i < 3; ++i) {
        cons[i].parent = parents[i];
        cons[i].offset = {0.f, 0.f, 0.f};
        cons[i].target = {0.5f * U1(rng1), float(parents[i]) * 0.9f + 0.3f * U1(rng1), 0.5f * U1(rng1)};
        cons[i].weight = 1.f;
      }
      one.setPositionConstraints(cons);
    }
    GaussNewtonSolver single(opt, &one);
    std::vector<float> th1(P, 0.f);
    const double e1 = one.getError(th1);
    const double eS = single.solve(th1);
    bool same = eS == eB[0] && single.iterations() == 10 && !MMX_SOLVE_FAILED(single.status());
    for (size_t i = 0; i < P; ++i) {
      same = same && th1[i] == thB[i];
    }
    std::printf("single instance: error %.6g -> %.3g (batched element 0: %.3g), identical %d\n", e1, eS, eB[0], int(same));
    if (!same || !(e1 > 0.0)) {
      ++bad;
    }
    single.setPrecision(MMX_PRECISION_F64);
    std::vector<float> thD(P, 0.f);
    const double eD = single.solve(thD);
    single.setPrecision(MMX_PRECISION_MIXED); // ABI 11: double theta / FK / residuals / g around the single-precision factor
    std::vector<float> thM(P, 0.f);
    const double eM = single.solve(thM);
    bool okp = (single.status() & MMX_SOLVE_MIXED) != 0 && (single.status() & MMX_SOLVE_ESCALATED_F64) == 0;
    single.setPrecision(MMX_PRECISION_AUTO, 1e-30f); // a bound nothing passes: the element is solved again -- by the mixed instantiation
    std::vector<float> thA(P, 0.f);
    const double eA = single.solve(thA);
    okp = okp && (single.status() & MMX_SOLVE_MIXED) != 0 && eA == eM;
    for (size_t i = 0; i < P; ++i) {
      okp = okp && thA[i] == thM[i] && std::fabs(thD[i] - th1[i]) <= 1e-2f * (1.f + std::fabs(th1[i]));
      okp = okp && std::fabs(thM[i] - thD[i]) <= 1e-5f * (1.f + std::fabs(thD[i])); // the mixed run follows the double one
    }
    std::printf("precision: double on float parameters %.3g, mixed %.3g, AUTO (second pass: mixed) %.3g, identical %d\n", eD, eM, eA, int(okp));
    if (!okp) {
      ++bad;
    }
  }
  std::printf(bad == 0 ? "OK\n" : "FAIL\n");
  return bad == 0 ? 0 : 1;
}
