// adapter_check.cpp -- exercises integration/tensor_ik_mmx_adapter.cpp: builds momentum's three-joint test character
// (momentum/test/character/character_helpers.cpp:38-149 shape) in the (stub) reference types, converts it, has the library
// validate the descriptor on the host (mmx_host_tables: no GPU needed), and -- when a device is present -- solves a batch
// through solveBatch and checks that the constraint points reach their targets.
#include <cmath>
#include <cstdio>

#include "tensor_ik_mmx_adapter.h"

static momentum::Character testCharacter() {
  momentum::Character c;
  const char* names[3] = {"root", "joint1", "joint2"};
  for (int j = 0; j < 3; ++j) {
    momentum::Joint jt;
    jt.name = names[j];
    jt.parent = j == 0 ? momentum::kInvalidIndex : size_t(j - 1);
    jt.translationOffset.v[1] = j == 0 ? 0.f : 1.f;
    c.skeleton.joints.push_back(jt);
  }
  auto& pt = c.parameterTransform;
  pt.name = {"root_tx", "root_ty", "root_tz", "root_rx", "root_ry", "root_rz", "scale_global", "joint1_rx", "shared_rz", "joint2_rx"};
  // rows 7 j + (tx ty tz rx ry rz sc); CSR by rows
  const int rowOf[12] = {0, 1, 2, 3, 4, 5, 6, 7 + 3, 7 + 5, 14 + 3, 14 + 5, -1};
  const int colOf[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 8, -1};
  const float valOf[12] = {1, 1, 1, 1, 1, 1, 1, 1, 0.5f, 1, 0.5f, 0};
  pt.transform.outer.assign(1, 0);
  for (int r = 0; r < 21; ++r) {
    for (int k = 0; k < 11; ++k) {
      if (rowOf[k] == r) {
        pt.transform.inner.push_back(colOf[k]);
        pt.transform.values.push_back(valOf[k]);
      }
    }
    pt.transform.outer.push_back(int(pt.transform.inner.size()));
  }
  pt.offsets.v.assign(21, 0.f);
  return c;
}

int main() {
  const momentum::Character c = testCharacter();
  const mmx_adapter::RigArrays a = mmx_adapter::makeRigDesc(c);
  int32_t level[3], tin[3], tout[3], enabledList[10], numEnabled = 0;
  uint8_t active[21];
  if (mmx_host_tables(&a.desc, nullptr, level, tin, tout, active, enabledList, &numEnabled) != MMX_OK) {
    std::printf("FAIL: mmx_host_tables rejected the converted character: %s\n", mmx_last_error());
    return 1;
  }
  if (numEnabled != 10 || level[2] != 2 || !active[7 + 3] || !active[14 + 5] || active[7 + 0]) {
    std::printf("FAIL: host tables of the converted character\n");
    return 1;
  }
  if (mmx_device_count() <= 0) {
    std::printf("descriptor accepted; no device: solveBatch compiled and linked, not run\nOK\n");
    return 0;
  }
  mmx_rig* rig = mmx_adapter::makeRig(c, 0);
  const int B = 4;
  const int32_t parents[1] = {2};
  std::vector<float> off(B * 3, 0.f), tgt(B * 3), w(B, 1.f), theta(B * 10, 0.f);
  for (int b = 0; b < B; ++b) {
    off[3 * b + 1] = 1.f; // UnitY on joint 2, as in momentum/test/character_solver/inverse_kinematics_test.cpp:60-99
    tgt[3 * b] = 0.3f * float(b), tgt[3 * b + 1] = 2.5f, tgt[3 * b + 2] = -0.2f * float(b);
  }
  mmx_adapter::BatchTensors t;
  t.nBatch = B, t.numPositions = 1, t.positionParents = parents;
  t.positionOffsets = off.data(), t.positionTargets = tgt.data(), t.positionWeights = w.data();
  std::vector<char> floored(B, 0);
  t.dampingFloored = reinterpret_cast<bool*>(floored.data());
  pymomentum::SolverOptions o;
  o.linearSolverType = pymomentum::LinearSolverType::QR, o.levmar_lambda = 0.01f, o.minIter = 4, o.maxIter = 50, o.threshold = 10.f, o.lineSearch = true;
  momentum::ParameterSet all;
  all.set();
  std::vector<int32_t> status(B, 0);
  t.status = status.data();
  const mmx_adapter::SolveReport rep = mmx_adapter::solveBatch(rig, all, t, o, theta.data()); // solveTensorIKProblem<float>: MMX_PRECISION_AUTO
  // whatever the policy decided per element (single precision where its own estimate holds 1e-5, the mixed-precision
  // instantiation or the double kernel elsewhere), nothing may have failed and the counts must add up
  if (rep.failed != 0 || rep.mixed + rep.escalatedF64 > B) {
    std::printf("FAIL: AUTO report: suspect %lld mixed %lld f64 %lld failed %lld\n", (long long)rep.precisionSuspect, (long long)rep.mixed, (long long)rep.escalatedF64, (long long)rep.failed);
    return 1;
  }
  // solveTensorIKProblem<double> through the same adapter: the double instantiation on double parameters; the float answer follows it
  std::vector<double> thetaD(B * 10, 0.0);
  const mmx_adapter::SolveReport repD = mmx_adapter::solveBatch(rig, all, t, o, thetaD.data());
  if (repD.failed != 0 || repD.mixed != 0) {
    std::printf("FAIL: double report\n");
    return 1;
  }
  for (int i = 0; i < B * 10; ++i) {
    if (!(std::fabs(double(theta[i]) - thetaD[i]) <= 1e-5 * (1.0 + std::fabs(thetaD[i])))) {
      std::printf("FAIL: parameter %d: float/AUTO %.9g, double %.9g\n", i, double(theta[i]), thetaD[i]);
      return 1;
    }
  }
  std::printf("AUTO: %lld of %d elements by the mixed-precision instantiation, %lld by the double kernel; within 1e-5 of solveBatch<double>\n", (long long)rep.mixed, B, (long long)rep.escalatedF64);
  // the solved pose puts joint 2's UnitY point on its target: check through the forward pass of the library
  mmx_problem* pb = nullptr;
  if (mmx_problem_create(rig, B, 0, nullptr, 0, nullptr, &pb) != MMX_OK) {
    std::printf("FAIL: %s\n", mmx_last_error());
    return 1;
  }
  std::vector<float> st(B * 3 * 8);
  if (mmx_eval_skeleton_state_host(pb, theta.data(), st.data()) != MMX_OK) {
    std::printf("FAIL: %s\n", mmx_last_error());
    return 1;
  }
  for (int b = 0; b < B; ++b) {
    const float* s = st.data() + (b * 3 + 2) * 8;
    const float qx = s[3], qy = s[4], qz = s[5], qw = s[6], sc = s[7];
    // q * (0, sc, 0)
    const float vx = 0.f, vy = sc, vz = 0.f;
    const float ux = 2.f * (qy * vz - qz * vy), uy = 2.f * (qz * vx - qx * vz), uz = 2.f * (qx * vy - qy * vx);
    const float px = s[0] + vx + qw * ux + (qy * uz - qz * uy), py = s[1] + vy + qw * uy + (qz * ux - qx * uz), pz = s[2] + vz + qw * uz + (qx * uy - qy * ux);
    const float d = std::sqrt((px - tgt[3 * b]) * (px - tgt[3 * b]) + (py - tgt[3 * b + 1]) * (py - tgt[3 * b + 1]) + (pz - tgt[3 * b + 2]) * (pz - tgt[3 * b + 2]));
    if (!(d <= 1e-3f)) {
      std::printf("FAIL: element %d ends %.3e from its target\n", b, d);
      return 1;
    }
  }
  mmx_problem_destroy(pb);
  mmx_rig_destroy(rig);
  std::printf("solveBatch: four elements on their targets\nOK\n");
  return 0;
}
