// mmx_capi.hip -- implementation of the C ABI declared in include/mmx.h.
// Host-side glue only: handle ownership, device buffers, table upload, kernel sequencing.
// No compute happens on the host and there is no CPU fallback: without a HIP device every compute
// entry point returns MMX_ERR_NO_DEVICE / MMX_ERR_DEVICE.
#include "../../include/mmx.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "mmx_host_tables.hpp"
#include "mmx_kernels.hpp"

namespace {

thread_local std::string g_lastError;

int32_t fail(int32_t code, const std::string& msg) {
  g_lastError = msg;
  return code;
}

#define MMX_HIP(call)                                                                             \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      return fail(                                                                                \
          e_ == hipErrorOutOfMemory ? MMX_ERR_OUT_OF_MEMORY : MMX_ERR_DEVICE,                     \
          std::string(#call) + ": " + hipGetErrorString(e_));                                     \
    }                                                                                             \
  } while (0)

// ---- profiling zones: roctx ranges around the ABI entry points and the phases of the explicit-Jacobian
// iteration, named after the reference's MT_PROFILE_EVENT zones of this path
// (momentum/solver/gauss_newton_solver.cpp:70,225,241,264,285,290; solver.cpp:51).  The roctx library is
// looked up at run time (librocprofiler-sdk-roctx.so, then libroctx64.so) so that libmmx_hip.so has no
// link-time dependency on a profiler; without it, or with MMX_NO_ROCTX set, a zone costs one branch.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (getenv("MMX_NO_ROCTX") != nullptr) {
      return;
    }
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
      if (h != nullptr) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push != nullptr && pop != nullptr) {
          return;
        }
        push = nullptr, pop = nullptr;
      }
    }
  }
};
const Roctx& roctx() {
  static const Roctx r;
  return r;
}
struct Zone {
  bool open;
  explicit Zone(const char* name) : open(roctx().push != nullptr) {
    if (open) {
      roctx().push(name);
    }
  }
  Zone(const Zone&) = delete;
  Zone& operator=(const Zone&) = delete;
  ~Zone() {
    if (open) {
      roctx().pop();
    }
  }
};
#define MMX_ZONE_CAT2(a, b) a##b
#define MMX_ZONE_CAT(a, b) MMX_ZONE_CAT2(a, b)
#define MMX_ZONE(name) Zone MMX_ZONE_CAT(zone_, __LINE__)(name)

// RAII device buffer
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() {
    release();
  }
  void release() {
    if (p != nullptr) {
      (void)hipFree(p);
      p = nullptr;
      bytes = 0;
    }
  }
  hipError_t ensure(size_t n) {
    if (n <= bytes && p != nullptr) {
      return hipSuccess;
    }
    release();
    if (n == 0) {
      return hipSuccess;
    }
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) {
      bytes = n;
    } else {
      p = nullptr;
    }
    return e;
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

template <class T>
hipError_t upload(DevBuf& buf, const std::vector<T>& v) {
  hipError_t e = buf.ensure(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (e != hipSuccess || v.empty()) {
    return e;
  }
  return hipMemcpy(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

} // namespace

namespace mmx {
int32_t failWith(int32_t code, const std::string& msg) { // for the other translation units of the C ABI
  return fail(code, msg);
}
} // namespace mmx

struct mmx_rig {
  int32_t device = 0;
  int32_t J = 0, P = 0;
  // host copy of the descriptor (needed to rebuild tables when the enabled set changes)
  std::vector<int32_t> parent, ptOuter, ptInner;
  std::vector<float> preRot, offset, ptValue, ptOffsets;
  mmx::HostTables topo; // built with all parameters enabled
  DevBuf dParent, dPreRot, dOffset, dPtOuter, dPtInner, dPtValue, dPtOffsets, dLevelOrder, dLevelStart, dPtEll, dJumpParent, dPtRowRec;
  mmx::RigDev dev{};

  mmx_rig_desc desc() const {
    mmx_rig_desc d{};
    d.num_joints = J;
    d.num_params = P;
    d.parent = parent.data();
    d.pre_rotation = preRot.data();
    d.translation_offset = offset.data();
    d.pt_outer = ptOuter.data();
    d.pt_inner = ptInner.data();
    d.pt_value = ptValue.data();
    d.pt_offsets = ptOffsets.data();
    return d;
  }
};

struct mmx_problem {
  mmx_rig* rig = nullptr;
  int32_t B = 0, Kp = 0, Ko = 0, U = 0, M = 0;
  std::vector<int32_t> posParent, oriParent;
  // per-instance characters / constraint parents (mmx_problem_set_instance_rig / _parents)
  mmx::RigDev rigDev{}; // the rig as this problem's kernels see it: rig->dev + the per-instance pointers
  DevBuf oInstOffset, oInstPreRot, oInstPosParent, oInstOriParent, dJointTin;
  std::vector<int32_t> unionPos, unionOri; // joints that carry a position / orientation constraint in some element
  std::vector<int32_t> instPosHost, instOriHost; // the lists of the last mmx_problem_set_instance_parents call (host copies)
  bool instParentsFromHost = false;
  bool instPos = false, instOri = false;
  mmx::HostTables tables; // for the current enabled set
  mmx::FusedTables fused;
  DevBuf dUnitJoint, dUnitTin, dColStart, dColSources, dEnabledList, dJacRecs, dMultiCols, dZeroCols;
  DevBuf dSubSize, dPosUnitStart, dPosUnits, dSolveList, dSrcStart, dSrcs, dTerms, dTerms16, dComb, dDfsJoint, dLoadedPos;
  DevBuf dLimStart, dLimOf, dPairDest, dPairStart, dPairLim, dPairCols;
  mmx::FusedDev fdev{};
  // constraint payload: owned copies (host ingest) or borrowed device pointers
  DevBuf oPosOffset, oPosTarget, oPosWeight, oOriOffset, oOriTarget, oOriWeight, oMpTarget, oMpWeights, dLimits, dEnabledMask, oFnWeights;
  std::vector<mmx_parameter_limit> limits; // host copy (solve-list bookkeeping)
  // further joint-constraint blocks: host copy of the descriptors (payload pointers cleared) and parents
  struct JointBlockHost {
    int32_t type = 0, count = 0;
    std::vector<int32_t> parent;
    DevBuf oLocalPoint, oLocalDir, oGlobal, oPlaneD, oWeight; // owned copies of a host payload
  };
  std::vector<std::unique_ptr<JointBlockHost>> blocks;
  std::vector<mmx::JointBlockDev> blockDev;
  std::vector<mmx_ellipsoid_limit> ellipsoids; // host copy
  DevBuf dEllipsoids;
  DevBuf dBlocks, dGenJoint, dGenTin, dGenBlock;
  int32_t genRows = 0;
  bool haveConstraints = false;
  bool tablesDirty = false; // blocks / ellipsoids / limits changed and uploadProblemTables has not succeeded since
  mmx::ProblemDev dev{};
  // the same problem with the structurally zero columns dropped from the solve (explicit-Jacobian solver)
  int32_t solveN = 0;
  DevBuf dSolveListV1; // [solveN]
  DevBuf sHess2F64; // mmx_solve_f64 under MMX_STEP_TRUST_REGION: J^T J without damping
  DevBuf dSolveListF64; // [solveN] the same parameters in index order: the double instantiation follows the reference's column order
  std::vector<int32_t> solveListF64;
  std::vector<std::pair<int32_t, int32_t>> limitPairs; // (row, col) solve columns of the off-diagonal H entries limits add
  DevBuf dTileMasks, dTileList; // tile structure of the factor in elimination order (mmx::TileMasks): [64] masks, the non-zero tiles
  mmx::TileMasks tileMasks;
  std::vector<int32_t> solveListV1; // host copy
  // scratch
  DevBuf sJac, sRes, sErr, sJtj, sJtr, sFactor, sThetaInit, sTheta;
  DevBuf sTreeState, sDvec, sRhoVec, sRefState, sGenState; // wide systems refined through the tree (no dense J)
  DevBuf sJacColMajor; // column-major J of an MMX_LAYOUT_ROW_MAJOR request, before its transposition
  DevBuf sJacF64, sHessF64; // scratch of the double-precision solve
  // mmx_solve_f64's assembly list (mmx::F64AssemblyList): built on first use for the launch's chunking, dropped when the
  // tables change (uploadProblemTables)
  DevBuf dF64Groups, dF64Extra, dF64ChunkStart;
  int32_t f64ListUnitsPerChunk = 0; // 0: not built
  DevBuf sDone, sIters, sStatus, sLastErr, sFinalErr, sHist, sClk, sDelta, sStepIter, sLambda, sTrust;
  DevBuf sFusedArgs; // the one-launch solve's descriptors, stashed per solve (mmx_fused.hip, kArgLazy)
  DevBuf sDiag; // [B][4] diagnostics of the last single-precision solve (mmx_problem_solve_diagnostics)
  DevBuf sDiagAcc; // [B][4] the wide route's accumulators behind it (mmx::StepParams::diagAcc)
  DevBuf sDiagErr0; // [B] ... and the first iteration's error (mmx::StepParams::diagErr0)
  bool diagValid = false;
  DevBuf sThetaAuto, sAutoMap, sAutoCount; // MMX_PRECISION_AUTO: initial parameters, the elements to escalate, their number
  bool autoAbort = false; // ... its single-precision pass may leave a marked element after the first factorisation
  mmx_tuning tuning{}; // mmx_problem_set_tuning
  int32_t lastRoute = MMX_ROUTE_AUTO;
};

namespace {

// model parameters that can carry a non-zero Jacobian entry of a limit row
std::vector<int32_t> limitParameters(const mmx_rig* rig, const mmx_parameter_limit& lm) {
  std::vector<int32_t> out;
  auto add = [&](int32_t p) {
    if (std::find(out.begin(), out.end(), p) == out.end()) {
      out.push_back(p);
    }
  };
  auto addRow = [&](int32_t row) {
    for (int32_t k = rig->ptOuter[size_t(row)]; k < rig->ptOuter[size_t(row) + 1]; ++k) {
      add(rig->ptInner[size_t(k)]);
    }
  };
  switch (lm.type) {
    case MMX_LIMIT_MINMAX:
      add(lm.index0);
      break;
    case MMX_LIMIT_LINEAR:
    case MMX_LIMIT_HALFPLANE:
      add(lm.index0);
      add(lm.index1);
      break;
    case MMX_LIMIT_MINMAX_JOINT:
      addRow(lm.index0);
      break;
    case MMX_LIMIT_LINEAR_JOINT:
      addRow(lm.index1);
      addRow(lm.index0);
      break;
    default:
      break;
  }
  return out;
}

int32_t uploadProblemTables(mmx_problem* pb) {
  const mmx_rig* rig = pb->rig;
  MMX_HIP(hipSetDevice(rig->device));
  const mmx::HostTables& t = pb->tables;
  std::vector<int32_t> unitJoint(size_t(std::max(pb->U, 1))), unitTin(size_t(std::max(pb->U, 1)));
  for (int32_t c = 0; c < pb->Kp; ++c) {
    unitJoint[c] = pb->posParent[c];
  }
  for (int32_t c = 0; c < pb->Ko; ++c) {
    for (int k = 0; k < 3; ++k) {
      unitJoint[pb->Kp + 3 * c + k] = pb->oriParent[c];
    }
  }
  for (int32_t u = 0; u < pb->U; ++u) {
    unitTin[u] = t.tin[unitJoint[u]];
  }
  static_assert(sizeof(mmx::ColumnSource) == sizeof(mmx::ColumnSourceDev), "ColumnSource layouts must match");
  MMX_HIP(upload(pb->dUnitJoint, unitJoint));
  MMX_HIP(upload(pb->dUnitTin, unitTin));
  MMX_HIP(upload(pb->dJointTin, t.tin));
  pb->dev.jointTin = pb->dJointTin.as<int32_t>();
  {
    std::vector<int32_t> gj, gt, gb;
    for (size_t i = 0; i < pb->blocks.size(); ++i) {
      for (int32_t j : pb->blocks[i]->parent) {
        gj.push_back(j);
        gt.push_back(t.tin[size_t(j)]);
        gb.push_back(int32_t(i));
      }
    }
    MMX_HIP(upload(pb->dGenJoint, gj));
    MMX_HIP(upload(pb->dGenTin, gt));
    MMX_HIP(upload(pb->dGenBlock, gb));
    MMX_HIP(upload(pb->dBlocks, pb->blockDev));
    pb->dev.numBlocks = int32_t(pb->blocks.size());
    pb->dev.G = int32_t(gj.size());
    pb->dev.blocks = pb->dBlocks.as<mmx::JointBlockDev>();
    pb->dev.genJoint = pb->dGenJoint.as<int32_t>();
    pb->dev.genTin = pb->dGenTin.as<int32_t>();
    pb->dev.genBlock = pb->dGenBlock.as<int32_t>();
    std::vector<mmx::EllipsoidDev> ed(pb->ellipsoids.size());
    for (size_t i = 0; i < ed.size(); ++i) {
      const mmx_ellipsoid_limit& e = pb->ellipsoids[i];
      static_assert(sizeof(mmx_ellipsoid_limit) == 30 * 4 && sizeof(mmx::EllipsoidDev) == 32 * 4, "ellipsoid layouts");
      std::memcpy(&ed[i], &e, sizeof(e));
      ed[i].tinParent = t.tin[size_t(e.parent)];
      const bool onChain = t.tin[size_t(e.ellipsoid_parent)] <= t.tin[size_t(e.parent)] && t.tin[size_t(e.parent)] < t.tout[size_t(e.ellipsoid_parent)];
      ed[i].tinStop = onChain ? t.tin[size_t(e.ellipsoid_parent)] : -1;
    }
    MMX_HIP(upload(pb->dEllipsoids, ed));
    pb->dev.NE = int32_t(ed.size());
    pb->dev.ellipsoids = pb->dEllipsoids.as<mmx::EllipsoidDev>();
  }
  MMX_HIP(upload(pb->dColStart, t.colStart));
  MMX_HIP(upload(pb->dColSources, t.colSources));
  MMX_HIP(upload(pb->dEnabledList, t.enabledList));
  mmx::ProblemDev& d = pb->dev;
  d.B = pb->B;
  d.Kp = pb->Kp;
  d.Ko = pb->Ko;
  d.U = pb->U;
  d.M = pb->M;
  d.n = int32_t(t.enabledList.size());
  d.unitJoint = pb->dUnitJoint.as<int32_t>();
  d.unitTin = pb->dUnitTin.as<int32_t>();
  d.colStart = pb->dColStart.as<int32_t>();
  d.colSources = pb->dColSources.as<mmx::ColumnSourceDev>();
  d.enabledList = pb->dEnabledList.as<int32_t>();
  {
    std::vector<uint8_t> mask(size_t(rig->P), 0);
    for (int32_t p : t.enabledList) {
      mask[size_t(p)] = 1;
    }
    MMX_HIP(upload(pb->dEnabledMask, mask));
    d.enabledMask = pb->dEnabledMask.as<uint8_t>();
    d.rowsJoint = 3 * pb->U + pb->genRows + 3 * int32_t(pb->ellipsoids.size());
  }
  static_assert(sizeof(mmx::JacRec) == sizeof(mmx::JacRecDev) && sizeof(mmx::JacRec) == 32, "JacRec layouts must match");
  MMX_HIP(upload(pb->dJacRecs, t.jacRecs));
  MMX_HIP(upload(pb->dMultiCols, t.multiCols));
  MMX_HIP(upload(pb->dZeroCols, t.zeroCols));
  d.jacRecs = pb->dJacRecs.as<mmx::JacRecDev>();
  d.multiCols = pb->dMultiCols.as<int32_t>();
  d.zeroCols = pb->dZeroCols.as<int32_t>();
  d.numJacRecs = int32_t(t.jacRecs.size());
  d.numMultiCols = int32_t(t.multiCols.size());
  d.numZeroCols = int32_t(t.zeroCols.size());
  // tables of the fused solve kernel
  {
    const mmx_rig_desc rd = rig->desc();
    std::string err;
    // parameters touched by a limit or (all of them) by the model-parameter block stay in the solve list
    std::vector<uint8_t> force(size_t(rig->P), pb->dev.hasModel ? 1 : 0);
    for (const mmx_parameter_limit& lm : pb->limits) {
      for (int32_t p : limitParameters(rig, lm)) {
        force[size_t(p)] = 1;
      }
    }
    // joints that carry a further joint error function or an ellipsoid limit count like constrained joints for the
    // structure (solve list, source slots): point-like ones see every dof above them, fixed-axis ones rotations only
    std::vector<int32_t> structPos, structOri;
    if (pb->instPos) {
      structPos = pb->unionPos;
    }
    if (pb->instOri) {
      structOri = pb->unionOri;
    }
    for (const auto& h : pb->blocks) {
      const bool fixedAxis = h->type == MMX_JC_FIXED_AXIS_DIFF || h->type == MMX_JC_FIXED_AXIS_COS || h->type == MMX_JC_FIXED_AXIS_ANGLE;
      for (int32_t j : h->parent) {
        (fixedAxis ? structOri : structPos).push_back(j);
      }
    }
    for (const mmx_ellipsoid_limit& e : pb->ellipsoids) {
      structPos.push_back(e.parent);
    }
    const int32_t rc = mmx::buildFusedTables(
        &rd, t, pb->Kp, pb->posParent.data(), pb->Ko, pb->oriParent.data(), force.data(), structPos.empty() ? nullptr : &structPos,
        structOri.empty() ? nullptr : &structOri, pb->fused, err);
    if (rc != MMX_OK) {
      return fail(rc, err);
    }
    const mmx::FusedTables& f = pb->fused;
    // Column program of the J-assembly kernel, specialised to this problem: a column whose sources
    // have no constraint vector below them (or that is disabled) is structurally zero -- it moves
    // to the zero list, which the kernel writes BEFORE forward kinematics (those stores overlap the
    // FK prologue and cost no arithmetic); the rest as in buildHostTables.
    {
      std::vector<mmx::JacRec> recs;
      std::vector<int32_t> multi, zero;
      for (int32_t p = 0; p < rig->P; ++p) {
        const int32_t cnt = t.colStart[size_t(p) + 1] - t.colStart[size_t(p)];
        if (cnt == 0 || !f.structNonZero[size_t(p)]) {
          zero.push_back(p);
          continue;
        }
        const mmx::ColumnSource& cs = t.colSources[size_t(t.colStart[size_t(p)])];
        if (cnt == 1 && cs.dof >= 3 && cs.dof < 6) {
          recs.push_back(mmx::JacRec{cs.joint, cs.dof, p, cs.tin, cs.tout, cs.parent, cs.weight, 1});
        } else {
          multi.push_back(p);
        }
      }
      std::stable_sort(recs.begin(), recs.end(), [](const mmx::JacRec& a, const mmx::JacRec& b) {
        return a.joint != b.joint ? a.joint < b.joint : a.dof < b.dof;
      });
      while (!recs.empty() && recs.size() % 4 != 0) {
        recs.push_back(recs.back());
      }
      MMX_HIP(upload(pb->dJacRecs, recs));
      MMX_HIP(upload(pb->dMultiCols, multi));
      MMX_HIP(upload(pb->dZeroCols, zero));
      d.jacRecs = pb->dJacRecs.as<mmx::JacRecDev>();
      d.multiCols = pb->dMultiCols.as<int32_t>();
      d.zeroCols = pb->dZeroCols.as<int32_t>();
      d.numJacRecs = int32_t(recs.size());
      d.numMultiCols = int32_t(multi.size());
      d.numZeroCols = int32_t(zero.size());
    }
    MMX_HIP(upload(pb->dSubSize, f.subSize));
    MMX_HIP(upload(pb->dDfsJoint, f.dfsJoint));
    std::vector<int32_t> loadedPos;
    for (int32_t k = 0; k < rig->J; ++k) {
      if (f.posUnitStart[k + 1] > f.posUnitStart[k]) {
        loadedPos.push_back(k);
      }
    }
    MMX_HIP(upload(pb->dLoadedPos, loadedPos));
    MMX_HIP(upload(pb->dPosUnitStart, f.posUnitStart));
    MMX_HIP(upload(pb->dPosUnits, f.posUnits));
    MMX_HIP(upload(pb->dSolveList, f.solveList));
    // Source SLOTS of the fused kernel.  Column c of the compacted system keeps its first source in slot c
    // (the "primary" source: with it alone the slot index IS the column index, so the 16 x 16 tiles of
    // H = J^T J come straight out of matrix-core products of the per-slot moment contractions, phase G);
    // slots n .. NP-1 pad the last block (weight 0); the further sources of multi-source columns (shared
    // parameters) follow from slot NP on, in column order: extras of column c = slots
    // NP + xStart[c] .. NP + xStart[c+1] - 1.  (Integer bookkeeping, host side.)
    const int nbFused = mmx::fusedBlocksFor(int32_t(f.solveList.size())); // -1: beyond the fused instantiations (tables unused then)
    const int nbSlots = nbFused > 0 ? nbFused : std::max((int32_t(f.solveList.size()) + 15) / 16, 1);
    const int32_t NPs = 16 * nbSlots;
    std::vector<mmx::ColumnSource> slots(size_t(NPs), mmx::ColumnSource{0, 3, 0, 0, -1, 0.f});
    std::vector<int32_t> xStart(size_t(NPs) + 1, 0);
    std::vector<int32_t> slotOf(f.srcs.size(), -1), slotColumn; // source e -> slot ; slot -> column
    {
      const int32_t ncol = int32_t(f.solveList.size());
      slotColumn.assign(size_t(NPs), -1);
      for (int32_t c = 0; c < ncol; ++c) {
        xStart[size_t(c)] = int32_t(slots.size()) - NPs;
        for (int32_t e = f.srcStart[size_t(c)]; e < f.srcStart[size_t(c) + 1]; ++e) {
          if (e == f.srcStart[size_t(c)]) {
            slots[size_t(c)] = f.srcs[size_t(e)];
            slotOf[size_t(e)] = c;
            slotColumn[size_t(c)] = c;
          } else {
            slotOf[size_t(e)] = int32_t(slots.size());
            slotColumn.push_back(c);
            slots.push_back(f.srcs[size_t(e)]);
          }
        }
      }
      for (int32_t c = ncol; c <= NPs; ++c) {
        xStart[size_t(c)] = int32_t(slots.size()) - NPs;
      }
      while (slots.size() % 4 != 0) {
        slots.push_back(mmx::ColumnSource{0, 3, 0, 0, -1, 0.f});
        slotColumn.push_back(-1);
      }
    }
    MMX_HIP(upload(pb->dSrcStart, xStart));
    MMX_HIP(upload(pb->dSrcs, slots));
    mmx::FusedDev& fd = pb->fdev;
    fd.U = pb->U;
    fd.Kp = pb->Kp;
    fd.n = int32_t(f.solveList.size());
    fd.nsrc = int32_t(slots.size());
    fd.slotBase = NPs;
    fd.nnz = rig->ptOuter.back();
    fd.subSize = pb->dSubSize.as<int32_t>();
    fd.dfsJoint = pb->dDfsJoint.as<int32_t>();
    fd.loadedPos = pb->dLoadedPos.as<int32_t>();
    fd.numLoaded = int32_t(loadedPos.size());
    fd.unitJoint = pb->dUnitJoint.as<int32_t>();
    fd.posUnitStart = pb->dPosUnitStart.as<int32_t>();
    fd.posUnits = pb->dPosUnits.as<int32_t>();
    fd.solveList = pb->dSolveList.as<int32_t>();
    fd.srcStart = pb->dSrcStart.as<int32_t>();
    fd.srcs = pb->dSrcs.as<mmx::ColumnSourceDev>();
    // Structural term records of H for the pairs the matrix-core pass does not cover: entry (row, col),
    // row >= col, receives one term per pair (source a of row, source c of col) whose joints are in an
    // ancestor relation AND of which at least one is an extra source; the deeper source supplies the
    // moment contractions, the other one alpha / B (mmx_fused.hip phase G; weights are folded into the
    // per-slot tables, the record's weight word stays 1).  Entries are dealt to the 256 threads of a
    // workgroup in contiguous runs of roughly equal term count; a thread's records are stored interleaved
    // (record k of thread t at [k * 256 + t]) so that a wave reads them coalesced.
    {
      struct Term {
        uint32_t deep, anc;
        float w;
      };
      struct Entry {
        int32_t dest;
        std::vector<Term> terms;
      };
      std::vector<Entry> entries;
      for (int32_t row = 0; row < fd.n; ++row) {
        for (int32_t col = 0; col <= row; ++col) {
          Entry en;
          for (int32_t er = f.srcStart[row]; er < f.srcStart[row + 1]; ++er) {
            for (int32_t ec = f.srcStart[col]; ec < f.srcStart[col + 1]; ++ec) {
              if (er == f.srcStart[row] && ec == f.srcStart[col]) {
                continue; // primary x primary: the matrix-core pass
              }
              const mmx::ColumnSource &sa = f.srcs[er], &sc = f.srcs[ec];
              int32_t deep, anc;
              if (sc.tin <= sa.tin && sa.tin < sc.tout) {
                deep = slotOf[size_t(er)], anc = slotOf[size_t(ec)];
              } else if (sa.tin <= sc.tin && sc.tin < sa.tout) {
                deep = slotOf[size_t(ec)], anc = slotOf[size_t(er)];
              } else {
                continue;
              }
              en.terms.push_back(Term{uint32_t(deep), uint32_t(anc), 1.f});
            }
          }
          if (en.terms.empty()) {
            continue;
          }
          const int I = row >> 4, Jc = col >> 4, r = row & 15, c = col & 15;
          const int t = I * (I + 1) / 2 + Jc;
          en.dest = t * 256 + r * 16 + ((((c >> 2) ^ (r >> 2)) & 3) << 2) + (c & 3); // tileAddr()
          entries.push_back(std::move(en));
        }
      }
      if (fd.nsrc >= (1 << 12)) {
        return fail(MMX_ERR_UNSUPPORTED, "more than 4095 column sources");
      }
      // Entries with many terms (pairs of shared parameters with many sources) are split into chunks of at
      // most kCap terms: chunk 0 stores to the entry itself, every further chunk to a private
      // partial cell that one thread adds to the entry afterwards, in a fixed order.
      constexpr size_t kCap = 8; // = the records one trip of the kernel's loop consumes
      struct Run {
        int32_t dest; // >= 0: float offset in the tile region ; < 0: -(cell + 1) partial cell
        size_t entry, first, count;
      };
      std::vector<Run> runs;
      std::vector<int32_t> combDest, combFirst, combCount;
      int32_t numCells = 0;
      for (size_t e = 0; e < entries.size(); ++e) {
        const size_t nt = entries[e].terms.size();
        const size_t chunks = (nt + kCap - 1) / kCap;
        for (size_t c = 0; c < chunks; ++c) {
          const size_t first = c * kCap, count = std::min(kCap, nt - first);
          runs.push_back(Run{c == 0 ? entries[e].dest : -(numCells + int32_t(c)), e, first, count});
        }
        if (chunks > 1) {
          combDest.push_back(entries[e].dest);
          combFirst.push_back(numCells);
          combCount.push_back(int32_t(chunks - 1));
          numCells += int32_t(chunks - 1);
        }
      }
      // (chunk c >= 1 of an entry uses partial cell combFirst + c - 1)
      // longest-processing-time-first assignment of runs to the threads of a workgroup (deterministic): 256 for the
      // one-launch solve and the four-wave tree kernels, 1024 for the sixteen-wave treeNormalEquationsKernel
      std::vector<size_t> order(runs.size());
      for (size_t i = 0; i < order.size(); ++i) {
        order[i] = i;
      }
      std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return runs[x].count > runs[y].count; });
      auto deal = [&](int threads, std::vector<uint32_t>& inter) -> size_t {
        const size_t nThreads = size_t(threads);
        std::vector<std::vector<uint32_t>> recs(nThreads); // 4 words per record
        std::vector<size_t> load(nThreads, 0);
        for (size_t oi : order) {
          const Run& rn = runs[oi];
          int thread = 0;
          for (int t = 1; t < threads; ++t) {
            if (load[size_t(t)] < load[size_t(thread)]) {
              thread = t;
            }
          }
          const Entry& en = entries[rn.entry];
          uint32_t destWord;
          if (rn.dest >= 0) {
            destWord = uint32_t(rn.dest);
          } else {
            destWord = (1u << 30) | uint32_t(-rn.dest - 1);
          }
          for (size_t i = 0; i < rn.count; ++i) {
            const Term& tm = en.terms[rn.first + i];
            uint32_t x = tm.deep | (tm.anc << 12) | (1u << 26);
            if (i == 0) {
              x |= 1u << 24;
            }
            if (i + 1 == rn.count) {
              x |= 1u << 25;
            }
            uint32_t wbits;
            std::memcpy(&wbits, &tm.w, 4);
            recs[size_t(thread)].insert(recs[size_t(thread)].end(), {x, destWord, wbits, 0u});
          }
          load[size_t(thread)] += rn.count;
        }
        size_t rounds = 0;
        for (const auto& r : recs) {
          rounds = std::max(rounds, r.size() / 4);
        }
        rounds = (rounds + 7) & ~size_t(7); // the kernels consume 8 records per trip
        inter.assign(std::max<size_t>(rounds, 8) * size_t(threads) * 4, 0u);
        for (int t = 0; t < threads; ++t) {
          for (size_t k = 0; k < recs[size_t(t)].size() / 4; ++k) {
            for (int w = 0; w < 4; ++w) {
              inter[(k * size_t(threads) + size_t(t)) * 4 + size_t(w)] = recs[size_t(t)][4 * k + size_t(w)];
            }
          }
        }
        return rounds;
      };
      std::vector<uint32_t> inter, inter16;
      const size_t rounds = deal(256, inter), rounds16 = deal(1024, inter16);
      if (numCells > 7 * rig->J) { // the kernels park the cells in a first-moment array: kC1 (= 7) floats per joint
        return fail(MMX_ERR_UNSUPPORTED, "too many split H entries for the partial-cell scratch");
      }
      std::vector<int32_t> comb;
      for (size_t i = 0; i < combDest.size(); ++i) {
        comb.push_back(combDest[i]);
        comb.push_back(combFirst[i]);
        comb.push_back(combCount[i]);
      }
      MMX_HIP(upload(pb->dComb, comb));
      fd.comb = pb->dComb.as<int32_t>();
      fd.numComb = int32_t(combDest.size());
      fd.numCells = numCells;
      MMX_HIP(upload(pb->dTerms, inter));
      fd.gTerms = pb->dTerms.as<uint4>();
      fd.termRounds = int32_t(rounds);
      MMX_HIP(upload(pb->dTerms16, inter16));
      fd.gTerms16 = pb->dTerms16.as<uint4>();
      fd.termRounds16 = int32_t(rounds16);
    }
    // limits per solve column, and the limits that share an off-diagonal entry of H
    pb->limitPairs.clear();
    {
      std::vector<int32_t> colOf(size_t(rig->P), -1);
      for (int32_t c = 0; c < fd.n; ++c) {
        colOf[size_t(f.solveList[c])] = c;
      }
      std::vector<std::vector<int32_t>> per(size_t(std::max(fd.n, 1)));
      std::map<int32_t, std::vector<int32_t>> pairs; // tile-region offset -> limits
      std::map<int32_t, std::pair<int32_t, int32_t>> pairColumns;
      for (size_t l = 0; l < pb->limits.size(); ++l) {
        std::vector<int32_t> cols;
        for (int32_t p : limitParameters(rig, pb->limits[l])) {
          if (colOf[size_t(p)] >= 0) {
            cols.push_back(colOf[size_t(p)]);
          }
        }
        for (int32_t c : cols) {
          per[size_t(c)].push_back(int32_t(l));
        }
        for (size_t x = 0; x < cols.size(); ++x) {
          for (size_t y = x + 1; y < cols.size(); ++y) {
            const int32_t row = std::max(cols[x], cols[y]), col = std::min(cols[x], cols[y]);
            const int I = row >> 4, Jc = col >> 4, r = row & 15, c = col & 15;
            const int32_t dest = (I * (I + 1) / 2 + Jc) * 256 + r * 16 + ((((c >> 2) ^ (r >> 2)) & 3) << 2) + (c & 3);
            pairs[dest].push_back(int32_t(l));
            pairColumns[dest] = {row, col};
          }
        }
      }
      std::vector<int32_t> limStart(1, 0), limOf, pairDest, pairStart(1, 0), pairLim;
      for (int32_t c = 0; c < fd.n; ++c) {
        limOf.insert(limOf.end(), per[size_t(c)].begin(), per[size_t(c)].end());
        limStart.push_back(int32_t(limOf.size()));
      }
      std::vector<int32_t> pairCols;
      for (const auto& kv : pairs) {
        pairDest.push_back(kv.first);
        pairCols.push_back(pairColumns[kv.first].first);
        pairCols.push_back(pairColumns[kv.first].second);
        pb->limitPairs.push_back(pairColumns[kv.first]);
        pairLim.insert(pairLim.end(), kv.second.begin(), kv.second.end());
        pairStart.push_back(int32_t(pairLim.size()));
      }
      MMX_HIP(upload(pb->dPairCols, pairCols));
      fd.pairCols = pb->dPairCols.as<int32_t>();
      MMX_HIP(upload(pb->dLimStart, limStart));
      MMX_HIP(upload(pb->dLimOf, limOf));
      MMX_HIP(upload(pb->dPairDest, pairDest));
      MMX_HIP(upload(pb->dPairStart, pairStart));
      MMX_HIP(upload(pb->dPairLim, pairLim));
      fd.numLimits = int32_t(pb->limits.size());
      fd.limStart = pb->dLimStart.as<int32_t>();
      fd.limOf = pb->dLimOf.as<int32_t>();
      fd.numPairDests = int32_t(pairDest.size());
      fd.pairDest = pb->dPairDest.as<int32_t>();
      fd.pairStart = pb->dPairStart.as<int32_t>();
      fd.pairLim = pb->dPairLim.as<int32_t>();
    }
    pb->fdev.GT = pb->dev.G + int32_t(pb->ellipsoids.size());
    pb->fdev.genRows = pb->fdev.GT > 0 ? pb->dev.rowsJoint - 3 * pb->U : 0;
  }
  // Solve list of the explicit-Jacobian solver: an enabled parameter none of whose joint-parameter
  // rows has a constraint below it has a zero column in J, so its step is 0 (H_pp = lambda, g_p = 0)
  // and it can leave the dense system -- exactly, like the fused kernel's solve list.  Parameters
  // touched by a limit or the model-parameter prior stay.  (Integer bookkeeping on the host.)
  {
    const size_t J = size_t(rig->J);
    std::vector<uint8_t> anyBelow(J, 0), pointBelow(J, 0); // a constraint vector / a constraint POINT in the joint's subtree
    auto mark = [&](int32_t joint, bool point) {
      for (int32_t a = joint; a >= 0; a = rig->parent[size_t(a)]) {
        anyBelow[size_t(a)] = 1;
        if (point) {
          pointBelow[size_t(a)] = 1;
        }
      }
    };
    for (int32_t c = 0; c < pb->Kp; ++c) {
      mark(pb->posParent[size_t(c)], true);
    }
    for (int32_t c = 0; c < pb->Ko; ++c) {
      mark(pb->oriParent[size_t(c)], false);
    }
    if (pb->instPos) {
      for (int32_t j : pb->unionPos) {
        mark(j, true);
      }
    }
    if (pb->instOri) {
      for (int32_t j : pb->unionOri) {
        mark(j, false);
      }
    }
    for (const auto& h : pb->blocks) {
      const bool fixedAxis = h->type == MMX_JC_FIXED_AXIS_DIFF || h->type == MMX_JC_FIXED_AXIS_COS || h->type == MMX_JC_FIXED_AXIS_ANGLE;
      for (int32_t j : h->parent) {
        mark(j, !fixedAxis);
      }
    }
    for (const mmx_ellipsoid_limit& e : pb->ellipsoids) {
      mark(e.parent, true);
    }
    std::vector<uint8_t> keep(size_t(rig->P), pb->dev.hasModel ? 1 : 0);
    for (const mmx_parameter_limit& lm : pb->limits) {
      for (int32_t p : limitParameters(rig, lm)) {
        keep[size_t(p)] = 1;
      }
    }
    std::vector<int32_t> list;
    for (int32_t p : t.eliminationList) {
      bool nz = keep[size_t(p)] != 0;
      for (int32_t e = t.colStart[size_t(p)]; !nz && e < t.colStart[size_t(p) + 1]; ++e) {
        const mmx::ColumnSource& cs = t.colSources[size_t(e)];
        nz = (cs.dof >= 3 && cs.dof < 6) ? anyBelow[size_t(cs.joint)] != 0 : pointBelow[size_t(cs.joint)] != 0;
      }
      if (nz) {
        list.push_back(p);
      }
    }
    if (list.empty()) {
      list = t.eliminationList; // nothing to solve for: keep the plain system (all steps are zero)
    }
    MMX_HIP(upload(pb->dSolveListV1, list));
    pb->solveN = int32_t(list.size());
    pb->solveListV1 = list;
    std::sort(list.begin(), list.end());
    MMX_HIP(upload(pb->dSolveListF64, list));
    pb->solveListF64 = list;
    pb->f64ListUnitsPerChunk = 0;
  }
  // Tile structure of the wide solve's factor (HostTables::eliminationList): entry (row, col) of H can be non-zero when
  // a source joint of the one column is an ancestor-or-self of a source joint of the other (their columns of J overlap
  // only then) or when a limit couples the two parameters.  The further joint error functions / ellipsoid limits (rows
  // over two joint chains) and systems the tree kernels do not take keep the dense structure.
  {
    const mmx::FusedTables& f = pb->fused;
    const int32_t n = int32_t(f.solveList.size());
    const bool dense = f.solveList != pb->solveListV1 || n > 512 || pb->fdev.GT > 0 || n == 0;
    std::vector<uint8_t> related;
    if (!dense) {
      const size_t J = size_t(rig->J);
      std::vector<uint8_t> reach(size_t(n) * J, 0); // joints in an ancestor relation with some source joint of column c
      for (int32_t c = 0; c < n; ++c) {
        uint8_t* rc = reach.data() + size_t(c) * J;
        for (int32_t e = f.srcStart[size_t(c)]; e < f.srcStart[size_t(c) + 1]; ++e) {
          const mmx::ColumnSource& cs = f.srcs[size_t(e)];
          for (int32_t k = cs.tin; k < cs.tout; ++k) {
            rc[size_t(f.dfsJoint[size_t(k)])] = 1;
          }
          for (int32_t a = cs.parent; a >= 0; a = rig->parent[size_t(a)]) {
            rc[size_t(a)] = 1;
          }
        }
      }
      related.assign(size_t(n) * size_t(n), 0);
      for (int32_t row = 0; row < n; ++row) {
        for (int32_t col = 0; col < row; ++col) {
          const uint8_t* rc = reach.data() + size_t(col) * J;
          bool any = false;
          for (int32_t e = f.srcStart[size_t(row)]; !any && e < f.srcStart[size_t(row) + 1]; ++e) {
            any = rc[size_t(f.srcs[size_t(e)].joint)] != 0;
          }
          related[size_t(row) * size_t(n) + size_t(col)] = any ? 1 : 0;
        }
      }
      for (const auto& rcPair : pb->limitPairs) {
        related[size_t(rcPair.first) * size_t(n) + size_t(rcPair.second)] = 1;
      }
    }
    pb->tileMasks = mmx::eliminationTileMasks(std::min(n, 512), related, dense);
    std::vector<uint32_t> masks(96, 0u);
    uint32_t base = 0;
    for (int i = 0; i < 32; ++i) {
      masks[size_t(i)] = pb->tileMasks.rowMask[i];
      masks[size_t(32 + i)] = pb->tileMasks.colMask[i];
      masks[size_t(64 + i)] = base; // first slot of block column i in the column-compact numbering
      base += uint32_t(__builtin_popcount(pb->tileMasks.colMask[i]));
    }
    for (int k = 0; k < pb->tileMasks.NB && k < 32; ++k) { // [96 + slot]: the tile in that slot, I | k << 8 (the resident kernels' load lists)
      for (int I = k; I < 32; ++I) {
        if (pb->tileMasks.colMask[k] >> I & 1u) {
          masks.push_back(uint32_t(I) | uint32_t(k) << 8);
        }
      }
    }
    // ... and behind the slots the level schedule of the factorisation (TileMasks::levelSteps; [96 + tiles]: numSteps, then
    // four words per step)
    for (int32_t w : pb->tileMasks.levelSteps) {
      masks.push_back(uint32_t(w));
    }
    MMX_HIP(upload(pb->dTileMasks, masks));
    MMX_HIP(upload(pb->dTileList, pb->tileMasks.tiles));
    pb->fdev.tileList = pb->dTileList.as<int32_t>();
    pb->fdev.numTiles = int32_t(pb->tileMasks.tiles.size());
  }

  return MMX_OK;
}

bool fusedUsable(const mmx_problem* pb) {
  const int nb = mmx::fusedBlocksFor(pb->fdev.n);
  if (nb < 0) {
    return false;
  }
  // (the further joint-constraint blocks and ellipsoid limits ride along as a dense block of rows in LDS -- fdev.GT /
  // genRows -- while they fit; MMX_ROUTE_EXPLICIT_JACOBIAN sends them to the explicit-Jacobian kernels)
  return pb->rig->J < 4096 &&
      mmx::fusedLdsBytes(nb, pb->rig->J, pb->rig->P, pb->U, pb->fdev.nsrc, pb->fdev.n, pb->fdev.numCells, true, pb->fdev.GT, pb->fdev.genRows, true, mmx::fusedCsrFloats(pb->rig->J, pb->fdev.nnz)) +
          size_t(8) * size_t(pb->rig->J + pb->rig->P) <=
      160 * 1024;
}

// H and g of the explicit-Jacobian solver from the tree moments instead of the dense J (treeNormalEquationsKernel):
// the same solve list on both sides, everything within the kernels' LDS
bool treeNormalEquationsUsable(const mmx_problem* pb) {
  // (limit / model-parameter rows ride along: evaluated on the fly from theta like in the fused solve; the further joint
  // error functions and ellipsoid limits as a dense block of rows in LDS while it fits)
  return pb->U > 0 && pb->fdev.n > 0 && pb->fdev.n <= 512 && pb->fdev.nsrc < 4096 &&
      pb->fused.solveList == pb->solveListV1 &&
      mmx::treeNormalEquationsLdsBytes(pb->rig->J, pb->rig->P, pb->U, pb->fdev.nsrc, pb->fdev.n, pb->fdev.GT, pb->fdev.genRows) <= 160 * 1024 - 64 &&
      mmx::treeRefineLdsBytes(pb->rig->J, pb->rig->P, pb->U, pb->fdev.n, pb->fdev.genRows) <= 160 * 1024 - 64;
}

// profiling aid (one of the library's two environment reads; the other is MMX_NO_ROCTX, the marker switch): per-phase cycle
// counters of workgroup 0, printed to stderr after the solve.  Not on any product path: the clocked instantiation is a
// separate kernel.
// refinement steps a solve may take per iteration (mmx_tuning::max_refinement_steps: 0 = the default, three; -1 = none)
int32_t refineSteps(const mmx_problem* pb) {
  const int32_t m = pb->tuning.max_refinement_steps;
  return m == 0 ? 3 : (m < 0 ? 0 : (m > 3 ? 3 : m));
}

bool phaseClocksWanted() {
  static const bool wanted = getenv("MMX_PHASE_CLOCKS") != nullptr;
  return wanted;
}

// MMX_PHASE_STOP=<stamp> beside MMX_PHASE_CLOCKS: the clocked instantiation's workgroups end at that stamp of their first
// iteration (slot 31 of the clock array carries stamp + 1) -- the solve's outputs are then meaningless; for counter passes only
hipError_t armPhaseStop(long long* clk, hipStream_t s) {
  static const long long stop = getenv("MMX_PHASE_STOP") != nullptr ? atoll(getenv("MMX_PHASE_STOP")) + 1 : 0;
  return stop > 0 ? hipMemcpyAsync(clk + 31, &stop, sizeof(long long), hipMemcpyHostToDevice, s) : hipSuccess;
}

int32_t checkProblem(const mmx_problem* pb, bool needConstraints) {
  // hipGetLastError() is per thread and shared with every other user of the runtime in this
  // process (torch probes devices / pointers and leaves benign errors behind): start clean so that
  // the launchers below only ever report their own failures
  (void)hipGetLastError();
  if (pb == nullptr || pb->rig == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "problem handle is null");
  }
  if (needConstraints && !pb->haveConstraints && pb->U > 0) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_problem_set_constraints has not been called");
  }
  return MMX_OK;
}

} // namespace

extern "C" {

void mmx_gn_options_default(mmx_gn_options* o) {
  if (o == nullptr) {
    return;
  }
  o->min_iterations = 1; // SolverOptions (momentum/solver/solver.h:19-34)
  o->max_iterations = 2;
  o->threshold = 1.0f;
  o->regularization = 0.05f; // GaussNewtonSolverBaseOptions (gauss_newton_solver.h:17-33)
  o->do_line_search = 0;
  o->step_rule = MMX_STEP_GN_FIXED_LAMBDA;
  o->lm_lambda_min = 1e-6f;
  o->lm_lambda_max = 1e6f;
  o->lm_up = 4.0f;
  o->lm_down = 0.5f;
  o->trust_region_radius = 1.0f; // TrustRegionQROptions::trustRegionRadius_ (trust_region_qr.h:24)
  o->precision = MMX_PRECISION_F32;
  o->precision_bound = 1e-5f;
}

int32_t mmx_abi_version(void) {
  return MMX_ABI_VERSION;
}

const char* mmx_last_error(void) {
  return g_lastError.c_str();
}

int32_t mmx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int32_t mmx_host_tables(
    const mmx_rig_desc* desc,
    const uint8_t* enabled,
    int32_t* level,
    int32_t* tin,
    int32_t* tout,
    uint8_t* active_joint_params,
    int32_t* enabled_list,
    int32_t* num_enabled) {
  mmx::HostTables t;
  std::string err;
  const int32_t rc = mmx::buildHostTables(desc, enabled, t, err);
  if (rc != MMX_OK) {
    return fail(rc, err);
  }
  if (level) {
    std::memcpy(level, t.level.data(), sizeof(int32_t) * t.J);
  }
  if (tin) {
    std::memcpy(tin, t.tin.data(), sizeof(int32_t) * t.J);
  }
  if (tout) {
    std::memcpy(tout, t.tout.data(), sizeof(int32_t) * t.J);
  }
  if (active_joint_params) {
    std::memcpy(active_joint_params, t.activeJointParams.data(), t.activeJointParams.size());
  }
  if (enabled_list) {
    std::memcpy(enabled_list, t.enabledList.data(), sizeof(int32_t) * t.enabledList.size());
  }
  if (num_enabled) {
    *num_enabled = int32_t(t.enabledList.size());
  }
  return MMX_OK;
}

int32_t mmx_host_elimination_order(const mmx_rig_desc* desc, const uint8_t* enabled, int32_t* order, int32_t* num_enabled) {
  mmx::HostTables t;
  std::string err;
  const int32_t rc = mmx::buildHostTables(desc, enabled, t, err);
  if (rc != MMX_OK) {
    return fail(rc, err);
  }
  if (order) {
    std::memcpy(order, t.eliminationList.data(), sizeof(int32_t) * t.eliminationList.size());
  }
  if (num_enabled) {
    *num_enabled = int32_t(t.eliminationList.size());
  }
  return MMX_OK;
}

int32_t mmx_host_tile_structure(int32_t n, const uint8_t* related, uint32_t* row_mask, uint32_t* col_mask, int64_t* products) {
  if (n < 0 || n > 512 || (n > 0 && related == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_host_tile_structure: n outside 0..512 or related is null");
  }
  const std::vector<uint8_t> rel(related, related + size_t(n) * size_t(n));
  const mmx::TileMasks m = mmx::eliminationTileMasks(n, rel, false);
  for (int i = 0; i < 32; ++i) {
    if (row_mask) {
      row_mask[i] = m.rowMask[i];
    }
    if (col_mask) {
      col_mask[i] = m.colMask[i];
    }
  }
  if (products) {
    *products = m.products;
  }
  return MMX_OK;
}

int32_t mmx_host_tile_level_schedule(int32_t n, const uint8_t* related, int32_t* steps) {
  if (n < 0 || n > 512 || (n > 0 && related == nullptr) || steps == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_host_tile_level_schedule: n outside 0..512, or related / steps is null");
  }
  const std::vector<uint8_t> rel(related, related + size_t(n) * size_t(n));
  const mmx::TileMasks m = mmx::eliminationTileMasks(n, rel, false);
  if (m.levelSteps.size() > size_t(1 + 4 * 32)) {
    return fail(MMX_ERR_UNSUPPORTED, "mmx_host_tile_level_schedule: more than 32 steps");
  }
  std::copy(m.levelSteps.begin(), m.levelSteps.end(), steps);
  return MMX_OK;
}

int32_t mmx_host_f64_assembly_list(
    const mmx_rig_desc* desc,
    const int32_t* solve_list,
    int32_t n,
    const int32_t* pos_parent,
    int32_t num_pos,
    const int32_t* ori_parent,
    int32_t num_ori,
    int32_t units_per_chunk,
    uint32_t* groups,
    int32_t* num_groups,
    int32_t* extra,
    int32_t* num_extra,
    int32_t* chunk_start,
    int32_t* num_chunks) {
  if (n < 0 || num_pos < 0 || num_ori < 0 || units_per_chunk <= 0 || units_per_chunk > 64 || (n > 0 && solve_list == nullptr) ||
      (num_pos > 0 && pos_parent == nullptr) || (num_ori > 0 && ori_parent == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_host_f64_assembly_list: sizes / pointers");
  }
  mmx::HostTables t;
  std::string err;
  const int32_t rc = mmx::buildHostTables(desc, nullptr, t, err);
  if (rc != MMX_OK) {
    return fail(rc, err);
  }
  for (int32_t i = 0; i < n; ++i) {
    if (solve_list[i] < 0 || solve_list[i] >= t.P) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_host_f64_assembly_list: parameter index out of range");
    }
  }
  for (int32_t i = 0; i < num_pos + num_ori; ++i) {
    const int32_t j = i < num_pos ? pos_parent[i] : ori_parent[i - num_pos];
    if (j < 0 || j >= t.J) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_host_f64_assembly_list: joint index out of range");
    }
  }
  mmx::F64AssemblyListHost h;
  if (!mmx::buildF64AssemblyListHost(t, std::vector<int32_t>(solve_list, solve_list + n), pos_parent, num_pos, ori_parent, num_ori, units_per_chunk, h)) {
    return fail(MMX_ERR_UNSUPPORTED, "mmx_host_f64_assembly_list: more than 8191 sources of one column apply to one constraint");
  }
  const int32_t chunks = (num_pos + 3 * num_ori + units_per_chunk - 1) / units_per_chunk;
  // sizes first (null arrays: a query), then the arrays when the caller's capacities (passed in the counters) suffice
  const int32_t capG = num_groups ? *num_groups : 0, capE = num_extra ? *num_extra : 0;
  if (num_groups) {
    *num_groups = int32_t(h.groups.size() / 2);
  }
  if (num_extra) {
    *num_extra = int32_t(h.extra.size());
  }
  if (num_chunks) {
    *num_chunks = chunks;
  }
  if (groups != nullptr && capG >= int32_t(h.groups.size() / 2)) {
    std::copy(h.groups.begin(), h.groups.end(), groups);
  }
  if (extra != nullptr && capE >= int32_t(h.extra.size())) {
    std::copy(h.extra.begin(), h.extra.end(), extra);
  }
  if (chunk_start != nullptr) {
    std::copy(h.chunkStart.begin(), h.chunkStart.end(), chunk_start); // [2 chunks + 1]
  }
  return MMX_OK;
}

int32_t mmx_problem_tile_structure(mmx_problem* pb, uint32_t* row_mask, uint32_t* col_mask, int32_t* num_blocks, int32_t* num_tiles, int64_t* products) {
  if (pb == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "problem is null");
  }
  const mmx::TileMasks& m = pb->tileMasks;
  for (int i = 0; i < 32; ++i) {
    if (row_mask) {
      row_mask[i] = m.rowMask[i];
    }
    if (col_mask) {
      col_mask[i] = m.colMask[i];
    }
  }
  if (num_blocks) {
    *num_blocks = m.NB;
  }
  if (num_tiles) {
    *num_tiles = int32_t(m.tiles.size());
  }
  if (products) {
    *products = m.products;
  }
  return MMX_OK;
}

int32_t mmx_rig_create(const mmx_rig_desc* d, int32_t device, mmx_rig** out) {
  MMX_ZONE("mmx_rig_create");
  if (out == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "out is null");
  }
  *out = nullptr;
  mmx::HostTables topo;
  std::string err;
  int32_t rc = mmx::buildHostTables(d, nullptr, topo, err);
  if (rc != MMX_OK) {
    return fail(rc, err);
  }
  const int ndev = mmx_device_count();
  if (ndev <= 0) {
    return fail(MMX_ERR_NO_DEVICE, "no HIP device visible; the MI355X path has no CPU fallback");
  }
  if (device < 0 || device >= ndev) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "device index out of range");
  }
  mmx_rig* r = new (std::nothrow) mmx_rig();
  if (r == nullptr) {
    return fail(MMX_ERR_OUT_OF_MEMORY, "host allocation failed");
  }
  r->device = device;
  r->J = d->num_joints;
  r->P = d->num_params;
  const int32_t R = MMX_PARAMS_PER_JOINT * r->J;
  r->parent.assign(d->parent, d->parent + r->J);
  r->preRot.assign(d->pre_rotation, d->pre_rotation + 4 * r->J);
  r->offset.assign(d->translation_offset, d->translation_offset + 3 * r->J);
  r->ptOuter.assign(d->pt_outer, d->pt_outer + R + 1);
  const int32_t nnz = r->ptOuter[R];
  r->ptInner.assign(d->pt_inner, d->pt_inner + nnz);
  r->ptValue.assign(d->pt_value, d->pt_value + nnz);
  if (d->pt_offsets != nullptr) {
    r->ptOffsets.assign(d->pt_offsets, d->pt_offsets + R);
  } else {
    r->ptOffsets.assign(R, 0.f);
  }
  r->topo = std::move(topo);
  auto cleanup = [&](int32_t code) {
    delete r;
    return code;
  };
  if (hipSetDevice(device) != hipSuccess) {
    return cleanup(fail(MMX_ERR_DEVICE, "hipSetDevice failed"));
  }
#define UP(buf, vec)                                                             \
  do {                                                                           \
    hipError_t e_ = upload(buf, vec);                                            \
    if (e_ != hipSuccess) {                                                      \
      return cleanup(fail(MMX_ERR_DEVICE, std::string("rig upload: ") + hipGetErrorString(e_))); \
    }                                                                            \
  } while (0)
  UP(r->dParent, r->parent);
  UP(r->dPreRot, r->preRot);
  UP(r->dOffset, r->offset);
  UP(r->dPtOuter, r->ptOuter);
  UP(r->dPtInner, r->ptInner);
  UP(r->dPtValue, r->ptValue);
  UP(r->dPtOffsets, r->ptOffsets);
  UP(r->dLevelOrder, r->topo.levelOrder);
  UP(r->dLevelStart, r->topo.levelStart);
  // two-slot ELL copy of the parameter transform (one 16-byte record per joint-parameter row) and
  // the packed level/parent table of the J-assembly kernel
  std::vector<int32_t> ell(size_t(R) * 4, 0), jumpParent(size_t(r->J));
  bool ellOk = true;
  for (int32_t row = 0; row < R; ++row) {
    const int32_t k0 = r->ptOuter[row], k1 = r->ptOuter[row + 1];
    if (k1 - k0 > 2) {
      ellOk = false;
      break;
    }
    for (int s = 0; s < 2; ++s) {
      int32_t idx = -1, bits = 0;
      if (k0 + s < k1) {
        idx = r->ptInner[k0 + s];
        std::memcpy(&bits, &r->ptValue[k0 + s], 4);
      }
      ell[4 * size_t(row) + 2 * s] = idx;
      ell[4 * size_t(row) + 2 * s + 1] = bits;
    }
  }
  for (int32_t j = 0; j < r->J; ++j) {
    jumpParent[j] = ((r->parent[j] + 1) << 16) | (r->parent[j] + 1);
  }
  if (ellOk) {
    UP(r->dPtEll, ell);
  }
  UP(r->dJumpParent, jumpParent);
  // records of the non-empty transform rows (RigDev::ptRowRec)
  std::vector<int32_t> rowRec;
  if (R < 65536 && r->P <= 65535) { // (every field of a record is an unsigned 16-bit number: row, rows to the next record, first column, entries)
    std::vector<int32_t> rows;
    for (int32_t row = 0; row < R; ++row) {
      if (r->ptOuter[row + 1] > r->ptOuter[row]) {
        rows.push_back(row);
      }
    }
    for (size_t t = 0; t < rows.size(); ++t) {
      const int32_t row = rows[t], next = t + 1 < rows.size() ? rows[t + 1] : R;
      const int32_t k0 = r->ptOuter[row], cnt = r->ptOuter[row + 1] - k0;
      if (cnt > 65535 || next - row > 65535) { // (does not fit: no records at all, the kernels walk the CSR)
        rowRec.clear();
        break;
      }
      int32_t bits = 0;
      std::memcpy(&bits, &r->ptValue[k0], 4);
      rowRec.insert(rowRec.end(), {row | ((next - row) << 16), r->ptInner[k0] | (cnt << 16), bits, k0});
    }
    if (!rowRec.empty()) {
      UP(r->dPtRowRec, rowRec);
    }
  }
#undef UP
  mmx::RigDev& dv = r->dev;
  dv.ptRowRec = rowRec.empty() ? nullptr : r->dPtRowRec.as<int4>();
  dv.numRowRec = int32_t(rowRec.size() / 4);
  dv.ptEll = ellOk ? r->dPtEll.as<int4>() : nullptr;
  dv.jumpParent = r->dJumpParent.as<int32_t>();
  dv.jumpRounds = 0;
  while ((1 << dv.jumpRounds) < int32_t(r->topo.levelStart.size()) - 1) {
    ++dv.jumpRounds;
  }
  dv.J = r->J;
  dv.P = r->P;
  dv.R = R;
  dv.numLevels = int32_t(r->topo.levelStart.size()) - 1;
  dv.parent = r->dParent.as<int32_t>();
  dv.preRot = r->dPreRot.as<float>();
  dv.offset = r->dOffset.as<float>();
  dv.ptOuter = r->dPtOuter.as<int32_t>();
  dv.ptInner = r->dPtInner.as<int32_t>();
  dv.ptValue = r->dPtValue.as<float>();
  dv.ptOffsets = r->dPtOffsets.as<float>();
  dv.ptOffsetsNonZero = 0;
  for (float v : r->ptOffsets) {
    dv.ptOffsetsNonZero |= v != 0.f ? 1 : 0;
  }
  dv.levelOrder = r->dLevelOrder.as<int32_t>();
  dv.levelStart = r->dLevelStart.as<int32_t>();
  *out = r;
  return MMX_OK;
}

void mmx_rig_destroy(mmx_rig* rig) {
  if (rig != nullptr) {
    (void)hipSetDevice(rig->device);
    delete rig;
  }
}

int32_t mmx_rig_num_joints(const mmx_rig* rig) {
  return rig ? rig->J : 0;
}
int32_t mmx_rig_num_params(const mmx_rig* rig) {
  return rig ? rig->P : 0;
}

int32_t mmx_problem_create(
    mmx_rig* rig,
    int32_t batch,
    int32_t num_pos,
    const int32_t* pos_parent,
    int32_t num_ori,
    const int32_t* ori_parent,
    mmx_problem** out) {
  MMX_ZONE("mmx_problem_create");
  if (out == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "out is null");
  }
  *out = nullptr;
  if (rig == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "rig handle is null");
  }
  if (batch <= 0 || num_pos < 0 || num_ori < 0) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "batch must be > 0 and constraint counts >= 0");
  }
  if ((num_pos > 0 && pos_parent == nullptr) || (num_ori > 0 && ori_parent == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "constraint parent array is null");
  }
  for (int32_t c = 0; c < num_pos; ++c) {
    // MT_CHECK(jntIndex < skeleton_.joints.size()) (joint_error_function-inl.h:230)
    if (pos_parent[c] < 0 || pos_parent[c] >= rig->J) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "position constraint parent joint out of range");
    }
  }
  for (int32_t c = 0; c < num_ori; ++c) {
    if (ori_parent[c] < 0 || ori_parent[c] >= rig->J) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "orientation constraint parent joint out of range");
    }
  }
  mmx_problem* pb = new (std::nothrow) mmx_problem();
  if (pb == nullptr) {
    return fail(MMX_ERR_OUT_OF_MEMORY, "host allocation failed");
  }
  pb->rig = rig;
  pb->B = batch;
  pb->Kp = num_pos;
  pb->Ko = num_ori;
  pb->U = num_pos + 3 * num_ori;
  pb->M = 3 * pb->U;
  pb->dev.lossPos = mmx::LossDev{0, 2.f, 1.f, 1.f};
  pb->dev.lossOri = mmx::LossDev{0, 2.f, 1.f, 1.f};
  if (num_pos > 0) {
    pb->posParent.assign(pos_parent, pos_parent + num_pos);
  }
  if (num_ori > 0) {
    pb->oriParent.assign(ori_parent, ori_parent + num_ori);
  }
  pb->tables = rig->topo; // all parameters enabled
  pb->rigDev = rig->dev;
  pb->dev.wPos = 1.f;
  pb->dev.wOri = 1.f;
  const int32_t rc = uploadProblemTables(pb);
  if (rc != MMX_OK) {
    delete pb;
    return rc;
  }
  *out = pb;
  return MMX_OK;
}

void mmx_problem_destroy(mmx_problem* pb) {
  if (pb != nullptr) {
    if (pb->rig != nullptr) {
      (void)hipSetDevice(pb->rig->device);
    }
    delete pb;
  }
}

int32_t mmx_problem_num_rows(const mmx_problem* pb) {
  return pb ? pb->M : 0;
}
int32_t mmx_problem_batch(const mmx_problem* pb) {
  return pb ? pb->B : 0;
}

int32_t mmx_problem_set_tuning(mmx_problem* pb, const mmx_tuning* tuning) {
  int32_t rc = checkProblem(pb, false);
  if (rc != MMX_OK) {
    return rc;
  }
  if (tuning == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "tuning is null");
  }
  if (tuning->route < MMX_ROUTE_AUTO || tuning->route > MMX_ROUTE_EXPLICIT_JACOBIAN) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_tuning::route: unknown MMX_ROUTE_* value");
  }
  if (tuning->max_refinement_steps < -1 || tuning->max_refinement_steps > 3) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_tuning::max_refinement_steps: -1 (none), 0 (default) or 1..3");
  }
  if (!(tuning->mixed_tolerance >= 0.f) || tuning->mixed_tolerance > 1e-2f || tuning->mixed_max_cg < 0 || tuning->mixed_max_cg > 64) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_tuning::mixed_tolerance: 0 (default) or (0, 1e-2]; mixed_max_cg: 0 (default) or 1..64");
  }
  for (int32_t r : tuning->reserved) {
    if (r != 0) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_tuning::reserved must be zero");
    }
  }
  pb->tuning = *tuning;
  return MMX_OK;
}

int32_t mmx_problem_last_route(const mmx_problem* pb) {
  return pb != nullptr ? pb->lastRoute : MMX_ROUTE_AUTO;
}

int32_t mmx_problem_set_enabled(mmx_problem* pb, const uint8_t* enabled) {
  MMX_ZONE("mmx_problem_set_enabled");
  int32_t rc = checkProblem(pb, false);
  if (rc != MMX_OK) {
    return rc;
  }
  if (enabled == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "enabled is null");
  }
  // an unchanged mask leaves every table as it is (callers like Solver.solve set it before every solve; the rebuild
  // is host bookkeeping plus some thirty synchronous uploads)
  if (!pb->tablesDirty && pb->tables.enabled.size() == size_t(pb->rig->P)) {
    bool same = true;
    for (size_t p = 0; p < pb->tables.enabled.size() && same; ++p) {
      same = (pb->tables.enabled[p] != 0) == (enabled[p] != 0);
    }
    if (same) {
      return MMX_OK;
    }
  }
  const mmx_rig_desc d = pb->rig->desc();
  std::string err;
  mmx::HostTables t;
  rc = mmx::buildHostTables(&d, enabled, t, err);
  if (rc != MMX_OK) {
    return fail(rc, err);
  }
  pb->tables = std::move(t);
  return uploadProblemTables(pb);
}

int32_t mmx_problem_set_instance_rig(mmx_problem* pb, const float* translation_offset, const float* pre_rotation, int32_t memory, void* stream) {
  MMX_ZONE("mmx_problem_set_instance_rig");
  int32_t rc = checkProblem(pb, false);
  if (rc != MMX_OK) {
    return rc;
  }
  if (memory != MMX_MEM_HOST && memory != MMX_MEM_DEVICE) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "instance rig: unknown memory space");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t B = size_t(pb->B), J = size_t(pb->rig->J);
  auto bring = [&](DevBuf& buf, const float* src, size_t count, const float*& dst) -> hipError_t {
    if (src == nullptr) {
      dst = nullptr;
      return hipSuccess;
    }
    if (memory == MMX_MEM_DEVICE) {
      dst = src;
      return hipSuccess;
    }
    hipError_t e = buf.ensure(count * sizeof(float));
    if (e != hipSuccess) {
      return e;
    }
    dst = buf.as<float>();
    return hipMemcpyAsync(buf.p, src, count * sizeof(float), hipMemcpyHostToDevice, s);
  };
  const float *off = nullptr, *pre = nullptr;
  MMX_HIP(bring(pb->oInstOffset, translation_offset, B * J * 3, off));
  MMX_HIP(bring(pb->oInstPreRot, pre_rotation, B * J * 4, pre));
  if (memory == MMX_MEM_HOST) {
    MMX_HIP(hipStreamSynchronize(s)); // the caller may free its host arrays on return
  }
  pb->rigDev.instOffset = off;
  pb->rigDev.instPreRot = pre;
  return MMX_OK;
}

int32_t mmx_problem_set_instance_parents(mmx_problem* pb, const int32_t* pos_parent, const int32_t* ori_parent, int32_t memory, void* stream) {
  MMX_ZONE("mmx_problem_set_instance_parents");
  int32_t rc = checkProblem(pb, false);
  if (rc != MMX_OK) {
    return rc;
  }
  if (memory != MMX_MEM_HOST && memory != MMX_MEM_DEVICE) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "instance parents: unknown memory space");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t B = size_t(pb->B);
  // host view of the lists (device input is read back: the integer bookkeeping is host work)
  std::vector<int32_t> hp, ho;
  auto fetch = [&](const int32_t* src, size_t count, std::vector<int32_t>& dst) -> hipError_t {
    dst.clear();
    if (src == nullptr || count == 0) {
      return hipSuccess;
    }
    dst.resize(count);
    if (memory == MMX_MEM_HOST) {
      std::memcpy(dst.data(), src, count * sizeof(int32_t));
      return hipSuccess;
    }
    hipError_t e = hipMemcpyAsync(dst.data(), src, count * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
  };
  MMX_HIP(fetch(pos_parent, B * size_t(pb->Kp), hp));
  MMX_HIP(fetch(ori_parent, B * size_t(pb->Ko), ho));
  // validate everything before anything is modified (MT_CHECK joint_error_function-inl.h:230)
  std::vector<uint8_t> seenP(size_t(pb->rig->J), 0), seenO(size_t(pb->rig->J), 0);
  for (int32_t j : hp) {
    if (j < 0 || j >= pb->rig->J) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "instance parents: position constraint parent joint out of range");
    }
    seenP[size_t(j)] = 1;
  }
  for (int32_t j : ho) {
    if (j < 0 || j >= pb->rig->J) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "instance parents: orientation constraint parent joint out of range");
    }
    seenO[size_t(j)] = 1;
  }
  // unchanged lists (the C++ shell re-sends them with every constraint update): nothing to do
  if (memory == MMX_MEM_HOST && !pb->tablesDirty && pb->instParentsFromHost && hp == pb->instPosHost && ho == pb->instOriHost) {
    return MMX_OK;
  }
  auto place = [&](DevBuf& buf, const int32_t* src, const std::vector<int32_t>& host, const int32_t*& dst) -> hipError_t {
    if (host.empty()) {
      dst = nullptr;
      return hipSuccess;
    }
    if (memory == MMX_MEM_DEVICE) {
      dst = src;
      return hipSuccess;
    }
    hipError_t e = buf.ensure(host.size() * sizeof(int32_t));
    if (e != hipSuccess) {
      return e;
    }
    dst = buf.as<int32_t>();
    // on the caller's stream: ordered after the kernels in flight there that still read the previous lists
    e = hipMemcpyAsync(buf.p, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice, s);
    return e != hipSuccess ? e : hipStreamSynchronize(s);
  };
  const int32_t *dp = nullptr, *dq = nullptr;
  MMX_HIP(place(pb->oInstPosParent, pos_parent, hp, dp));
  MMX_HIP(place(pb->oInstOriParent, ori_parent, ho, dq));
  std::vector<int32_t> unionPos, unionOri;
  for (int32_t j = 0; j < pb->rig->J; ++j) {
    if (seenP[size_t(j)]) {
      unionPos.push_back(j);
    }
    if (seenO[size_t(j)]) {
      unionOri.push_back(j);
    }
  }
  // the integer bookkeeping (solve list, structurally zero columns) depends on the UNION of the batch's parents only:
  // new lists over the same joints need no rebuild -- the kernels read the lists themselves
  const bool sameTables = !pb->tablesDirty && (dp != nullptr) == pb->instPos && (dq != nullptr) == pb->instOri && unionPos == pb->unionPos && unionOri == pb->unionOri;
  pb->dev.instPosParent = dp;
  pb->dev.instOriParent = dq;
  pb->instPos = dp != nullptr;
  pb->instOri = dq != nullptr;
  pb->instParentsFromHost = memory == MMX_MEM_HOST;
  pb->instPosHost = std::move(hp);
  pb->instOriHost = std::move(ho);
  if (sameTables) {
    return MMX_OK;
  }
  pb->tablesDirty = true;
  pb->unionPos = std::move(unionPos);
  pb->unionOri = std::move(unionOri);
  rc = uploadProblemTables(pb); // solve list, structurally zero columns: over the union of the batch
  if (rc != MMX_OK) {
    return rc;
  }
  pb->tablesDirty = false;
  return MMX_OK;
}

int32_t mmx_problem_set_constraints_sized(mmx_problem* pb, const mmx_constraint_data* c, size_t dataSize, void* stream) {
  if (c == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "constraint data is null");
  }
  // the layout every ABI since 1 shares ends with the `memory` field
  if (dataSize < offsetof(mmx_constraint_data, memory) + sizeof(int32_t)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_problem_set_constraints_sized: data_size is smaller than any mmx_constraint_data");
  }
  if (dataSize > sizeof(mmx_constraint_data)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_problem_set_constraints_sized: data_size is larger than this library's mmx_constraint_data (caller built against a newer ABI)");
  }
  mmx_constraint_data full;
  std::memset(&full, 0, sizeof(full)); // fields the caller's header did not have: absent
  std::memcpy(&full, c, dataSize);
  return mmx_problem_set_constraints(pb, &full, stream);
}

int32_t mmx_problem_set_constraints(mmx_problem* pb, const mmx_constraint_data* c, void* stream) {
  MMX_ZONE("mmx_problem_set_constraints");
  int32_t rc = checkProblem(pb, false);
  if (rc != MMX_OK) {
    return rc;
  }
  if (c == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "constraint data is null");
  }
  if (pb->Kp > 0 && (!c->pos_offset || !c->pos_target || !c->pos_weight)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "position constraint arrays are null");
  }
  if (pb->Ko > 0 && (!c->ori_offset || !c->ori_target || !c->ori_weight)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "orientation constraint arrays are null");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  mmx::ProblemDev& d = pb->dev;
  const size_t B = size_t(pb->B);
  if (c->memory != MMX_MEM_DEVICE && c->memory != MMX_MEM_HOST) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "constraint data: unknown memory space");
  }
  auto makeLoss = [](float alpha, float cc) { // GeneralizedLossT ctor (generalized_loss.cpp:81-101), kEps = 1e-9
    mmx::LossDev l{0, 2.f, 1.f, 1.f};
    if (cc > 0.f) {
      l.alpha = alpha;
      l.invC2 = 1.f / (cc * cc);
      l.c = cc;
      const float kEps = 1e-9f;
      if (alpha >= 2.f - kEps && alpha <= 2.f + kEps) {
        l.type = 0;
      } else if (alpha >= 1.f - kEps && alpha <= 1.f + kEps) {
        l.type = 1;
      } else if (alpha >= -kEps && alpha <= kEps) {
        l.type = 2;
      } else if (alpha == MMX_LOSS_WELSCH) {
        l.type = 3;
      } else {
        l.type = 4;
      }
    }
    return l;
  };
  // ---- optional parameter-space blocks
  const int32_t P = pb->rig->P;
  if (c->num_limits < 0 || (c->num_limits > 0 && c->limits == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "limits: negative count or null array");
  }
  if ((c->model_target == nullptr) != (c->model_weights == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "model_target and model_weights must be given together");
  }
  for (int32_t l = 0; l < c->num_limits; ++l) {
    const mmx_parameter_limit& lm = c->limits[l];
    const bool model = lm.type == MMX_LIMIT_MINMAX || lm.type == MMX_LIMIT_LINEAR || lm.type == MMX_LIMIT_HALFPLANE;
    const bool joint = lm.type == MMX_LIMIT_MINMAX_JOINT || lm.type == MMX_LIMIT_LINEAR_JOINT;
    const bool two = lm.type == MMX_LIMIT_LINEAR || lm.type == MMX_LIMIT_HALFPLANE || lm.type == MMX_LIMIT_LINEAR_JOINT;
    if (lm.type == MMX_LIMIT_MINMAX_JOINT_PASSIVE) {
      continue; // LimitErrorFunctionT skips the passive type: no error, no row (limit_error_function.cpp:836-837,1051-1052)
    }
    if (!model && !joint) {
      return fail(
          MMX_ERR_UNSUPPORTED,
          "limit " + std::to_string(l) +
              ": MinMax, MinMaxJoint, MinMaxJointPassive (ignored like in the reference), Linear, LinearJoint and HalfPlane go here; "
              "Ellipsoid limits travel in mmx_constraint_data::ellipsoid_limits");
    }
    const int32_t bound = joint ? MMX_PARAMS_PER_JOINT * pb->rig->J : P;
    if (lm.index0 < 0 || lm.index0 >= bound || (two && (lm.index1 < 0 || lm.index1 >= bound))) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "limit " + std::to_string(l) + ": parameter index out of range"); // MT_CHECK :574-575,:611-612
    }
    if (limitParameters(pb->rig, lm).size() > 4) {
      return fail(MMX_ERR_UNSUPPORTED, "limit " + std::to_string(l) + ": more than four model parameters drive the limited joint parameters");
    }
  }
  if (c->num_function_weights < 0 || c->num_function_weights > 4 + MMX_MAX_JOINT_BLOCKS) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "num_function_weights out of range (columns: position, orientation, limits, model parameters, joint blocks)");
  }
  // ---- further joint-constraint blocks
  if (c->num_joint_blocks < 0 || c->num_joint_blocks > MMX_MAX_JOINT_BLOCKS || (c->num_joint_blocks > 0 && c->joint_blocks == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "joint_blocks: count out of range or null array");
  }
  bool blocksChanged = size_t(c->num_joint_blocks) != pb->blocks.size();
  int32_t genRows = 0, genCount = 0;
  for (int32_t i = 0; i < c->num_joint_blocks; ++i) {
    const mmx_joint_constraint_block& jb = c->joint_blocks[i];
    const std::string tag = "joint block " + std::to_string(i);
    if (jb.type < MMX_JC_PLANE || jb.type > MMX_JC_NORMAL) {
      return fail(MMX_ERR_UNSUPPORTED, tag + ": unknown error-function type");
    }
    if (jb.count < 0 || (jb.count > 0 && (jb.parent == nullptr || jb.global == nullptr || jb.weight == nullptr))) {
      return fail(MMX_ERR_INVALID_ARGUMENT, tag + ": negative count or null parent / global / weight array");
    }
    const bool plane = jb.type == MMX_JC_PLANE || jb.type == MMX_JC_HALF_PLANE;
    const bool fixedAxis = jb.type == MMX_JC_FIXED_AXIS_DIFF || jb.type == MMX_JC_FIXED_AXIS_COS || jb.type == MMX_JC_FIXED_AXIS_ANGLE;
    if (jb.count > 0 && ((!fixedAxis && jb.local_point == nullptr) || (!plane && jb.local_dir == nullptr) || (plane && jb.plane_d == nullptr))) {
      return fail(MMX_ERR_INVALID_ARGUMENT, tag + ": a payload array its error function needs is null");
    }
    for (int32_t k = 0; k < jb.count; ++k) {
      if (jb.parent[k] < 0 || jb.parent[k] >= pb->rig->J) {
        return fail(MMX_ERR_INVALID_ARGUMENT, tag + ": parent joint out of range"); // MT_CHECK joint_error_function-inl.h:230
      }
    }
    const bool three = jb.type == MMX_JC_AIM_DIST || jb.type == MMX_JC_AIM_DIR || jb.type == MMX_JC_FIXED_AXIS_DIFF;
    genRows += (three ? 3 : 1) * jb.count;
    genCount += jb.count;
    if (!blocksChanged) {
      const mmx_problem::JointBlockHost& h = *pb->blocks[size_t(i)];
      blocksChanged = h.type != jb.type || h.count != jb.count || !std::equal(h.parent.begin(), h.parent.end(), jb.parent);
    }
  }
  if (genCount > 1024) {
    return fail(MMX_ERR_UNSUPPORTED, "more than 1024 constraints in the further joint-constraint blocks");
  }
  // ---- Ellipsoid entries of the limit block
  if (c->num_ellipsoid_limits < 0 || c->num_ellipsoid_limits > 256 || (c->num_ellipsoid_limits > 0 && c->ellipsoid_limits == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "ellipsoid_limits: count out of range or null array");
  }
  for (int32_t i = 0; i < c->num_ellipsoid_limits; ++i) {
    const mmx_ellipsoid_limit& e = c->ellipsoid_limits[i];
    if (e.parent < 0 || e.parent >= pb->rig->J || e.ellipsoid_parent < 0 || e.ellipsoid_parent >= pb->rig->J) {
      return fail(MMX_ERR_INVALID_ARGUMENT, "ellipsoid limit " + std::to_string(i) + ": joint index out of range");
    }
  }
  // ---- everything above only validated; from here on the problem is modified.  Should a device
  // call fail half-way, tablesDirty keeps the derived tables from being trusted by the next call.
  if (c->memory == MMX_MEM_DEVICE) {
    d.posOffset = c->pos_offset;
    d.posTarget = c->pos_target;
    d.posWeight = c->pos_weight;
    d.oriOffset = c->ori_offset;
    d.oriTarget = c->ori_target;
    d.oriWeight = c->ori_weight;
  } else if (c->memory == MMX_MEM_HOST) {
    auto ingest = [&](DevBuf& buf, const float* src, size_t count, const float*& dst) -> hipError_t {
      if (count == 0) {
        dst = nullptr;
        return hipSuccess;
      }
      hipError_t e = buf.ensure(count * sizeof(float));
      if (e != hipSuccess) {
        return e;
      }
      dst = buf.as<float>();
      return hipMemcpyAsync(buf.p, src, count * sizeof(float), hipMemcpyHostToDevice, s);
    };
    MMX_HIP(ingest(pb->oPosOffset, c->pos_offset, B * pb->Kp * 3, d.posOffset));
    MMX_HIP(ingest(pb->oPosTarget, c->pos_target, B * pb->Kp * 3, d.posTarget));
    MMX_HIP(ingest(pb->oPosWeight, c->pos_weight, B * pb->Kp, d.posWeight));
    MMX_HIP(ingest(pb->oOriOffset, c->ori_offset, B * pb->Ko * 4, d.oriOffset));
    MMX_HIP(ingest(pb->oOriTarget, c->ori_target, B * pb->Ko * 4, d.oriTarget));
    MMX_HIP(ingest(pb->oOriWeight, c->ori_weight, B * pb->Ko, d.oriWeight));
    if (c->function_weights != nullptr && c->num_function_weights > 0) {
      MMX_HIP(ingest(pb->oFnWeights, c->function_weights, B * size_t(c->num_function_weights), d.fnWeights));
    }
    MMX_HIP(hipStreamSynchronize(s)); // the caller may free its host arrays on return
  }
  // per-element error-function weights (errorFunctionWeights[iBatch][...] of the batched driver)
  if (c->function_weights != nullptr && c->num_function_weights > 0) {
    if (c->memory == MMX_MEM_DEVICE) {
      d.fnWeights = c->function_weights;
    }
    d.fnCols = c->num_function_weights;
  } else {
    d.fnWeights = nullptr;
    d.fnCols = 0;
  }
  d.wPos = c->pos_function_weight;
  d.wOri = c->ori_function_weight;
  d.lossPos = makeLoss(c->pos_loss_alpha, c->pos_loss_c);
  d.lossOri = makeLoss(c->ori_loss_alpha, c->ori_loss_c);
  if (blocksChanged) {
    pb->tablesDirty = true;
    pb->blocks.clear();
    for (int32_t i = 0; i < c->num_joint_blocks; ++i) {
      auto h = std::make_unique<mmx_problem::JointBlockHost>();
      h->type = c->joint_blocks[i].type;
      h->count = c->joint_blocks[i].count;
      h->parent.assign(c->joint_blocks[i].parent, c->joint_blocks[i].parent + h->count);
      pb->blocks.push_back(std::move(h));
    }
  }
  pb->blockDev.assign(size_t(c->num_joint_blocks), mmx::JointBlockDev{});
  {
    int32_t first = 0, row = 3 * pb->U;
    for (int32_t i = 0; i < c->num_joint_blocks; ++i) {
      const mmx_joint_constraint_block& jb = c->joint_blocks[i];
      mmx_problem::JointBlockHost& h = *pb->blocks[size_t(i)];
      mmx::JointBlockDev& k = pb->blockDev[size_t(i)];
      k.type = jb.type;
      k.count = jb.count;
      k.first = first;
      k.rowStart = row;
      k.fw = jb.function_weight;
      k.loss = makeLoss(jb.loss_alpha, jb.loss_c);
      const size_t cnt = B * size_t(jb.count);
      auto bring = [&](DevBuf& buf, const float* src, size_t count, const float*& dst) -> hipError_t {
        if (src == nullptr || count == 0) {
          dst = nullptr;
          return hipSuccess;
        }
        if (c->memory == MMX_MEM_DEVICE) {
          dst = src;
          return hipSuccess;
        }
        hipError_t e = buf.ensure(count * sizeof(float));
        if (e != hipSuccess) {
          return e;
        }
        dst = buf.as<float>();
        return hipMemcpy(buf.p, src, count * sizeof(float), hipMemcpyHostToDevice);
      };
      MMX_HIP(bring(h.oLocalPoint, jb.local_point, 3 * cnt, k.localPoint));
      MMX_HIP(bring(h.oLocalDir, jb.local_dir, 3 * cnt, k.localDir));
      MMX_HIP(bring(h.oGlobal, jb.global, 3 * cnt, k.global));
      MMX_HIP(bring(h.oPlaneD, jb.plane_d, cnt, k.planeD));
      MMX_HIP(bring(h.oWeight, jb.weight, cnt, k.weight));
      const bool three = jb.type == MMX_JC_AIM_DIST || jb.type == MMX_JC_AIM_DIR || jb.type == MMX_JC_FIXED_AXIS_DIFF;
      first += jb.count;
      row += (three ? 3 : 1) * jb.count;
    }
  }
  pb->genRows = genRows;
  const bool ellipsoidsChanged = size_t(c->num_ellipsoid_limits) != pb->ellipsoids.size() ||
      (c->num_ellipsoid_limits > 0 &&
       std::memcmp(c->ellipsoid_limits, pb->ellipsoids.data(), size_t(c->num_ellipsoid_limits) * sizeof(mmx_ellipsoid_limit)) != 0);
  pb->ellipsoids.assign(c->ellipsoid_limits, c->ellipsoid_limits + c->num_ellipsoid_limits);
  std::vector<mmx_parameter_limit> kept; // the limits that produce a row (the passive type does not)
  for (int32_t l = 0; l < c->num_limits; ++l) {
    if (c->limits[l].type != MMX_LIMIT_MINMAX_JOINT_PASSIVE) {
      kept.push_back(c->limits[l]);
    }
  }
  const bool structureChanged = pb->tablesDirty || blocksChanged || ellipsoidsChanged || (c->model_target != nullptr) != (d.hasModel != 0) || kept.size() != pb->limits.size() ||
      (!kept.empty() && std::memcmp(kept.data(), pb->limits.data(), kept.size() * sizeof(mmx_parameter_limit)) != 0);
  pb->tablesDirty = pb->tablesDirty || structureChanged;
  pb->limits = std::move(kept);
  static_assert(sizeof(mmx_parameter_limit) == sizeof(mmx::LimitDev) && sizeof(mmx::LimitDev) == 32, "limit layouts must match");
  MMX_HIP(upload(pb->dLimits, pb->limits));
  d.NL = int32_t(pb->limits.size());
  d.limits = pb->dLimits.as<mmx::LimitDev>();
  d.wLimit = c->limit_function_weight;
  d.hasModel = c->model_target != nullptr ? 1 : 0;
  d.wModel = c->model_function_weight;
  if (d.hasModel) {
    if (c->memory == MMX_MEM_DEVICE) {
      d.mpTarget = c->model_target;
      d.mpWeights = c->model_weights;
    } else {
      MMX_HIP(pb->oMpTarget.ensure(B * P * sizeof(float)));
      MMX_HIP(pb->oMpWeights.ensure(B * P * sizeof(float)));
      MMX_HIP(hipMemcpy(pb->oMpTarget.p, c->model_target, B * P * sizeof(float), hipMemcpyHostToDevice));
      MMX_HIP(hipMemcpy(pb->oMpWeights.p, c->model_weights, B * P * sizeof(float), hipMemcpyHostToDevice));
      d.mpTarget = pb->oMpTarget.as<float>();
      d.mpWeights = pb->oMpWeights.as<float>();
    }
  } else {
    d.mpTarget = d.mpWeights = nullptr;
  }
  const int32_t rowsE = 3 * int32_t(pb->ellipsoids.size());
  pb->M = 3 * pb->U + pb->genRows + rowsE + d.NL + (d.hasModel ? P : 0);
  d.M = pb->M;
  d.rowsJoint = 3 * pb->U + pb->genRows + rowsE;
  if (!structureChanged) { // payload pointers / weights / losses of the blocks may still have changed
    MMX_HIP(upload(pb->dBlocks, pb->blockDev));
    d.blocks = pb->dBlocks.as<mmx::JointBlockDev>();
  }
  if (structureChanged) { // the fused kernel's solve list and limit tables depend on it
    rc = uploadProblemTables(pb);
    if (rc != MMX_OK) {
      return rc;
    }
    pb->tablesDirty = false;
  }
  pb->haveConstraints = true;
  return MMX_OK;
}

int32_t mmx_eval_jacobian(
    mmx_problem* pb,
    const float* theta_dev,
    float* jac_dev,
    float* res_dev,
    double* err_dev,
    int32_t layout,
    void* stream) {
  MMX_ZONE("mmx_eval_jacobian (initializeJacobianComputation + computeJacobianBlock)");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta is null");
  }
  if (layout != MMX_LAYOUT_COL_MAJOR && layout != MMX_LAYOUT_ROW_MAJOR) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown Jacobian layout");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (layout == MMX_LAYOUT_ROW_MAJOR && jac_dev != nullptr) {
    // assembled column-major (the layout the kernel's coalesced column stores are built for) into the
    // problem's scratch, then transposed per instance: one extra read + write of J
    MMX_HIP(pb->sJacColMajor.ensure(size_t(pb->B) * size_t(pb->M) * size_t(pb->rig->P) * sizeof(float)));
    MMX_HIP(mmx::launchFkJacobian(pb->rigDev, pb->dev, theta_dev, pb->sJacColMajor.as<float>(), res_dev, err_dev, nullptr, nullptr, s));
    MMX_HIP(mmx::launchTransposeJacobian(pb->sJacColMajor.as<float>(), jac_dev, pb->B, pb->M, pb->rig->P, s));
    return MMX_OK;
  }
  MMX_HIP(mmx::launchFkJacobian(pb->rigDev, pb->dev, theta_dev, jac_dev, res_dev, err_dev, nullptr, nullptr, s));
  return MMX_OK;
}

int32_t mmx_eval_jacobian_timed(
    mmx_problem* pb,
    const float* theta_dev,
    float* jac_dev,
    float* res_dev,
    double* err_dev,
    int32_t layout,
    void* stream,
    float* kernel_ms) {
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_dev == nullptr || jac_dev == nullptr || kernel_ms == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta / jac / kernel_ms is null");
  }
  if (layout != MMX_LAYOUT_COL_MAJOR) {
    return fail(MMX_ERR_UNSUPPORTED, "only MMX_LAYOUT_COL_MAJOR (the reference's layout) is implemented");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  MMX_HIP(hipEventCreate(&e0));
  hipError_t err = hipEventCreate(&e1);
  if (err == hipSuccess) {
    err = mmx::launchFkJacobian(
        pb->rigDev, pb->dev, theta_dev, jac_dev, res_dev, err_dev, nullptr, nullptr, static_cast<hipStream_t>(stream), e0, e1);
  }
  if (err == hipSuccess) {
    err = hipEventSynchronize(e1);
  }
  if (err == hipSuccess) {
    err = hipEventElapsedTime(kernel_ms, e0, e1);
  }
  (void)hipEventDestroy(e0);
  if (e1 != nullptr) {
    (void)hipEventDestroy(e1);
  }
  MMX_HIP(err);
  return MMX_OK;
}

int32_t mmx_debug_store_pattern(mmx_problem* pb, float* jac_dev, void* stream, float* kernel_ms) {
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (jac_dev == nullptr || kernel_ms == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "jac / kernel_ms is null");
  }
  if (pb->dev.M != pb->dev.rowsJoint || pb->dev.M != 3 * pb->dev.U) {
    return fail(MMX_ERR_UNSUPPORTED, "store pattern: position / orientation rows only");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  MMX_HIP(hipEventCreate(&e0));
  hipError_t err = hipEventCreate(&e1);
  if (err == hipSuccess) {
    err = mmx::launchStorePattern(jac_dev, pb->B, pb->dev.M, pb->rig->dev.P, mmx::fkJacobianWavesPerInstance(pb->rigDev, pb->dev, true), static_cast<hipStream_t>(stream), e0, e1);
  }
  if (err == hipSuccess) {
    err = hipEventSynchronize(e1);
  }
  if (err == hipSuccess) {
    err = hipEventElapsedTime(kernel_ms, e0, e1);
  }
  (void)hipEventDestroy(e0);
  if (e1 != nullptr) {
    (void)hipEventDestroy(e1);
  }
  MMX_HIP(err);
  return MMX_OK;
}

int32_t mmx_eval_skeleton_state(mmx_problem* pb, const float* theta_dev, float* state_dev, void* stream) {
  MMX_ZONE("mmx_eval_skeleton_state (SkeletonState::set)");
  int32_t rc = checkProblem(pb, false);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_dev == nullptr || state_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta / state is null");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  MMX_HIP(mmx::launchFkJacobian(
      pb->rigDev, pb->dev, theta_dev, nullptr, nullptr, nullptr, state_dev, nullptr, static_cast<hipStream_t>(stream), nullptr, nullptr, true));
  return MMX_OK;
}

namespace {
int32_t ensureStepScratch(mmx_problem* pb, bool needJacobian = true) {
  const size_t B = size_t(pb->B), M = size_t(pb->M), P = size_t(pb->rig->P), n = size_t(pb->dev.n);
  if (needJacobian) {
    MMX_HIP(pb->sJac.ensure(B * M * P * sizeof(float)));
  }
  MMX_HIP(pb->sRes.ensure(B * std::max<size_t>(M, 1) * sizeof(float)));
  MMX_HIP(pb->sErr.ensure(B * sizeof(double)));
  MMX_HIP(pb->sJtj.ensure(B * n * n * sizeof(float)));
  MMX_HIP(pb->sJtr.ensure(B * n * sizeof(float)));
  return MMX_OK;
}
} // namespace

int32_t mmx_eval_normal_equations(
    mmx_problem* pb,
    const float* theta_dev,
    float* jtj_dev,
    float* jtr_dev,
    double* err_dev,
    void* stream) {
  MMX_ZONE("mmx_eval_normal_equations (Get JtJ and JtR)");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_dev == nullptr || jtj_dev == nullptr || jtr_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta / jtj / jtr is null");
  }
  if (pb->dev.n > mmx::kMaxSolved) {
    return fail(MMX_ERR_UNSUPPORTED, "more than 2048 enabled parameters (kMaxModelParams)");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  rc = ensureStepScratch(pb);
  if (rc != MMX_OK) {
    return rc;
  }
  MMX_HIP(mmx::launchFkJacobian(
      pb->rigDev, pb->dev, theta_dev, pb->sJac.as<float>(), pb->sRes.as<float>(), err_dev, nullptr, nullptr, s));
  MMX_HIP(mmx::launchNormalEquations(
      pb->dev, pb->rig->P, pb->sJac.as<float>(), pb->sRes.as<float>(), jtj_dev, jtr_dev, nullptr, false, s));
  return MMX_OK;
}

int32_t mmx_debug_tree_normal_equations(mmx_problem* pb, const float* theta_dev, float* jtj_dev, float* jtr_dev, void* stream) {
  MMX_ZONE("mmx_debug_tree_normal_equations");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_dev == nullptr || jtj_dev == nullptr || jtr_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta / jtj / jtr is null");
  }
  if (!treeNormalEquationsUsable(pb) || pb->solveN != pb->dev.n) {
    return fail(MMX_ERR_UNSUPPORTED, "mmx_debug_tree_normal_equations: problem outside the tree-moment kernel's scope (or structurally zero columns present)");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  MMX_HIP(mmx::zeroAsync(jtj_dev, size_t(pb->B) * size_t(pb->dev.n) * size_t(pb->dev.n) * sizeof(float), s));
  MMX_HIP(mmx::launchTreeNormalEquations(pb->rigDev, pb->dev, pb->fdev, theta_dev, jtj_dev, jtr_dev, nullptr, nullptr, nullptr, nullptr, nullptr, false, s));
  return MMX_OK;
}

static int32_t solveImpl(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    double* step_history,
    void* stream);

int32_t mmx_solve(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    void* stream) {
  return solveImpl(pb, o, theta_dev, final_error, iterations, status, error_history, nullptr, nullptr, stream);
}

int32_t mmx_solve_with_history(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    void* stream) {
  return solveImpl(pb, o, theta_dev, final_error, iterations, status, error_history, parameter_history, nullptr, stream);
}

int32_t mmx_solve_with_step_history(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    double* step_history,
    void* stream) {
  if (step_history != nullptr && o != nullptr && o->step_rule != MMX_STEP_LM_SCHEDULE) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "mmx_solve_with_step_history: step_history is the LM schedule's (MMX_STEP_LM_SCHEDULE)");
  }
  return solveImpl(pb, o, theta_dev, final_error, iterations, status, error_history, parameter_history, step_history, stream);
}

int32_t mmx_problem_solve_diagnostics(mmx_problem* pb, float* diag_dev, void* stream) {
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (diag_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "diag_dev is null");
  }
  if (!pb->diagValid) {
    return fail(MMX_ERR_UNSUPPORTED, "mmx_problem_solve_diagnostics: no single-precision solve on the one-launch or wide route has run on this handle");
  }
  MMX_HIP(hipSetDevice(pb->rig->device));
  MMX_HIP(hipMemcpyAsync(diag_dev, pb->sDiag.p, size_t(pb->B) * 4 * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return MMX_OK;
}

static int32_t solveF32Impl(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    double* step_history,
    void* stream);
static int32_t solveF64Impl(
    mmx_problem* pb,
    const mmx_gn_options* o,
    double* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    double* step_history,
    const mmx::F64Select& select,
    void* stream);
static int32_t preflightF64(mmx_problem* pb);

// MMX_PRECISION_MIXED (and MMX_PRECISION_AUTO's second pass): does the mixed-precision instantiation of the one-launch solve
// take this problem with these options?  Position / orientation constraints and the parameter-space rows (limits on model / joint
// parameters, the model-parameter prior); not the further joint error functions / ellipsoid limits, not the trust region, the route
// not pinned away from the one-launch solve.
static bool mixedUsable(const mmx_problem* pb, const mmx_gn_options* o) {
  if (o == nullptr || pb == nullptr || pb->rig == nullptr) {
    return false;
  }
  const int32_t route = pb->tuning.route;
  return (route == MMX_ROUTE_AUTO || route == MMX_ROUTE_FUSED) && o->step_rule != MMX_STEP_TRUST_REGION && pb->fdev.GT == 0 &&
      pb->U > 0 && pb->fdev.n > 0 && fusedUsable(pb) &&
      mmx::fusedMixedUsable(pb->rig->J, pb->rig->P, pb->U, pb->fdev.nsrc, pb->fdev.n, pb->fdev.numCells);
}

static int32_t solveMixedImpl(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    double* step_history,
    const mmx::MixSelect& sel,
    void* stream) {
  MMX_ZONE("mmx_solve, mixed precision (SolverT<double>::solve around a single-precision factor)");
  if (o->max_iterations < 0 || o->min_iterations < 0) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "iteration counts must be >= 0");
  }
  if (o->step_rule != MMX_STEP_GN_FIXED_LAMBDA && o->step_rule != MMX_STEP_LM_SCHEDULE) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown step_rule");
  }
  if (o->do_line_search != MMX_LINE_SEARCH_NONE && o->do_line_search != MMX_LINE_SEARCH_GAUSS_NEWTON && o->do_line_search != MMX_LINE_SEARCH_DIRECTIONAL) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown do_line_search rule");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool escalation = sel.map != nullptr; // (the single-precision pass's outputs stay for the other elements)
  pb->lastRoute = MMX_ROUTE_FUSED;
  MMX_HIP(pb->sIters.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sFinalErr.ensure(B * sizeof(double)));
  mmx::SolveStateDev fst{};
  fst.iterations = iterations != nullptr ? iterations : pb->sIters.as<int32_t>();
  fst.status = status != nullptr ? status : pb->sStatus.as<int32_t>();
  fst.finalError = final_error != nullptr ? final_error : pb->sFinalErr.as<double>();
  fst.errorHistory = error_history;
  fst.paramHistory = parameter_history;
  fst.stepHistory = step_history;
  if (!escalation) {
    MMX_HIP(pb->sDiag.ensure(B * 4 * sizeof(float)));
    fst.diag = pb->sDiag.as<float>();
    pb->diagValid = true;
    if (step_history != nullptr && o->max_iterations > 0) {
      MMX_HIP(mmx::zeroAsync(step_history, B * size_t(o->max_iterations) * 2 * sizeof(double), s));
    }
    if (error_history != nullptr && o->max_iterations > 0) {
      MMX_HIP(mmx::zeroAsync(error_history, B * size_t(o->max_iterations) * sizeof(double), s));
    }
    if (parameter_history != nullptr && o->max_iterations > 0) {
      MMX_HIP(mmx::zeroAsync(parameter_history, B * size_t(o->max_iterations) * P * sizeof(float), s));
    }
  }
  fst.precisionBound = o->precision_bound > 0.f ? o->precision_bound : 1e-5f;
  mmx::FusedParams fp{};
  fp.lambda = o->regularization;
  fp.threshold = o->threshold;
  fp.minIterations = o->min_iterations;
  fp.maxIterations = o->max_iterations;
  fp.refine = 0;
  fp.doLineSearch = o->do_line_search;
  fp.stepRule = o->step_rule;
  fp.lmLambdaMin = o->lm_lambda_min;
  fp.lmLambdaMax = o->lm_lambda_max;
  fp.lmUp = o->lm_up;
  fp.lmDown = o->lm_down;
  fp.trustRadius = 1.f;
  fp.mixTol = pb->tuning.mixed_tolerance > 0.f ? pb->tuning.mixed_tolerance : 3e-9f;
  fp.mixMaxCg = pb->tuning.mixed_max_cg > 0 ? pb->tuning.mixed_max_cg : 12;
  long long* clk = nullptr;
  if (phaseClocksWanted()) { // profiling aid: per-phase cycles of block 0
    MMX_HIP(pb->sClk.ensure(32 * sizeof(long long)));
    MMX_HIP(hipMemsetAsync(pb->sClk.p, 0, 32 * sizeof(long long), s));
    clk = pb->sClk.as<long long>();
    MMX_HIP(armPhaseStop(clk, s));
  }
  MMX_HIP(mmx::launchFusedMixed(pb->rigDev, pb->dev, pb->fdev, theta_dev, fst, fp, sel, pb->B, clk, s));
  if (clk != nullptr) {
    long long h[32];
    MMX_HIP(hipMemcpyAsync(h, clk, sizeof(h), hipMemcpyDeviceToHost, s));
    MMX_HIP(hipStreamSynchronize(s));
    static const char* names[24] = {"-", "A-C fk + units (double)", "(reused state)", "D subtree sums", "E srcTables", "F g (double adjoint)",
                                    "G combine + pull", "H cholesky(tail)", "-", "-", "K update",
                                    "G extras: records", "G term records", "H.bc panel", "H.d mfma", "D own sums", "CG z0 solve + rz",
                                    "CG operator (double)", "CG decide + direction", "CG dots + update", "CG solve", "G extras: loads", "H.b load+barrier", "H.b chain"};
    long long tot = 0;
    for (int i = 0; i < 24; ++i) {
      tot += h[i];
    }
    fprintf(stderr, "[mmx phase clocks, MIXED, block 0, all iterations] total %lld\n", tot);
    for (int i = 0; i < 24; ++i) {
      fprintf(stderr, "  %-26s %10lld  %5.1f%%\n", names[i], h[i], 100.0 * double(h[i]) / double(tot > 0 ? tot : 1));
    }
  }
  return MMX_OK;
}

// mmx_gn_options::precision: the single-precision routes, the double instantiation on float parameters, or the first
// followed by the second on the elements it marked
static int32_t solveImpl(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    double* step_history,
    void* stream) {
  if (o == nullptr || o->precision == MMX_PRECISION_F32) {
    return solveF32Impl(pb, o, theta_dev, final_error, iterations, status, error_history, parameter_history, step_history, stream);
  }
  if (o->precision != MMX_PRECISION_F64 && o->precision != MMX_PRECISION_AUTO && o->precision != MMX_PRECISION_MIXED) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown precision (MMX_PRECISION_*)");
  }
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "options / theta is null");
  }
  const bool mixedOk = mixedUsable(pb, o);
  if (parameter_history != nullptr && !(o->precision == MMX_PRECISION_MIXED && mixedOk)) {
    return fail(MMX_ERR_UNSUPPORTED, "parameter_history is single precision's (MMX_PRECISION_F32) and the mixed-precision instantiation's: the double instantiation does not record it");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P);
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (o->precision == MMX_PRECISION_MIXED && mixedOk) {
    return solveMixedImpl(pb, o, theta_dev, final_error, iterations, status, error_history, parameter_history, step_history, mmx::MixSelect{nullptr, nullptr, nullptr}, stream);
  }
  // MMX_PRECISION_AUTO with the trust region: the rule's elements are marginal by construction (it starts from lambda = 1e-10, the
  // factor's damping floor engages on every element, and its Newton updates of lambda divide two fp32 quadratic forms: the
  // single-precision instantiation is held to 1e-4, tests/test_gpu_trust_region.py) and the mixed instantiation does not carry the
  // rule -- the policy's answer is the double kernel for every element, without a single-precision pass thrown away first
  const bool autoTrust = o->precision == MMX_PRECISION_AUTO && o->step_rule == MMX_STEP_TRUST_REGION;
  if (o->precision == MMX_PRECISION_F64 || o->precision == MMX_PRECISION_MIXED || autoTrust) { // (MIXED outside the mixed instantiation's scope: the double one)
    // every workgroup reads its element's parameters before it writes them: in place on the caller's float array
    return solveF64Impl(pb, o, nullptr, final_error, iterations, status, error_history, step_history, mmx::F64Select{nullptr, nullptr, theta_dev, theta_dev}, stream);
  }
  // (everything that can refuse or fail in the LATER stages is settled before the single-precision pass touches theta, status
  // and the histories: the double kernel's LDS budget, its scratch, the element lists)
  rc = preflightF64(pb);
  if (rc != MMX_OK) {
    return rc;
  }
  MMX_HIP(pb->sThetaAuto.ensure(B * P * sizeof(float)));
  MMX_HIP(pb->sAutoMap.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sAutoCount.ensure(sizeof(int32_t)));
  MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
  MMX_HIP(hipMemcpyAsync(pb->sThetaAuto.p, theta_dev, B * P * sizeof(float), hipMemcpyDeviceToDevice, s));
  int32_t* st = status != nullptr ? status : pb->sStatus.as<int32_t>();
  pb->autoAbort = mixedOk; // (the single-precision pass leaves a class its first factorisation marks: the second pass starts over anyway)
  rc = solveF32Impl(pb, o, theta_dev, final_error, iterations, st, error_history, nullptr, step_history, stream);
  pb->autoAbort = false;
  if (rc != MMX_OK) {
    return rc;
  }
  // a solve without the estimate (explicit-Jacobian route, the wide route's host-driven trust region): the cue stays the damping floor
  const int32_t mask = MMX_SOLVE_ERROR_MASK | MMX_SOLVE_PRECISION_SUSPECT | (pb->diagValid ? 0 : MMX_SOLVE_DAMPING_FLOORED);
  MMX_HIP(mmx::launchSelectSuspect(st, pb->B, mask, pb->sAutoMap.as<int32_t>(), pb->sAutoCount.as<int32_t>(), s));
  if (mixedOk) { // the marked elements again, from the initial parameters, by the mixed-precision instantiation ...
    rc = solveMixedImpl(
        pb, o, theta_dev, final_error, iterations, st, error_history, nullptr, step_history,
        mmx::MixSelect{pb->sAutoMap.as<int32_t>(), pb->sAutoCount.as<int32_t>(), pb->sThetaAuto.as<float>()}, stream);
    if (rc != MMX_OK) {
      return rc;
    }
    // ... and the ones whose conjugate gradients ran into the step limit there (the single-precision factor is no preconditioner
    // any more: cond x eps ~ 1 -- BASELINE configs[1] at lambda = 1e-5) or that failed, by the double instantiation
    MMX_HIP(mmx::launchSelectSuspect(st, pb->B, MMX_SOLVE_ERROR_MASK | MMX_SOLVE_PRECISION_SUSPECT, pb->sAutoMap.as<int32_t>(), pb->sAutoCount.as<int32_t>(), s, MMX_SOLVE_MIXED));
  }
  return solveF64Impl(
      pb, o, nullptr, final_error, iterations, st, error_history, step_history,
      mmx::F64Select{pb->sAutoMap.as<int32_t>(), pb->sAutoCount.as<int32_t>(), pb->sThetaAuto.as<float>(), theta_dev}, stream);
}

static int32_t solveF32Impl(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    float* parameter_history,
    double* step_history,
    void* stream) {
  MMX_ZONE("mmx_solve (SolverT::solve)");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (o == nullptr || theta_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "options / theta is null");
  }
  if (o->max_iterations < 0 || o->min_iterations < 0) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "iteration counts must be >= 0");
  }
  if (o->step_rule != MMX_STEP_GN_FIXED_LAMBDA && o->step_rule != MMX_STEP_LM_SCHEDULE && o->step_rule != MMX_STEP_TRUST_REGION) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown step_rule");
  }
  if (o->do_line_search != MMX_LINE_SEARCH_NONE && o->do_line_search != MMX_LINE_SEARCH_GAUSS_NEWTON &&
      o->do_line_search != MMX_LINE_SEARCH_DIRECTIONAL) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown do_line_search rule");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P);
  mmx::ProblemDev ds = pb->dev; // the explicit-Jacobian solver's view: structurally zero columns dropped
  ds.n = pb->solveN;
  ds.enabledList = pb->dSolveListV1.as<int32_t>();
  const int n = ds.n;
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Systems of 129-224 solved parameters fit the fused solve only with one workgroup per CU (its instantiations for ten,
  // twelve and fourteen 16-blocks: 55-105 KB of tiles): when the wide path's tree kernels cover the problem it is the
  // faster route since its factor is tile-sparse (72-joint humanoid with 219 parameters, random enabled subsets, B = 4096,
  // scripts/route_crossover.py, wide against fused solves/s: n = 105: 5.0e5 / 5.0e5, 126: 4.6e5 / 4.6e5, 136: 4.5e5 / 3.7e5,
  // 154: 4.2e5 / 3.7e5, 166: 3.9e5 / 2.8e5, 183: 3.5e5 / 2.7e5, 219: 3.3e5 / 2.2e5) -- up to eight blocks (two or three
  // workgroups per CU) the fused solve stays.  mmx_tuning::route pins either.
  const int32_t route = pb->tuning.route;
  const bool forceWide = route == MMX_ROUTE_WIDE;
  const bool trust = o->step_rule == MMX_STEP_TRUST_REGION;
  // (the trust region's re-solve loops live in the one-launch solve; the wide route drives them from the host, several
  // small kernels per trust step: it takes the rule only where the fused solve cannot -- more than 224 solved
  // parameters, further joint error functions / ellipsoid limits -- or when pinned)
  const bool trustNeedsWide = trust && (!fusedUsable(pb) || pb->fdev.GT > 0);
  const bool preferWide = (forceWide || trustNeedsWide || (!trust && mmx::fusedBlocksFor(pb->fdev.n) >= 10)) &&
      treeNormalEquationsUsable(pb) && route != MMX_ROUTE_FUSED && route != MMX_ROUTE_EXPLICIT_JACOBIAN;
  const bool legacy = route == MMX_ROUTE_EXPLICIT_JACOBIAN;
  const bool takeFused = fusedUsable(pb) && !legacy && !preferWide && !(pb->fdev.GT > 0 && o->step_rule == MMX_STEP_TRUST_REGION);
  if (route == MMX_ROUTE_FUSED && !takeFused) {
    return fail(MMX_ERR_UNSUPPORTED, "MMX_ROUTE_FUSED: the problem does not fit the one-launch solve (more than 224 solved parameters, or its tables beyond the LDS budget)");
  }
  if (route == MMX_ROUTE_WIDE && !preferWide) {
    return fail(MMX_ERR_UNSUPPORTED, "MMX_ROUTE_WIDE: the problem is outside the tree kernels' scope");
  }
  if (takeFused) {
    pb->lastRoute = MMX_ROUTE_FUSED;
    // fused path: the whole SolverT::solve loop in one launch, one workgroup per instance
    MMX_HIP(pb->sIters.ensure(B * sizeof(int32_t)));
    MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
    MMX_HIP(pb->sFinalErr.ensure(B * sizeof(double)));
    mmx::SolveStateDev fst{};
    fst.done = nullptr;
    fst.iterations = iterations != nullptr ? iterations : pb->sIters.as<int32_t>();
    fst.status = status != nullptr ? status : pb->sStatus.as<int32_t>();
    fst.lastError = nullptr;
    fst.finalError = final_error != nullptr ? final_error : pb->sFinalErr.as<double>();
    fst.errorHistory = error_history;
    fst.paramHistory = parameter_history;
    fst.stepHistory = step_history;
    MMX_HIP(pb->sDiag.ensure(B * 4 * sizeof(float)));
    fst.diag = pb->sDiag.as<float>();
    fst.precisionBound = o->precision_bound > 0.f ? o->precision_bound : 1e-5f;
    pb->diagValid = true;
    if (step_history != nullptr && o->max_iterations > 0) {
      MMX_HIP(mmx::zeroAsync(step_history, B * size_t(o->max_iterations) * 2 * sizeof(double), s));
    }
    if (error_history != nullptr && o->max_iterations > 0) {
      MMX_HIP(mmx::zeroAsync(error_history, B * size_t(o->max_iterations) * sizeof(double), s));
    }
    if (parameter_history != nullptr && o->max_iterations > 0) {
      MMX_HIP(mmx::zeroAsync(parameter_history, B * size_t(o->max_iterations) * P * sizeof(float), s));
    }
    mmx::FusedParams fp{};
    fp.lambda = o->regularization;
    fp.threshold = o->threshold;
    fp.minIterations = o->min_iterations;
    fp.maxIterations = o->max_iterations;
    fp.refine = refineSteps(pb);
    fp.doLineSearch = o->do_line_search;
    fp.stepRule = o->step_rule;
    fp.lmLambdaMin = o->lm_lambda_min;
    fp.lmLambdaMax = o->lm_lambda_max;
    fp.lmUp = o->lm_up;
    fp.lmDown = o->lm_down;
    fp.trustRadius = o->trust_region_radius > 0.f ? o->trust_region_radius : 1.f;
    fp.autoAbort = pb->autoAbort ? 1 : 0;
    long long* clk = nullptr;
    if (phaseClocksWanted()) { // profiling aid: per-phase cycles of block 0
      MMX_HIP(pb->sClk.ensure(32 * sizeof(long long)));
      MMX_HIP(hipMemsetAsync(pb->sClk.p, 0, 32 * sizeof(long long), s));
      clk = pb->sClk.as<long long>();
      MMX_HIP(armPhaseStop(clk, s));
    }
    {
      MMX_ZONE("fused solve: all iterations of GaussNewtonSolverT::doIteration in one launch");
      MMX_HIP(pb->sFusedArgs.ensure(mmx::fusedArgsBytes()));
      MMX_HIP(mmx::launchFusedSolve(pb->rigDev, pb->dev, pb->fdev, theta_dev, fst, fp, nullptr, nullptr, clk, pb->sFusedArgs.p, s));
    }
    if (clk != nullptr) {
      long long h[32];
      MMX_HIP(hipMemcpyAsync(h, clk, sizeof(h), hipMemcpyDeviceToHost, s));
      MMX_HIP(hipStreamSynchronize(s));
      static const char* names[24] = {"A jointParams", "B fk", "C units", "D subtree sums", "E srcTables", "F g + zero tiles",
                                      "G combine + pull", "H cholesky(tail)", "I solve", "J tail (d0 += rho)", "K update",
                                      "G extras: records", "G term records", "H.bc panel", "H.d mfma", "D own sums", "J jd",
                                      "J tangent+own", "J subtree", "J rho", "J solve", "G extras: loads", "H.b load+barrier", "H.b chain"};
      long long tot = h[24] + h[25] + h[26];
      for (int i = 0; i < 24; ++i) {
        tot += h[i];
      }
      fprintf(stderr, "[mmx phase clocks, block 0, all iterations] total %lld  (termRounds %d, numComb %d, nsrc %d, n %d)\n", tot, pb->fdev.termRounds, pb->fdev.numComb, pb->fdev.nsrc, pb->fdev.n);
      for (int i = 0; i < 24; ++i) {
        fprintf(stderr, "  %-16s %10lld  %5.1f%%\n", names[i], h[i], 100.0 * double(h[i]) / double(tot > 0 ? tot : 1));
      }
      fprintf(stderr, "  (of B fk: joint parameters %lld, local transforms %lld, pointer jumping %lld, axes = the rest of B)\n", h[26], h[24], h[25]);
    }
    return MMX_OK;
  }
  if (trust && !(preferWide && treeNormalEquationsUsable(pb) && route != MMX_ROUTE_EXPLICIT_JACOBIAN)) {
    return fail(MMX_ERR_UNSUPPORTED, "MMX_STEP_TRUST_REGION: available in the one-launch solve and on the wide route (tree kernels); this problem takes neither (MMX_ROUTE_EXPLICIT_JACOBIAN, or outside the tree kernels' scope)");
  }
  if (n > mmx::kMaxSolved) {
    return fail(MMX_ERR_UNSUPPORTED, "more than 2048 solved parameters (kMaxModelParams; 512 on the tree routes, the explicit-Jacobian route takes the rest)");
  }
  // Wide systems (the in-LDS Cholesky step does not fit) inside the tree kernels' scope: normal equations from the tree
  // moments, left-looking factor in HBM, refinement through the tree.  No dense J is written or read.
  // MMX_ROUTE_EXPLICIT_JACOBIAN (and problems outside the tree kernels' scope): dense J, J^T J on the matrix cores,
  // the refinement streams J.
  const bool wide = mmx::choleskyStepLdsBytes(ds.n, ds.M) > 160 * 1024; // (the in-LDS Cholesky step does not fit)
  const bool treeRefine = (wide || preferWide) && treeNormalEquationsUsable(pb) && route != MMX_ROUTE_EXPLICIT_JACOBIAN;
  if (route == MMX_ROUTE_WIDE && !treeRefine) {
    return fail(MMX_ERR_UNSUPPORTED, "MMX_ROUTE_WIDE: the tree-refined wide solve is not available for this problem");
  }
  pb->lastRoute = treeRefine ? MMX_ROUTE_WIDE : MMX_ROUTE_EXPLICIT_JACOBIAN;
  rc = ensureStepScratch(pb, !treeRefine);
  if (rc != MMX_OK) {
    return rc;
  }
  MMX_HIP(pb->sThetaInit.ensure(B * P * sizeof(float)));
  MMX_HIP(pb->sDone.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sIters.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sLastErr.ensure(B * sizeof(double)));
  MMX_HIP(pb->sFinalErr.ensure(B * sizeof(double)));
  mmx::SolveStateDev st{};
  st.done = pb->sDone.as<int32_t>();
  st.iterations = iterations != nullptr ? iterations : pb->sIters.as<int32_t>();
  st.status = status != nullptr ? status : pb->sStatus.as<int32_t>();
  st.lastError = pb->sLastErr.as<double>();
  st.finalError = final_error != nullptr ? final_error : pb->sFinalErr.as<double>();
  st.errorHistory = error_history;
  st.paramHistory = nullptr; // (explicit-Jacobian path: the history is copied after every iteration, below)
  st.stepHistory = step_history;
  pb->diagValid = false; // (the wide route's estimate: below, where its kernels are set up)
  if (step_history != nullptr && o->max_iterations > 0) {
    MMX_HIP(mmx::zeroAsync(step_history, B * size_t(o->max_iterations) * 2 * sizeof(double), s));
  }
  if (error_history != nullptr && o->max_iterations > 0) {
    MMX_HIP(mmx::zeroAsync(error_history, B * size_t(o->max_iterations) * sizeof(double), s));
  }
  MMX_HIP(hipMemcpyAsync(pb->sThetaInit.p, theta_dev, B * P * sizeof(float), hipMemcpyDeviceToDevice, s));
  const bool deferred = trust || o->do_line_search != 0 || o->step_rule == MMX_STEP_LM_SCHEDULE;
  const bool schedule = trust || o->step_rule == MMX_STEP_LM_SCHEDULE; // per-instance damping
  if (deferred) {
    MMX_HIP(pb->sDelta.ensure(B * size_t(std::max(n, 1)) * sizeof(float)));
    MMX_HIP(pb->sStepIter.ensure(B * sizeof(int32_t)));
    MMX_HIP(mmx::zeroAsync(pb->sStepIter.p, B * sizeof(int32_t), s));
  }
  if (schedule) {
    MMX_HIP(pb->sLambda.ensure(B * sizeof(float)));
  }
  // the precision estimate on the wide route (tree kernels + tile factor + finish stage); the trust region's host-driven
  // re-factorisations and the explicit-Jacobian route keep MMX_SOLVE_DAMPING_FLOORED as their cue
  const bool wideDiag = treeRefine && !trust && refineSteps(pb) > 0;
  if (wideDiag) {
    MMX_HIP(pb->sDiag.ensure(B * 4 * sizeof(float)));
    MMX_HIP(pb->sDiagAcc.ensure(B * 4 * sizeof(float)));
    MMX_HIP(pb->sDiagErr0.ensure(B * sizeof(float)));
    st.diag = pb->sDiag.as<float>();
    st.precisionBound = o->precision_bound > 0.f ? o->precision_bound : 1e-5f;
    pb->diagValid = true;
  }
  MMX_HIP(mmx::launchSolveInit(st, pb->B, schedule ? pb->sLambda.as<float>() : nullptr, o->regularization, s, wideDiag ? pb->sDiagAcc.as<float>() : nullptr));
  mmx::StepParams sp{};
  sp.diagAcc = wideDiag ? pb->sDiagAcc.as<float>() : nullptr;
  sp.diagErr0 = wideDiag ? pb->sDiagErr0.as<float>() : nullptr;
  sp.lambda = o->regularization;
  sp.threshold = o->threshold;
  sp.minIterations = o->min_iterations;
  sp.maxIterations = o->max_iterations;
  sp.refine = refineSteps(pb);
  sp.tileMasks = pb->dTileMasks.as<uint32_t>();
  sp.numTiles = int32_t(pb->tileMasks.tiles.size());
  sp.delta = deferred ? pb->sDelta.as<float>() : nullptr;
  sp.stepIter = deferred ? pb->sStepIter.as<int32_t>() : nullptr;
  sp.lambdaPer = schedule ? pb->sLambda.as<float>() : nullptr;
  sp.stepHistory = step_history;
  sp.doLineSearch = trust ? 0 : o->do_line_search; // (the trust region reads neither do_line_search nor regularization)
  sp.stepRule = o->step_rule;
  if (trust) {
    MMX_HIP(pb->sTrust.ensure(B * 6 * sizeof(int32_t) + 16));
    char* base = static_cast<char*>(pb->sTrust.p);
    sp.tr.lambda = reinterpret_cast<float*>(base);
    sp.tr.radius = reinterpret_cast<float*>(base + B * 4);
    sp.tr.phase = reinterpret_cast<int32_t*>(base + B * 8);
    sp.tr.newton = reinterpret_cast<int32_t*>(base + B * 12);
    sp.tr.step = reinterpret_cast<int32_t*>(base + B * 16);
    sp.tr.mask = reinterpret_cast<int32_t*>(base + B * 20);
    sp.tr.active = reinterpret_cast<int32_t*>(base + B * 24);
    MMX_HIP(mmx::launchTrustInit(sp.tr, pb->B, o->trust_region_radius > 0.f ? o->trust_region_radius : 1.f, s));
  }
  sp.lmLambdaMin = o->lm_lambda_min;
  sp.lmLambdaMax = o->lm_lambda_max;
  sp.lmUp = o->lm_up;
  sp.lmDown = o->lm_down;
  if (phaseClocksWanted()) {
    MMX_HIP(pb->sClk.ensure(32 * sizeof(long long)));
    MMX_HIP(hipMemsetAsync(pb->sClk.p, 0, 32 * sizeof(long long), s));
    sp.clk = pb->sClk.as<long long>();
  }
  float* factorScratch = nullptr; // wide systems: the left-looking Cholesky step keeps L in its own tile-major scratch
  if (wide || treeRefine) {
    MMX_HIP(pb->sFactor.ensure(size_t(B) * mmx::choleskyFactorFloats(ds.n) * sizeof(float)));
    factorScratch = pb->sFactor.as<float>();
  }
  const size_t NPs = (size_t(n) + 15) & ~size_t(15);
  float* genState = nullptr; // J_g of the further joint error functions / ellipsoid limits, tree kernels' hand-over
  if (treeRefine) {
    // (H travels tile-major on this route: NB (NB + 1) / 2 tiles of 256 floats, more than n^2 for small systems)
    MMX_HIP(pb->sJtj.ensure(std::max(size_t(B) * size_t(pb->dev.n) * size_t(pb->dev.n), size_t(B) * mmx::choleskyFactorFloats(ds.n)) * sizeof(float)));
    MMX_HIP(pb->sTreeState.ensure(size_t(B) * mmx::treeStateFloats(pb->rig->J, pb->fdev.U) * sizeof(float)));
    MMX_HIP(pb->sDvec.ensure(size_t(B) * NPs * sizeof(float)));
    MMX_HIP(pb->sRhoVec.ensure(size_t(B) * NPs * sizeof(float)));
    MMX_HIP(pb->sRefState.ensure(size_t(B) * sizeof(int32_t)));
    if (pb->fdev.GT > 0) {
      MMX_HIP(pb->sGenState.ensure(size_t(B) * mmx::treeGenStateFloats(pb->fdev.n, pb->fdev.genRows) * sizeof(float)));
      genState = pb->sGenState.as<float>();
    }
  }
  for (int it = 0; it < o->max_iterations; ++it) { // solver.cpp:89
    sp.iteration = it;
    MMX_ZONE("GaussNewtonSolverT::doIteration");
    if (treeRefine) {
      {
        MMX_ZONE("Get JtJ and JtR");
        MMX_HIP(mmx::launchTreeNormalEquations(
            pb->rigDev, pb->dev, pb->fdev, theta_dev, pb->sJtj.as<float>(), pb->sJtr.as<float>(), st.done, pb->sErr.as<double>(),
            pb->sTreeState.as<float>(), sp.clk != nullptr ? sp.clk + 8 : nullptr, genState, true, s));
      }
      MMX_ZONE("Dense gauss newton step");
      // factor + first solve, then up to three refinement rounds; an instance whose correction fell below 1e-3 of its step
      // applies the step and sits out the remaining rounds (its workgroups return at once)
      auto linearSolve = [&](const mmx::SolveStateDev& who) -> int32_t {
        MMX_HIP(mmx::launchCholeskyFactorTiled(
            ds, pb->rig->P, pb->sJtj.as<float>(), pb->sJtr.as<float>(), factorScratch, pb->sDvec.as<float>(), pb->sRefState.as<int32_t>(),
            pb->sErr.as<double>(), theta_dev, who, sp, s));
        for (int round = 0; round < sp.refine; ++round) {
          MMX_HIP(mmx::launchTreeRefine(
              pb->rigDev, pb->dev, pb->fdev, theta_dev, pb->sTreeState.as<float>(), genState, pb->sDvec.as<float>(), pb->sRhoVec.as<float>(), pb->sRefState.as<int32_t>(),
              sp.lambda, sp.lambdaPer, s));
          MMX_HIP(mmx::launchCholeskyFinishTiled(
              ds, pb->rig->P, factorScratch, pb->sDvec.as<float>(), pb->sRhoVec.as<float>(), pb->sRefState.as<int32_t>(), pb->sErr.as<double>(),
              theta_dev, who, sp, round, s));
        }
        return MMX_OK;
      };
      if (trust) {
        // TrustRegionQRT::doIteration (trust_region_qr.cpp:52-270) from the host: per pass, the instances whose damping
        // changed factor and solve, every instance with a step on the table decides (try it, or a Newton update of lambda
        // first), the ready ones run their trial.  At most ten trust steps of up to four solves each; the loop ends as
        // soon as no instance is left in the iteration (one 4-byte read-back per pass).
        MMX_ZONE("TrustRegionQR: trust steps");
        MMX_HIP(mmx::launchTrustBegin(sp.tr, st, sp.lambdaPer, pb->B, s));
        mmx::SolveStateDev masked = st;
        masked.done = sp.tr.mask;
        for (int pass = 0; pass < 40; ++pass) {
          rc = linearSolve(masked);
          if (rc != MMX_OK) {
            return rc;
          }
          MMX_HIP(mmx::launchTrustDecide(ds, factorScratch, pb->sJtr.as<float>(), pb->sErr.as<double>(), st, sp, s));
          MMX_HIP(mmx::zeroAsync(sp.tr.active, sizeof(int32_t), s));
          MMX_HIP(mmx::launchStepUpdate(pb->rigDev, ds, theta_dev, pb->sJtr.as<float>(), pb->sErr.as<double>(), sp, s));
          int32_t active = 0;
          MMX_HIP(hipMemcpyAsync(&active, sp.tr.active, sizeof(int32_t), hipMemcpyDeviceToHost, s));
          MMX_HIP(hipStreamSynchronize(s));
          if (active == 0) {
            break;
          }
        }
        MMX_HIP(mmx::launchTrustEnd(st, sp, pb->sErr.as<double>(), pb->B, s));
      } else {
        rc = linearSolve(st);
        if (rc != MMX_OK) {
          return rc;
        }
      }
    } else {
      {
        MMX_ZONE("Get JtJ and JtR");
        MMX_HIP(mmx::launchFkJacobian(
            pb->rigDev, pb->dev, theta_dev, pb->sJac.as<float>(), pb->sRes.as<float>(), pb->sErr.as<double>(), nullptr, st.done, s, nullptr, nullptr, true));
        MMX_HIP(mmx::launchNormalEquations(
            ds, pb->rig->P, pb->sJac.as<float>(), pb->sRes.as<float>(), pb->sJtj.as<float>(), pb->sJtr.as<float>(), st.done, wide, s));
      }
      {
        MMX_ZONE("Dense gauss newton step");
        MMX_HIP(mmx::launchCholeskyStep(
            ds, pb->rig->P, pb->sJac.as<float>(), pb->sRes.as<float>(), pb->sJtj.as<float>(), pb->sJtr.as<float>(), pb->sErr.as<double>(),
            theta_dev, st, sp, factorScratch, s));
      }
    }
    if (deferred && !trust) {
      MMX_ZONE("Line search");
      MMX_HIP(mmx::launchStepUpdate(pb->rigDev, ds, theta_dev, pb->sJtr.as<float>(), pb->sErr.as<double>(), sp, s));
    }
    if (parameter_history != nullptr) { // parameters after iteration `it`, every element (rows past an element's last iteration are zeroed at the end)
      MMX_HIP(hipMemcpy2DAsync(
          parameter_history + size_t(it) * P, size_t(o->max_iterations) * P * sizeof(float), theta_dev, P * sizeof(float), P * sizeof(float), B,
          hipMemcpyDeviceToDevice, s));
    }
  }
  if (parameter_history != nullptr) {
    MMX_HIP(mmx::launchParamHistoryFinalize(parameter_history, st.iterations, pb->B, o->max_iterations, pb->rig->P, s));
  }
  MMX_HIP(mmx::launchSolveFinalize(theta_dev, pb->sThetaInit.as<float>(), pb->rig->P, st, pb->B, s, sp.diagAcc));
  if (sp.clk != nullptr) {
    long long h[16];
    MMX_HIP(hipMemcpyAsync(h, sp.clk, sizeof(h), hipMemcpyDeviceToHost, s));
    MMX_HIP(hipStreamSynchronize(s));
    static const char* names[8] = {"load H / fence", "factor (in-HBM: write-back + trailing)", "solve", "refine: w = r - J d", "refine: rho = J^T w", "refine: solve", "factor: panel load (in-HBM)", "factor: panel chain (in-HBM)"};
    fprintf(stderr, "[mmx phase clocks, choleskyStepKernel block 0, all iterations] n = %d\n", n);
    for (int i = 0; i < 8; ++i) {
      fprintf(stderr, "  %-40s %10lld\n", names[i], h[i]);
    }
    if (treeRefine) {
      static const char* tnames[8] = {"tree NE: load", "tree NE: A, B forward kinematics", "tree NE: C units + hand-over", "tree NE: D own sums",
                                      "tree NE: D subtree sums", "tree NE: E slot tables", "tree NE: F, G tiles", "tree NE: G extras"};
      for (int i = 0; i < 8; ++i) {
        fprintf(stderr, "  %-40s %10lld\n", tnames[i], h[8 + i]);
      }
    }
  }
  return MMX_OK;
}

namespace {
// mmx::F64AssemblyList for chunks of `uc` units (mmx::buildF64AssemblyListHost), uploaded
int32_t buildF64AssemblyList(mmx_problem* pb, int32_t uc) {
  mmx::F64AssemblyListHost h;
  if (!mmx::buildF64AssemblyListHost(pb->tables, pb->solveListF64, pb->posParent.data(), pb->Kp, pb->oriParent.data(), pb->Ko, uc, h)) {
    return fail(MMX_ERR_UNSUPPORTED, "mmx_solve_f64: more than 8191 sources of one column apply to one constraint");
  }
  if (h.groups.empty()) {
    h.groups.assign(2, 0u);
  }
  if (h.extra.empty()) {
    h.extra.push_back(0);
  }
  MMX_HIP(upload(pb->dF64Groups, h.groups));
  MMX_HIP(upload(pb->dF64Extra, h.extra));
  MMX_HIP(upload(pb->dF64ChunkStart, h.chunkStart));
  pb->f64ListUnitsPerChunk = uc;
  return MMX_OK;
}
} // namespace

int32_t mmx_solve_f64(
    mmx_problem* pb,
    const mmx_gn_options* o,
    double* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    void* stream) {
  if (theta_dev == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "options / theta is null");
  }
  return solveF64Impl(pb, o, theta_dev, final_error, iterations, status, error_history, nullptr, mmx::F64Select{nullptr, nullptr, nullptr, nullptr}, stream);
}

// theta_dev: double [B][P] in / out, or null when `select` carries float arrays (MMX_PRECISION_F64 / AUTO of mmx_solve)
// What the double kernel needs before it can run on this problem: its LDS budget and, in the scratch form, the problem's global
// scratch for J and H.  solveF64Impl calls it first; MMX_PRECISION_AUTO calls it before its single-precision pass, so that a
// refusal or an allocation failure of the later stage leaves the caller's theta / status / histories untouched.
static int32_t preflightF64(mmx_problem* pb) {
  const size_t B = size_t(pb->B), n = size_t(pb->solveN), M = size_t(pb->dev.rowsJoint);
  const int genRowsF64 = pb->dev.rowsJoint - 3 * pb->U;
  MMX_HIP(hipSetDevice(pb->rig->device));
  if (!mmx::solveF64IsResident(pb->rig->J, pb->rig->P, pb->U, pb->solveN, pb->dev.G + pb->dev.NE, genRowsF64)) {
    if (mmx::solveF64LdsBytes(pb->rig->J, pb->rig->P, pb->U, pb->solveN, pb->dev.G + pb->dev.NE, genRowsF64) > 160 * 1024) {
      return fail(MMX_ERR_UNSUPPORTED, "mmx_solve_f64: rig beyond the kernel's LDS budget");
    }
    MMX_HIP(pb->sJacF64.ensure(std::max<size_t>(B * n * M, 1) * sizeof(double)));
    MMX_HIP(pb->sHessF64.ensure(std::max<size_t>(B * n * n, 1) * sizeof(double)));
  }
  return MMX_OK;
}

static int32_t solveF64Impl(
    mmx_problem* pb,
    const mmx_gn_options* o,
    double* theta_dev,
    double* final_error,
    int32_t* iterations,
    int32_t* status,
    double* error_history,
    double* step_history,
    const mmx::F64Select& select,
    void* stream) {
  MMX_ZONE("mmx_solve_f64 (SolverT<double>::solve)");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (o == nullptr || (theta_dev == nullptr && select.thetaInit == nullptr)) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "options / theta is null");
  }
  if (o->max_iterations < 0 || o->min_iterations < 0) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "iteration counts must be >= 0");
  }
  if (o->step_rule != MMX_STEP_GN_FIXED_LAMBDA && o->step_rule != MMX_STEP_LM_SCHEDULE && o->step_rule != MMX_STEP_TRUST_REGION) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown step_rule");
  }
  if (o->do_line_search != MMX_LINE_SEARCH_NONE && o->do_line_search != MMX_LINE_SEARCH_GAUSS_NEWTON &&
      o->do_line_search != MMX_LINE_SEARCH_DIRECTIONAL) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "unknown do_line_search rule");
  }
  const size_t B = size_t(pb->B), n = size_t(pb->solveN), M = size_t(pb->dev.rowsJoint);
  const int genRowsF64 = pb->dev.rowsJoint - 3 * pb->U;
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool residentF64 = mmx::solveF64IsResident(pb->rig->J, pb->rig->P, pb->U, pb->solveN, pb->dev.G + pb->dev.NE, genRowsF64);
  rc = preflightF64(pb); // (the scratch form: dense J and H of every element in a global scratch of the problem)
  if (rc != MMX_OK) {
    return rc;
  }
  const bool trustF64 = o->step_rule == MMX_STEP_TRUST_REGION;
  if (trustF64) {
    MMX_HIP(pb->sHess2F64.ensure(std::max<size_t>(B * n * n, 1) * sizeof(double)));
  }
  MMX_HIP(pb->sIters.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sFinalErr.ensure(B * sizeof(double)));
  mmx::SolveStateDev st{};
  st.iterations = iterations != nullptr ? iterations : pb->sIters.as<int32_t>();
  st.status = status != nullptr ? status : pb->sStatus.as<int32_t>();
  st.finalError = final_error != nullptr ? final_error : pb->sFinalErr.as<double>();
  st.errorHistory = error_history;
  st.stepHistory = step_history;
  const bool escalation = select.map != nullptr; // (the single-precision solve's outputs stay for the other elements)
  if (error_history != nullptr && o->max_iterations > 0 && !escalation) {
    MMX_HIP(mmx::zeroAsync(error_history, B * size_t(o->max_iterations) * sizeof(double), s));
  }
  if (step_history != nullptr && o->max_iterations > 0 && !escalation) {
    MMX_HIP(mmx::zeroAsync(step_history, B * size_t(o->max_iterations) * 2 * sizeof(double), s));
  }
  mmx::FusedParams fp{};
  fp.lambda = o->regularization;
  fp.threshold = o->threshold;
  fp.minIterations = o->min_iterations;
  fp.maxIterations = o->max_iterations;
  fp.doLineSearch = o->do_line_search;
  fp.stepRule = o->step_rule;
  fp.lmLambdaMin = o->lm_lambda_min;
  fp.lmLambdaMax = o->lm_lambda_max;
  fp.lmUp = o->lm_up;
  fp.lmDown = o->lm_down;
  fp.trustRadius = o->trust_region_radius > 0.f ? o->trust_region_radius : 1.f;
  mmx::F64AssemblyList alist{nullptr, nullptr, nullptr, 0};
  if (residentF64 && pb->dev.instPosParent == nullptr && pb->dev.instOriParent == nullptr && pb->rig->J < 4096 && pb->solveN <= 4096) {
    const int32_t uc = mmx::solveF64ResidentChunkRows(pb->rig->J, pb->rig->P, pb->U, pb->solveN, pb->dev.G + pb->dev.NE, genRowsF64) / 3;
    if (uc > 0 && uc <= 64) {
      if (pb->f64ListUnitsPerChunk != uc) {
        rc = buildF64AssemblyList(pb, uc);
        if (rc != MMX_OK) {
          return rc;
        }
      }
      alist = mmx::F64AssemblyList{pb->dF64Groups.as<uint2>(), pb->dF64Extra.as<int32_t>(), pb->dF64ChunkStart.as<int32_t>(), uc};
    }
  }
  MMX_HIP(mmx::launchSolveF64(
      pb->rigDev, pb->dev, pb->dSolveListF64.as<int32_t>(), pb->solveN, theta_dev, st, fp, pb->sJacF64.as<double>(), pb->sHessF64.as<double>(), trustF64 ? pb->sHess2F64.as<double>() : nullptr, s, alist, select));
  return MMX_OK;
}

int32_t mmx_debug_fused_normal_equations(
    mmx_problem* pb,
    const float* theta_dev,
    float* jtj_dev,
    float* jtr_dev,
    int32_t* solve_list_host,
    int32_t* num_solved,
    void* stream) {
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  const int n = pb->fdev.n;
  if (num_solved != nullptr) {
    *num_solved = n;
  }
  if (solve_list_host != nullptr) {
    std::memcpy(solve_list_host, pb->fused.solveList.data(), sizeof(int32_t) * size_t(n));
  }
  if (theta_dev == nullptr || jtj_dev == nullptr || jtr_dev == nullptr) {
    return MMX_OK; // size query only
  }
  if (!fusedUsable(pb)) {
    return fail(MMX_ERR_UNSUPPORTED, "problem shape outside the fused kernel's instantiations");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P);
  MMX_HIP(hipSetDevice(pb->rig->device));
  hipStream_t s = static_cast<hipStream_t>(stream);
  MMX_HIP(pb->sTheta.ensure(B * P * sizeof(float)));
  MMX_HIP(pb->sIters.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sFinalErr.ensure(B * sizeof(double)));
  MMX_HIP(hipMemcpyAsync(pb->sTheta.p, theta_dev, B * P * sizeof(float), hipMemcpyDeviceToDevice, s));
  mmx::SolveStateDev fst{};
  fst.iterations = pb->sIters.as<int32_t>();
  fst.status = pb->sStatus.as<int32_t>();
  fst.finalError = pb->sFinalErr.as<double>();
  mmx::FusedParams fp{};
  fp.lambda = 0.05f;
  fp.threshold = 1.f;
  fp.minIterations = 1;
  fp.maxIterations = 1;
  fp.refine = 0;
  MMX_HIP(mmx::launchFusedSolve(pb->rigDev, pb->dev, pb->fdev, pb->sTheta.as<float>(), fst, fp, jtj_dev, jtr_dev, nullptr, nullptr, s));
  return MMX_OK;
}

int32_t mmx_solve_host(
    mmx_problem* pb,
    const mmx_gn_options* o,
    float* theta_host,
    double* final_error_host,
    int32_t* iterations_host,
    int32_t* status_host) {
  MMX_ZONE("mmx_solve_host");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_host == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta is null");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P);
  MMX_HIP(hipSetDevice(pb->rig->device));
  MMX_HIP(pb->sTheta.ensure(B * P * sizeof(float)));
  MMX_HIP(pb->sIters.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sFinalErr.ensure(B * sizeof(double)));
  MMX_HIP(hipMemcpy(pb->sTheta.p, theta_host, B * P * sizeof(float), hipMemcpyHostToDevice));
  rc = mmx_solve(pb, o, pb->sTheta.as<float>(), pb->sFinalErr.as<double>(), pb->sIters.as<int32_t>(), pb->sStatus.as<int32_t>(), nullptr, nullptr);
  if (rc != MMX_OK) {
    return rc;
  }
  MMX_HIP(hipDeviceSynchronize());
  MMX_HIP(hipMemcpy(theta_host, pb->sTheta.p, B * P * sizeof(float), hipMemcpyDeviceToHost));
  if (final_error_host) {
    MMX_HIP(hipMemcpy(final_error_host, pb->sFinalErr.p, B * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (iterations_host) {
    MMX_HIP(hipMemcpy(iterations_host, pb->sIters.p, B * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  if (status_host) {
    MMX_HIP(hipMemcpy(status_host, pb->sStatus.p, B * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return MMX_OK;
}

int32_t mmx_solve_f64_host(
    mmx_problem* pb,
    const mmx_gn_options* o,
    double* theta_host,
    double* final_error_host,
    int32_t* iterations_host,
    int32_t* status_host) {
  MMX_ZONE("mmx_solve_f64_host");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_host == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta is null");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P);
  MMX_HIP(hipSetDevice(pb->rig->device));
  MMX_HIP(pb->sTheta.ensure(B * P * sizeof(double)));
  MMX_HIP(pb->sIters.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sStatus.ensure(B * sizeof(int32_t)));
  MMX_HIP(pb->sFinalErr.ensure(B * sizeof(double)));
  MMX_HIP(hipMemcpy(pb->sTheta.p, theta_host, B * P * sizeof(double), hipMemcpyHostToDevice));
  rc = mmx_solve_f64(pb, o, pb->sTheta.as<double>(), pb->sFinalErr.as<double>(), pb->sIters.as<int32_t>(), pb->sStatus.as<int32_t>(), nullptr, nullptr);
  if (rc != MMX_OK) {
    return rc;
  }
  MMX_HIP(hipDeviceSynchronize());
  MMX_HIP(hipMemcpy(theta_host, pb->sTheta.p, B * P * sizeof(double), hipMemcpyDeviceToHost));
  if (final_error_host) {
    MMX_HIP(hipMemcpy(final_error_host, pb->sFinalErr.p, B * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (iterations_host) {
    MMX_HIP(hipMemcpy(iterations_host, pb->sIters.p, B * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  if (status_host) {
    MMX_HIP(hipMemcpy(status_host, pb->sStatus.p, B * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return MMX_OK;
}

int32_t mmx_eval_skeleton_state_host(mmx_problem* pb, const float* theta_host, float* state_host) {
  MMX_ZONE("mmx_eval_skeleton_state_host");
  int32_t rc = checkProblem(pb, false);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_host == nullptr || state_host == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta / state is null");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P), J = size_t(pb->rig->J);
  MMX_HIP(hipSetDevice(pb->rig->device));
  MMX_HIP(pb->sTheta.ensure(B * P * sizeof(float)));
  MMX_HIP(pb->sRes.ensure(B * J * 8 * sizeof(float)));
  MMX_HIP(hipMemcpy(pb->sTheta.p, theta_host, B * P * sizeof(float), hipMemcpyHostToDevice));
  rc = mmx_eval_skeleton_state(pb, pb->sTheta.as<float>(), pb->sRes.as<float>(), nullptr);
  if (rc != MMX_OK) {
    return rc;
  }
  MMX_HIP(hipDeviceSynchronize());
  MMX_HIP(hipMemcpy(state_host, pb->sRes.p, B * J * 8 * sizeof(float), hipMemcpyDeviceToHost));
  return MMX_OK;
}

int32_t mmx_eval_jacobian_host(
    mmx_problem* pb,
    const float* theta_host,
    float* jac_host,
    float* res_host,
    double* err_host,
    int32_t layout) {
  MMX_ZONE("mmx_eval_jacobian_host");
  int32_t rc = checkProblem(pb, true);
  if (rc != MMX_OK) {
    return rc;
  }
  if (theta_host == nullptr) {
    return fail(MMX_ERR_INVALID_ARGUMENT, "theta is null");
  }
  const size_t B = size_t(pb->B), P = size_t(pb->rig->P), M = size_t(pb->M);
  MMX_HIP(hipSetDevice(pb->rig->device));
  MMX_HIP(pb->sTheta.ensure(B * P * sizeof(float)));
  MMX_HIP(pb->sJac.ensure(B * M * P * sizeof(float)));
  MMX_HIP(pb->sRes.ensure(B * std::max<size_t>(M, 1) * sizeof(float)));
  MMX_HIP(pb->sErr.ensure(B * sizeof(double)));
  MMX_HIP(hipMemcpy(pb->sTheta.p, theta_host, B * P * sizeof(float), hipMemcpyHostToDevice));
  rc = mmx_eval_jacobian(pb, pb->sTheta.as<float>(), pb->sJac.as<float>(), pb->sRes.as<float>(), pb->sErr.as<double>(), layout, nullptr);
  if (rc != MMX_OK) {
    return rc;
  }
  MMX_HIP(hipDeviceSynchronize());
  if (jac_host) {
    MMX_HIP(hipMemcpy(jac_host, pb->sJac.p, B * M * P * sizeof(float), hipMemcpyDeviceToHost));
  }
  if (res_host) {
    MMX_HIP(hipMemcpy(res_host, pb->sRes.p, B * M * sizeof(float), hipMemcpyDeviceToHost));
  }
  if (err_host) {
    MMX_HIP(hipMemcpy(err_host, pb->sErr.p, B * sizeof(double), hipMemcpyDeviceToHost));
  }
  return MMX_OK;
}

} // extern "C"
