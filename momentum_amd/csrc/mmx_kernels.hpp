// mmx_kernels.hpp -- host-visible launch interface of mmx_kernels.hip.
#pragma once

#include <atomic>

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "mmx_device.hpp"

namespace mmx {

// per-instance state of SolverT::solve kept on the device (solver.cpp:84-119)
struct SolveStateDev {
  int32_t* done; // [B] 1 once the convergence test fired (instance idles afterwards)
  int32_t* iterations; // [B] errorHistory_.size()
  int32_t* status; // [B] MMX_SOLVE_*
  double* lastError; // [B] lastError_
  double* finalError; // [B] error_ (what solve() returns)
  double* errorHistory; // [B][maxIterations] or null
  float* paramHistory; // [B][maxIterations][P] or null: the parameters after iteration i (iterationHistory_["parameters"], solver.cpp:101-106)
  double* stepHistory; // [B][maxIterations][2] or null, MMX_STEP_LM_SCHEDULE: (lambda the iteration factored with, its gain ratio rho)
  float* diag; // [B][4] or null: numerical diagnostics of a single-precision solve (mmx_problem_solve_diagnostics, include/mmx.h):
               // estimated relative error of theta, smallest pivot ratio, largest refinement ratio, |theta|
  float precisionBound; // status |= MMX_SOLVE_PRECISION_SUSPECT when the estimate exceeds it (<= 0: never)
};

// mmx_gn_options::precision == MMX_PRECISION_AUTO: the double solve runs on the elements map[0 .. *count - 1] only (workgroup
// i takes element map[i]; workgroups beyond *count leave at once), reads their initial parameters from the float copy taken
// before the single-precision solve and writes the result back rounded to float.
struct F64Select {
  const int32_t* map; // [B] or null: every element, theta in double in place
  const int32_t* count; // [1]
  const float* thetaInit; // [B][P]
  float* thetaOut; // [B][P]
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: the largest value requested so far is
// remembered per device (one process may drive several GPUs from several threads: BatchedMultiGpuSolver), so that the
// attribute is set once per (kernel, device, size class) and not once per launch.
struct LdsLimitCache {
  std::atomic<size_t> bytes[64];
  LdsLimitCache() {
    for (auto& b : bytes) {
      b.store(64 * 1024); // default dynamic-LDS limit
    }
  }
  hipError_t ensure(const void* kernel, size_t lds) {
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) {
      return rc;
    }
    std::atomic<size_t>& slot = bytes[dev & 63];
    if (lds > slot.load(std::memory_order_relaxed)) {
      rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
      if (rc != hipSuccess) {
        return rc;
      }
      slot.store(lds, std::memory_order_relaxed);
    }
    return hipSuccess;
  }
};

// TrustRegionQRT on the wide route (several kernels per trust step, driven from the host): per-instance state.
// phase: 0 the damping changed (or the trust step begins): factor + solve, 1 a step is on the table and the radius
// changed: decide again, 2 the step is the one to try, 3 the iteration is over for this instance.
struct TrustStateDev {
  float* lambda; // [B] TrustRegionQRT's lambda (starts at 1e-10 every iteration, only grows within it)
  float* radius; // [B] curTrustRegionRadius_ (lives across iterations)
  int32_t* phase; // [B]
  int32_t* newton; // [B] Newton updates of lambda taken in the current trust step (<= 3)
  int32_t* step; // [B] trust steps tried in the current iteration (<= 10)
  int32_t* mask; // [B] 1 = the factor kernels skip this instance (phase != 0, or converged)
  int32_t* active; // [1] instances whose iteration is not over (phase != 3), counted by the trial kernel
};

struct StepParams {
  float lambda; // regularization added to the diagonal of the compacted system
  float threshold; // SolverOptions::threshold
  int32_t iteration; // iteration_ (0-based)
  int32_t minIterations;
  int32_t maxIterations;
  int32_t refine; // refinement steps through J allowed per iteration (0..3)
  // ---- line search / damping schedule of the explicit-Jacobian path: the Cholesky kernel leaves
  // the step in `delta` and stepUpdateKernel applies it (all null / 0: theta -= delta in place)
  float* delta; // [B][n] step of this iteration
  int32_t* stepIter; // [B] iteration + 1 when `delta` holds a step, -(iteration + 1) when H was not positive definite
  float* lambdaPer; // [B] per-instance damping (LM schedule, trust region) or null: `lambda` for everyone
  double* stepHistory; // SolveStateDev::stepHistory for stepUpdateKernel (which applies the schedule on the wide / explicit routes)
  float* diagErr0; // [B] with diagAcc: the first iteration's error (the estimate weights an iteration with sqrt(e_it / e_0))
  float* diagAcc; // [B][4] or null: the wide route's precision estimate in the making -- [0] largest (w of an iteration) x sqrt(e_it / e_0), [3] w of the iteration under way, [1] largest kPivotFloor (H_jj + mu) / d_jj
                  // of the solve (float bits, by atomicMax), [2] largest squared refinement ratio (fusedSolveKernel keeps
                  // the same two in LDS); solveFinalizeKernel turns them into SolveStateDev::diag and the status bit
  TrustStateDev tr; // MMX_STEP_TRUST_REGION on the wide route (all null otherwise)
  int32_t doLineSearch; // GaussNewtonSolverOptions::doLineSearch
  int32_t stepRule; // MMX_STEP_*
  float lmLambdaMin, lmLambdaMax, lmUp, lmDown;
  long long* clk; // profiling aid (MMX_PHASE_CLOCKS): per-phase cycles of block 0, or null
  // tile structure of the tile-major factor (mmx::TileMasks): [0..31] rowMask, [32..63] colMask, [64..95] colBase (tiles
  // in the columns before block column k: the column-compact slot numbering of the LDS-resident factor), [96 + slot]:
  // the tile in that slot, I | k << 8.  The tiled
  // factor, its sweeps and treeNormalEquationsKernel touch only the tiles named here; the others are never written nor read.
  const uint32_t* tileMasks;
  int32_t numTiles; // structurally non-zero tiles
};

// device view of mmx::FusedTables (mmx_host_tables.hpp)
struct FusedDev {
  int32_t U, Kp, n, nsrc, nnz; // units, position constraints, solved parameters, column sources, CSR entries
  int32_t slotBase; // first extra source slot = number of primary slots (16 x the fused instantiation's blocks when there is one, else n rounded up to 16)
  const int32_t* subSize; // [J] by DFS position
  const int32_t* dfsJoint; // [J] joint at DFS position k
  const int32_t* loadedPos; // [numLoaded] DFS positions of the joints that carry constraint units, ascending
  int32_t numLoaded;
  const int32_t* unitJoint; // [U]
  const int32_t* posUnitStart; // [J+1] by DFS position
  const int32_t* posUnits; // [U]
  const int32_t* solveList; // [n]
  const int32_t* srcStart; // [n+1]
  const ColumnSourceDev* srcs; // [nsrc]
  // Structural term records of H = J^T J (integer bookkeeping done on the host, mmx_capi.hip):
  // thread t of the workgroup processes records gTerms[k * 256 + t], k = 0..termRounds-1; the
  // non-zero terms of one H entry are consecutive records of ONE thread, balanced over threads.
  const uint4* gTerms; // x: deep | anc << 12 | first << 24 | last << 25 | valid << 26 ; y: LDS address
                       // of the entry inside the tile region ; z: weight product (float bits)
  int32_t termRounds;
  const uint4* gTerms16; // the same records dealt to 1024 threads (the sixteen-wave treeNormalEquationsKernel): [k * 1024 + t]
  int32_t termRounds16;
  const int32_t* comb; // [numComb][3] (tile-region offset, first partial cell, cell count) of split entries
  int32_t numComb;
  int32_t numCells; // partial cells of the split entries (the kernels park them in a scratch array)
  // parameter-space rows (limits on model parameters, model-parameter targets): which limits touch
  // a solve column, and which share an off-diagonal H entry
  int32_t numLimits;
  const int32_t* limStart; // [n+1]
  const int32_t* limOf; // limit indices per solve column
  int32_t numPairDests;
  const int32_t* pairDest; // [numPairDests] float offset of the H entry inside the tile region
  const int32_t* pairCols; // [numPairDests][2] the two solve columns of that entry
  const int32_t* pairStart; // [numPairDests+1]
  const int32_t* pairLim; // limit indices per destination
  // rows of the further joint error functions + ellipsoid limits handled inside the fused solve (kGen instantiations):
  // GT = constraints (ProblemDev::G + NE), genRows = their Jacobian rows (rowsJoint - 3 U); 0 / 0: none
  int32_t GT, genRows;
  // the structurally non-zero tiles of the factor (and of H) in elimination order: I | J << 8 (mmx::TileMasks::tiles)
  const int32_t* tileList;
  int32_t numTiles;
};

struct FusedParams {
  float lambda;
  float threshold;
  int32_t minIterations;
  int32_t maxIterations;
  int32_t refine;
  int32_t doLineSearch; // GaussNewtonSolverT::updateParameters backtracking (gauss_newton_solver.cpp:283-313)
  int32_t stepRule; // MMX_STEP_*
  float lmLambdaMin, lmLambdaMax, lmUp, lmDown;
  float trustRadius; // TrustRegionQROptions::trustRegionRadius_
  // the mixed-precision instantiation (MMX_PRECISION_MIXED): stop the conjugate gradients when the predicted next correction
  // falls below mixTol |step|; at most mixMaxCg operator applications per iteration
  float mixTol;
  int32_t mixMaxCg;
  int32_t autoAbort; // single-precision instantiations: leave an element as soon as a factorisation marks it MMX_SOLVE_PRECISION_SUSPECT (MMX_PRECISION_AUTO's first pass)
};

// mmx_gn_options::precision == MMX_PRECISION_AUTO with the mixed-precision instantiation as its second pass: workgroup i takes
// element map[i] (the ones beyond *count leave at once) and starts from the float copy of the initial parameters taken before
// the single-precision pass.  All null: every element, in place.
struct MixSelect {
  const int32_t* map; // [B] or null
  const int32_t* count; // [1]
  const float* thetaInit; // [B][P] or null: theta itself
};

// cellsBehindRho: the instantiations that carry parameter-space rows park their diagonal in rho while the term records run,
// so the split entries' partial cells lie behind rho / invDiag there (numCells floats more)
size_t fusedLdsBytes(int NB, int J, int P, int U, int nsrc, int n, int numCells, bool cellsBehindRho, int GT = 0, int genRows = 0, bool separateUy = false, size_t csrFloats = 0, bool mix = false);
size_t fusedCsrFloats(int J, int nnz); // LDS copy of the transform's CSR (the instantiations below four workgroups per CU)
// H = J^T J, g = J^T r from the tree moments for the explicit-Jacobian solver (mmx_fused.hip, treeNormalEquationsKernel)
size_t treeNormalEquationsLdsBytes(int J, int P, int U, int nsrc, int n, int GT = 0, int genRows = 0);
size_t treeRefineLdsBytes(int J, int P, int U, int n, int genRows);
size_t treeGenStateFloats(int n, int genRows); // per instance: J_g and its residual rows, handed from the tree normal equations to the tree refinement
hipError_t launchTreeNormalEquations(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    const float* theta,
    float* jtj,
    float* jtr,
    const int32_t* done,
    double* errOut, // [B] error at theta, or null
    float* state, // [B][treeStateFloats(J, U)] hand-over to launchTreeRefine, or null
    long long* clk, // profiling aid: eight per-phase cycle counters of block 0, or null
    float* genState, // [B][treeGenStateFloats] (problems with further joint error functions / ellipsoid limits), or null
    bool tileMajor, // jtj as [tile (I,J) at I(I+1)/2 + J][col][row] (launchCholeskyFactorTiled reads that) instead of [n][n]
    hipStream_t stream);
size_t treeStateFloats(int J, int U);
// rho = J^T (r - J d) - lambda d through the tree, for the instances with refState[b] == 0 (dvec / rhoVec: [B][NP])
hipError_t launchTreeRefine(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    const float* theta,
    const float* state,
    const float* genState,
    const float* dvec,
    float* rhoVec,
    const int32_t* refState,
    float lambda,
    const float* lambdaPer,
    hipStream_t stream);
int fusedBlocksFor(int n); // number of 16-wide blocks the fused kernel is instantiated for, or -1
hipError_t launchFusedSolve(
    const RigDev& rig,
    const ProblemDev& pb,
    const FusedDev& fd,
    float* theta,
    const SolveStateDev& st,
    const FusedParams& fp,
    float* dbgH,
    float* dbgG,
    long long* dbgClk,
    void* argsBuf, // fusedArgsBytes() of device memory owned by the problem, or null: where the descriptors are stashed for the lazy-argument instantiations (mmx_fused.hip, kArgLazy); written in stream order before the solve
    hipStream_t stream);
size_t fusedArgsBytes();
// the mixed-precision instantiation of the one-launch solve (mmx_gn_options::precision == MMX_PRECISION_MIXED): usable for
// problems without parameter-space rows / further joint error functions / the trust region, up to 128 solved parameters
bool fusedMixedUsable(int J, int P, int U, int nsrc, int n, int numCells);
hipError_t launchFusedMixed(
    const RigDev& rig, const ProblemDev& pb, const FusedDev& fd, float* theta, const SolveStateDev& st, const FusedParams& fp, const MixSelect& sel, int blocks, long long* dbgClk, hipStream_t stream);

// double-precision solve (mmx_f64.hip)
size_t solveF64LdsBytes(int J, int P, int U, int n, int G = 0, int genRows = 0);
bool solveF64IsResident(int J, int P, int U, int n, int G = 0, int genRows = 0); // the system stays in LDS: no J / H scratch is read or written
// rows of J the resident form assembles per chunk (a multiple of twelve; 0: the scratch form runs)
int solveF64ResidentChunkRows(int J, int P, int U, int n, int G = 0, int genRows = 0);
// The resident form's assembly list (batch-shared, iteration-invariant; built on the host, mmx_capi.hip): per chunk of
// units the entries (column, unit) of J that have an applicable source at all -- one in six on the 72-joint rig -- each
// with the indices of those sources in the kernel's packed source table.  groups[g] = {column | unit-in-chunk << 12 |
// count << 18, count == 1 ? the source : offset into extra}; chunkStart[chunk] .. chunkStart[chunk + 1] the chunk's groups.
// Null pointers: the kernel tests every (column, unit) pair itself.
struct F64AssemblyList {
  const uint2* groups;
  const int32_t* extra;
  const int32_t* chunkStart; // [chunks + 1], then [chunks] the chunk's BLOCK MASK: bit I set when some entry of the chunk lies in
                             // the 16-column block I -- a tile of H whose row or column block is empty in a chunk gets nothing from it
  int32_t unitsPerChunk; // the chunking the list was built for (must equal the launch's)
};
hipError_t launchSolveF64(
    const RigDev& rig,
    const ProblemDev& pb,
    const int32_t* solveList,
    int n,
    double* theta,
    const SolveStateDev& st,
    const FusedParams& fp,
    double* Jg,
    double* Hg,
    double* Hg2, // second [B][n][n] scratch, MMX_STEP_TRUST_REGION only (else null)
    hipStream_t stream,
    const F64AssemblyList& list = F64AssemblyList{nullptr, nullptr, nullptr, 0},
    const F64Select& select = F64Select{nullptr, nullptr, nullptr, nullptr});
// elements whose status has a bit of `mask` set, in index order: map[0 .. *count - 1]
hipError_t launchSelectSuspect(const int32_t* status, int B, int32_t mask, int32_t* map, int32_t* count, hipStream_t stream, int32_t require = 0);

size_t fkJacobianLdsBytes(int J, int P, int U);
// store-only counterpart of the J-assembly kernel (profiling aid; see storePatternKernel)
// wavefronts per instance launchFkJacobian takes for this problem (J-assembly when withJacobian, else FK / r / error only)
int fkJacobianWavesPerInstance(const RigDev& rig, const ProblemDev& pb, bool withJacobian);
// (waves: fkJacobianWavesPerInstance of the problem the pattern stands for -- the probe keeps the graded kernel's launch shape)
hipError_t launchStorePattern(float* jac, int B, int M, int P, int waves, hipStream_t stream, hipEvent_t startEvent, hipEvent_t stopEvent);
// column-major [B][P][M] -> row-major [B][M][P] (MMX_LAYOUT_ROW_MAJOR)
hipError_t launchTransposeJacobian(const float* colMajor, float* rowMajor, int B, int M, int P, hipStream_t stream);
// Most solved parameters of a problem: up to 512 on every route (tile masks of 32 blocks); 513 ... kMaxSolved on the
// explicit-Jacobian route only (dense J, VALU normal equations, unmasked left-looking factor in HBM, refinement through J)
// -- the reference's kMaxModelParams (momentum/math/types.h:426-429).  What bounds this route is the factor's panel
// (16 columns x n rows) + the refinement's chunk of J in one workgroup's LDS: 154 KB at 2048.
constexpr int kMaxSolved = 2048;
size_t normalEquationsLdsBytes(int n);
size_t choleskyStepLdsBytes(int n, int M);

hipError_t launchFkJacobian(
    const RigDev& rig,
    const ProblemDev& pb,
    const float* theta,
    float* jac,
    float* res,
    double* err,
    float* state,
    const int32_t* done,
    hipStream_t stream,
    hipEvent_t startEvent = nullptr, // attached to the J-assembly dispatch itself (hipExtLaunchKernelGGL)
    hipEvent_t stopEvent = nullptr,
    // the pointer-jumping rounds in double like the solve kernels' (mmx_device.hpp fkJumpRoundsD): the explicit-Jacobian SOLVE
    // route and mmx_eval_skeleton_state; mmx_eval_jacobian (the graded kernel) keeps the single-precision rounds
    bool accurateFk = false);

hipError_t launchNormalEquations(
    const ProblemDev& pb,
    int P,
    const float* jac,
    const float* res,
    float* jtj,
    float* jtr,
    const int32_t* done,
    bool lowerOnly, // the caller only reads the lower triangle of jtj (wide systems skip the mirror stores)
    hipStream_t stream);

hipError_t launchCholeskyStep(
    const ProblemDev& pb,
    int P,
    const float* jac,
    const float* res,
    const float* jtj,
    const float* jtr,
    const double* errIter,
    float* theta,
    const SolveStateDev& st,
    const StepParams& sp,
    float* factor, // wide systems: [B][choleskyFactorFloats(n)] scratch for the tile-major factor (null: factor H in place)
    hipStream_t stream);
// the same step when the refinement goes through the tree (launchTreeRefine): factor + first solve, then one call per
// refinement round.  dvec / rhoVec: [B][NP] (NP = n rounded up to 16), refState: [B].
hipError_t launchCholeskyFactorTiled(
    const ProblemDev& pb,
    int P,
    const float* jtj, // tile-major (launchTreeNormalEquations with tileMajor)
    const float* jtr,
    float* factor,
    float* dvec,
    int32_t* refState,
    const double* errIter,
    float* theta,
    const SolveStateDev& st,
    const StepParams& sp,
    hipStream_t stream);
hipError_t launchCholeskyFinishTiled(
    const ProblemDev& pb,
    int P,
    const float* factor,
    float* dvec,
    const float* rhoVec,
    int32_t* refState,
    const double* errIter,
    float* theta,
    const SolveStateDev& st,
    const StepParams& sp,
    int round,
    hipStream_t stream);
inline size_t choleskyFactorFloats(int n) {
  const size_t nb = (size_t(n) + 15) / 16;
  return nb * (nb + 1) / 2 * 256;
}

// applies the deferred step of the iteration: Armijo backtracking (GaussNewtonSolverT::updateParameters,
// gauss_newton_solver.cpp:283-313) or the LM gain-ratio schedule (see fusedSolveKernel phase K)
hipError_t launchStepUpdate(
    const RigDev& rig,
    const ProblemDev& pb,
    float* theta,
    const float* jtr,
    const double* errIter,
    const StepParams& sp,
    hipStream_t stream);

// bytes (a multiple of four) of zeros at p, by a KERNEL: the solve paths use no memset nodes.  Inside a captured graph (torch's
// CUDAGraph around mmx_solve) a 4-byte hipMemsetAsync between dependent kernels -- the element count of MMX_PRECISION_AUTO's
// compaction -- brought the replay down on ROCm 7.2 (abort in the runtime once the second and the third pass both had
// elements; tests/test_gpu_graph.py); a kernel node in its place replays bit for bit.
hipError_t zeroAsync(void* p, size_t bytes, hipStream_t stream);
hipError_t launchSolveInit(const SolveStateDev& st, int B, float* lambdaPer, float lambda0, hipStream_t stream, float* diagAcc = nullptr);
// TrustRegionQRT on the wide route: state of a solve / of an iteration, the decision after a linear solve (is the step
// within the radius, or does lambda take a Newton update first), the bookkeeping at the end of an iteration
hipError_t launchTrustInit(const TrustStateDev& tr, int B, float radius0, hipStream_t stream);
hipError_t launchTrustBegin(const TrustStateDev& tr, const SolveStateDev& st, float* lambdaPer, int B, hipStream_t stream);
hipError_t launchTrustDecide(const ProblemDev& pb, const float* factor, const float* jtr, const double* errIter, const SolveStateDev& st, const StepParams& sp, hipStream_t stream);
hipError_t launchTrustEnd(const SolveStateDev& st, const StepParams& sp, const double* errIter, int B, hipStream_t stream);
// zeroes the rows of paramHistory [B][maxIterations][P] from iterations[b] on (the reference leaves them at setZero())
hipError_t launchParamHistoryFinalize(float* paramHistory, const int32_t* iterations, int B, int maxIterations, int P, hipStream_t stream);
hipError_t launchSolveFinalize(float* theta, const float* thetaInit, int P, const SolveStateDev& st, int B, hipStream_t stream, const float* diagAcc = nullptr);

} // namespace mmx
