// mmx_host_tables.hpp -- host-side integer bookkeeping of the batched-IK path (no HIP here, so it
// is unit-testable bit-exactly without a GPU).  Everything in this file is index arithmetic:
// tree levels and DFS intervals of the Skeleton, the enabled-parameter list of
// GaussNewtonSolverT::updateEnabledParameters (momentum/solver/gauss_newton_solver.cpp:57-66),
// ParameterTransformT::computeActiveJointParams (momentum/character/parameter_transform.cpp:97-107)
// and a column-wise (CSC) view of the enabled part of the parameter transform, which is what lets
// the kernels GATHER a Jacobian column instead of scattering like the reference's ancestor walk
// (momentum/character_solver/joint_error_function-inl.h:228-294).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mmx.h"

namespace mmx {

// One source term of a Jacobian column p: joint-parameter row (joint, dof) of the parameter
// transform with weight w = transform(7*joint+dof, p).  tin/tout = DFS interval of `joint`:
// the term contributes to a constraint on joint j iff tin <= tin[j] < tout (joint is j or one of
// its ancestors -- exactly the set the reference's while(jntIndex != kInvalidIndex) loop visits).
struct ColumnSource {
  int32_t joint;
  int32_t dof; // 0..2 translation, 3..5 rotation, 6 scale
  int32_t tin;
  int32_t tout;
  int32_t parent; // parent joint of `joint` (-1 for a root): translationAxis = parent.toLinear()
  float weight;
};

// One record of the J-assembly kernel's column program: a Jacobian column with exactly ONE
// source that is a ROTATION dof -- the overwhelmingly common case -- listed grouped by joint so
// that the kernel computes the ancestor test and v - t_joint once per joint.
// 32 bytes = one s_load_dwordx8.
struct JacRec {
  int32_t joint, dof, col, tin, tout, parent;
  float weight;
  int32_t valid; // unused (padding records are copies of the last real record)
};

struct HostTables {
  int32_t J = 0, P = 0;
  // --- skeleton topology
  std::vector<int32_t> level; // [J] depth, root = 0
  std::vector<int32_t> levelOrder; // [J] joints sorted by (level, index)
  std::vector<int32_t> levelStart; // [numLevels+1] offsets into levelOrder
  std::vector<int32_t> tin, tout; // [J] DFS pre-order interval
  // --- enabled-parameter dependent
  std::vector<uint8_t> enabled; // [P]
  std::vector<uint8_t> activeJointParams; // [7J]
  std::vector<int32_t> enabledList; // [n] ascending
  std::vector<int32_t> fullToSubset; // [P] index into enabledList or -1
  // the enabled parameters in ELIMINATION order (a permutation of enabledList): the order in which the solvers number
  // the columns of their normal equations.  Two columns of J overlap only when a joint of the one is an ancestor of a
  // joint of the other, so with every subtree's parameters ahead of those of the joints above it (a post-order of the
  // skeleton, small subtrees first) the Cholesky factor of J^T J fills in next to nothing outside that ancestor pattern
  // -- the blocked solvers skip the 16 x 16 tiles that stay zero (eliminationTileMasks).  The reference's dense QR has
  // no such order; the step it computes does not depend on one.
  std::vector<int32_t> eliminationList;
  std::vector<int32_t> colStart; // [P+1] offsets into colSources (disabled columns are empty)
  std::vector<ColumnSource> colSources;
  int32_t maxColSources = 0;
  // column program of the J-assembly kernel
  std::vector<JacRec> jacRecs; // single-source rotation columns, grouped by joint, padded to a multiple of 4
  std::vector<int32_t> multiCols; // all other non-empty columns (generic gather path)
  std::vector<int32_t> zeroCols; // columns without sources (disabled parameters): written as zeros
};

// Tables of the fused solve kernel, for one (rig, constraint topology, enabled set):
//  - joints in DFS pre-order, so that the subtree of a joint is a contiguous index range
//  - the constraint vectors ("units", 3 Jacobian rows each) attached to each joint
//  - the SOLVE list: enabled parameters whose Jacobian column is not structurally zero, i.e. that
//    drive at least one joint with a constrained joint in its subtree.  A structurally zero column
//    has H row/col = 0 and g = 0, so the reference's step for it is exactly 0 / (0 + lambda) = 0
//    (gauss_newton_solver.cpp:248-257); dropping it from the dense system changes nothing.
struct FusedTables {
  int32_t U = 0; // units = Kp + 3 Ko
  std::vector<int32_t> dfsJoint; // [J] joint at DFS position k
  std::vector<int32_t> subSize; // [J] subtree size of the joint at DFS position k
  std::vector<int32_t> unitJoint; // [U]
  std::vector<int32_t> posUnitStart; // [J+1] CSR over DFS positions -> units attached to that joint
  std::vector<int32_t> posUnits; // [U]
  std::vector<uint8_t> structNonZero; // [P] the parameter's column of the joint-constraint rows can be non-zero
  std::vector<int32_t> solveList; // [n] parameter index of compacted column s (in HostTables::eliminationList order)
  std::vector<int32_t> srcStart; // [n+1] offsets into srcs per compacted column
  std::vector<ColumnSource> srcs;
  int32_t maxDepth = 0;
};

int32_t buildFusedTables(
    const mmx_rig_desc* d,
    const HostTables& t,
    int32_t Kp,
    const int32_t* posParent,
    int32_t Ko,
    const int32_t* oriParent,
    const uint8_t* forceSolve, // [P] or null: enabled parameters to keep in the solve list although
                               // no joint constraint reaches them (limit / model-parameter rows)
    const std::vector<int32_t>* unionPos, // or null: with per-instance constraint parents, every joint that
    const std::vector<int32_t>* unionOri, // carries a position / an orientation constraint in SOME element
    FusedTables& out,
    std::string& err);

// Tile structure of the Cholesky factor of H = J^T J + (parameter-space rows) for a solve list in elimination order:
// `related(row, col)` (row > col) says whether entry (row, col) of H can be non-zero.  Symbolic factorisation on the
// 16 x 16 tile grid (tile (I, J) of L is non-zero when it is in H or when two tiles (I, k), (J, k), k < J, are).
// rowMask[I]: bit J set = tile (I, J <= I) of L is structurally non-zero; colMask[k]: bit I set = tile (I >= k, k) is.
// n <= 512 (32 blocks).  Integer bookkeeping, unit-tested on the CPU.
struct TileMasks {
  int32_t NB = 0;
  uint32_t rowMask[32] = {};
  uint32_t colMask[32] = {};
  std::vector<int32_t> tiles; // the non-zero tiles, I | J << 8, in tile-index order (I (I + 1) / 2 + J ascending)
  int64_t products = 0; // tile products L(I,j) L(k,j)^T of the masked factorisation (dense: NB (NB^2 - 1) / 6)
  // Level schedule of the factorisation (the resident factor kernel): block column k can be factored once every column j
  // with a tile (k, j) is -- level[k] = 1 + max level[j] --, so the columns of one level are independent (the subtrees of
  // the skeleton's elimination tree: finger chains next to the spine's).  steps: [numSteps, then 4 words per step], a word
  // = k | firstWave << 8 | numWaves << 12 (a column's panel of nt tiles takes ceil((16 nt - 16) / 48) of the workgroup's
  // four waves, at least one; 0xf = the whole workgroup, for panels beyond 208 rows) or -1; the columns of a step are
  // factored side by side.
  std::vector<int32_t> levelSteps;
};
TileMasks eliminationTileMasks(int32_t n, const std::vector<uint8_t>& related /* [n][n], lower triangle used */, bool dense);

// mmx_solve_f64's assembly list (mmx::F64AssemblyList, mmx_kernels.hpp) for chunks of `unitsPerChunk` units: every entry
// (solved column, unit) of J with an applicable source -- the source's joint an ancestor-or-self of the unit's joint (DFS
// interval) and, for translation / scale dofs, a point unit (joint_error_function-inl.h:248-291) -- with the indices of
// those sources in the kernel's packed table (columns in solve-list order, a column's sources in colSources order).
// groups: two words per entry, column | unit-in-chunk << 12 | count << 18 and (count == 1 ? the source : offset into
// extra); chunkStart: [chunks + 1] first group of a chunk, then [chunks] the chunk's mask of 16-column blocks with an entry.
// Units: Kp position constraints (points), then three per orientation constraint.  False when a count does not fit.
struct F64AssemblyListHost {
  std::vector<uint32_t> groups;
  std::vector<int32_t> extra, chunkStart;
};
bool buildF64AssemblyListHost(
    const HostTables& t, const std::vector<int32_t>& solveList, const int32_t* posParent, int32_t Kp, const int32_t* oriParent, int32_t Ko,
    int32_t unitsPerChunk, F64AssemblyListHost& out);

// Validates the descriptor the way the reference's constructors / MT_CHECKs do
// (skeleton.cpp:16-22 parent-before-child; parameter_transform.cpp:112-121 sizes).
// Returns MMX_OK or an error code with a message in `err`.
int32_t validateRigDesc(const mmx_rig_desc* d, std::string& err);

// Builds all tables.  `enabled` may be null (= all parameters enabled).
int32_t buildHostTables(const mmx_rig_desc* d, const uint8_t* enabled, HostTables& out, std::string& err);

} // namespace mmx
