// mmx_host_tables.cpp -- see mmx_host_tables.hpp.  Pure host C++17, integer bookkeeping only.
#include "mmx_host_tables.hpp"

#include <algorithm>

namespace mmx {

int32_t validateRigDesc(const mmx_rig_desc* d, std::string& err) {
  if (d == nullptr) {
    err = "rig descriptor is null";
    return MMX_ERR_INVALID_ARGUMENT;
  }
  const int32_t J = d->num_joints, P = d->num_params;
  if (J <= 0 || P <= 0) {
    err = "rig needs at least one joint and one model parameter";
    return MMX_ERR_INVALID_ARGUMENT;
  }
  if (J > 32767) {
    err = "num_joints exceeds 32767 (joint and level indices are packed into 16 bits on the device)";
    return MMX_ERR_INVALID_ARGUMENT;
  }
  if (P > MMX_MAX_MODEL_PARAMS) {
    err = "num_params exceeds kMaxModelParams (2048)";
    return MMX_ERR_INVALID_ARGUMENT;
  }
  if (!d->parent || !d->pre_rotation || !d->translation_offset || !d->pt_outer || !d->pt_inner || !d->pt_value) {
    err = "rig descriptor has a null array";
    return MMX_ERR_INVALID_ARGUMENT;
  }
  for (int32_t j = 0; j < J; ++j) {
    // Skeleton invariant (momentum/character/skeleton.cpp:16-22): parents precede children
    if (d->parent[j] != MMX_INVALID_PARENT && (d->parent[j] < 0 || d->parent[j] >= j)) {
      err = "joint " + std::to_string(j) + " has parent " + std::to_string(d->parent[j]) +
          ": joints must be listed parent-before-child";
      return MMX_ERR_INVALID_ARGUMENT;
    }
  }
  const int32_t R = MMX_PARAMS_PER_JOINT * J;
  if (d->pt_outer[0] != 0) {
    err = "pt_outer[0] must be 0";
    return MMX_ERR_SIZE_MISMATCH;
  }
  for (int32_t r = 0; r < R; ++r) {
    if (d->pt_outer[r + 1] < d->pt_outer[r]) {
      err = "pt_outer must be non-decreasing";
      return MMX_ERR_SIZE_MISMATCH;
    }
  }
  const int32_t nnz = d->pt_outer[R];
  for (int32_t k = 0; k < nnz; ++k) {
    if (d->pt_inner[k] < 0 || d->pt_inner[k] >= P) {
      err = "parameter-transform column index out of range (transform.cols() != num_params)";
      return MMX_ERR_SIZE_MISMATCH;
    }
  }
  return MMX_OK;
}

int32_t buildHostTables(const mmx_rig_desc* d, const uint8_t* enabled, HostTables& t, std::string& err) {
  const int32_t rc = validateRigDesc(d, err);
  if (rc != MMX_OK) {
    return rc;
  }
  const int32_t J = d->num_joints, P = d->num_params, R = MMX_PARAMS_PER_JOINT * J;
  t.J = J;
  t.P = P;

  // ---- levels (parents precede children, so one forward sweep suffices)
  t.level.assign(J, 0);
  int32_t maxLevel = 0;
  for (int32_t j = 0; j < J; ++j) {
    t.level[j] = d->parent[j] < 0 ? 0 : t.level[d->parent[j]] + 1;
    maxLevel = std::max(maxLevel, t.level[j]);
  }
  t.levelStart.assign(maxLevel + 2, 0);
  for (int32_t j = 0; j < J; ++j) {
    t.levelStart[t.level[j] + 1]++;
  }
  for (int32_t l = 0; l <= maxLevel; ++l) {
    t.levelStart[l + 1] += t.levelStart[l];
  }
  t.levelOrder.assign(J, 0);
  {
    std::vector<int32_t> cursor(t.levelStart.begin(), t.levelStart.end() - 1);
    for (int32_t j = 0; j < J; ++j) { // ascending j keeps (level, index) order
      t.levelOrder[cursor[t.level[j]]++] = j;
    }
  }

  // ---- DFS pre-order intervals.  Subtree sizes by a backward sweep, then tin by a forward sweep
  // that hands each child the next free slot of its parent (children in index order).
  std::vector<int32_t> size(J, 1), nextSlot(J, 0);
  for (int32_t j = J - 1; j >= 0; --j) {
    if (d->parent[j] >= 0) {
      size[d->parent[j]] += size[j];
    }
  }
  t.tin.assign(J, 0);
  t.tout.assign(J, 0);
  int32_t rootCursor = 0;
  for (int32_t j = 0; j < J; ++j) {
    const int32_t p = d->parent[j];
    if (p < 0) {
      t.tin[j] = rootCursor;
      rootCursor += size[j];
    } else {
      t.tin[j] = nextSlot[p];
    }
    nextSlot[j] = t.tin[j] + 1;
    if (p >= 0) {
      nextSlot[p] = t.tin[j] + size[j];
    }
    t.tout[j] = t.tin[j] + size[j];
  }

  // ---- enabled set
  t.enabled.assign(P, 1);
  if (enabled != nullptr) {
    for (int32_t p = 0; p < P; ++p) {
      t.enabled[p] = enabled[p] ? 1 : 0;
    }
  }
  // GaussNewtonSolverT::updateEnabledParameters (gauss_newton_solver.cpp:57-66)
  t.enabledList.clear();
  t.fullToSubset.assign(P, -1);
  for (int32_t p = 0; p < P; ++p) {
    if (t.enabled[p]) {
      t.fullToSubset[p] = int32_t(t.enabledList.size());
      t.enabledList.push_back(p);
    }
  }
  // ---- elimination order: joints in post-order (children before their parent, smaller subtrees first, ties by index);
  // a parameter sits where the LAST of the joints it drives does (a shared parameter joins the highest of its joints)
  {
    std::vector<std::vector<int32_t>> children(J);
    std::vector<int32_t> roots;
    for (int32_t j = 0; j < J; ++j) {
      (d->parent[j] >= 0 ? children[d->parent[j]] : roots).push_back(j);
    }
    auto bySize = [&](int32_t a, int32_t b) { return size[a] != size[b] ? size[a] < size[b] : a < b; };
    std::stable_sort(roots.begin(), roots.end(), bySize);
    for (auto& c : children) {
      std::stable_sort(c.begin(), c.end(), bySize);
    }
    std::vector<int32_t> postPos(J, 0), stack, cursor(J, 0);
    int32_t next = 0;
    for (int32_t r : roots) {
      stack.push_back(r);
      while (!stack.empty()) {
        const int32_t j = stack.back();
        if (cursor[j] < int32_t(children[j].size())) {
          stack.push_back(children[j][cursor[j]++]);
        } else {
          postPos[j] = next++;
          stack.pop_back();
        }
      }
    }
    std::vector<int32_t> key(P, J); // parameters that drive no joint: last, by index
    for (int32_t r = 0; r < R; ++r) {
      for (int32_t k = d->pt_outer[r]; k < d->pt_outer[r + 1]; ++k) {
        const int32_t p = d->pt_inner[k], pos = postPos[r / MMX_PARAMS_PER_JOINT];
        key[p] = key[p] == J ? pos : std::max(key[p], pos);
      }
    }
    t.eliminationList = t.enabledList;
    std::stable_sort(t.eliminationList.begin(), t.eliminationList.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  }
  // ParameterTransformT::computeActiveJointParams (parameter_transform.cpp:97-107)
  t.activeJointParams.assign(R, 0);
  for (int32_t r = 0; r < R; ++r) {
    for (int32_t k = d->pt_outer[r]; k < d->pt_outer[r + 1]; ++k) {
      if (t.enabled[d->pt_inner[k]]) {
        t.activeJointParams[r] = 1;
      }
    }
  }

  // ---- CSC view of the enabled entries: column p lists its (joint, dof, weight) sources in
  // ascending joint-parameter row order.  A zero stored value is kept (the reference multiplies by
  // it too).  Rows of a disabled column are dropped here, which is the
  // enabledParameters_.test(inner[k]) test of joint_error_function-inl.h:257,273,286.
  std::vector<int32_t> count(P + 1, 0);
  for (int32_t r = 0; r < R; ++r) {
    for (int32_t k = d->pt_outer[r]; k < d->pt_outer[r + 1]; ++k) {
      if (t.enabled[d->pt_inner[k]]) {
        count[d->pt_inner[k] + 1]++;
      }
    }
  }
  t.colStart.assign(P + 1, 0);
  for (int32_t p = 0; p < P; ++p) {
    t.colStart[p + 1] = t.colStart[p] + count[p + 1];
  }
  t.colSources.assign(t.colStart[P], ColumnSource{});
  std::vector<int32_t> cursor(t.colStart.begin(), t.colStart.end() - 1);
  for (int32_t r = 0; r < R; ++r) {
    const int32_t joint = r / MMX_PARAMS_PER_JOINT, dof = r % MMX_PARAMS_PER_JOINT;
    for (int32_t k = d->pt_outer[r]; k < d->pt_outer[r + 1]; ++k) {
      const int32_t p = d->pt_inner[k];
      if (!t.enabled[p]) {
        continue;
      }
      ColumnSource& s = t.colSources[cursor[p]++];
      s.joint = joint;
      s.dof = dof;
      s.tin = t.tin[joint];
      s.tout = t.tout[joint];
      s.parent = d->parent[joint];
      s.weight = d->pt_value[k];
    }
  }
  t.maxColSources = 0;
  for (int32_t p = 0; p < P; ++p) {
    t.maxColSources = std::max(t.maxColSources, t.colStart[p + 1] - t.colStart[p]);
  }
  // column program of the J-assembly kernel: single-source ROTATION columns (the bulk of any rig)
  // go to the record list grouped by joint; every other non-empty column takes the generic path
  t.jacRecs.clear();
  t.multiCols.clear();
  t.zeroCols.clear();
  for (int32_t p = 0; p < P; ++p) {
    const int32_t cnt = t.colStart[p + 1] - t.colStart[p];
    if (cnt == 0) {
      t.zeroCols.push_back(p);
      continue;
    }
    const ColumnSource& s = t.colSources[t.colStart[p]];
    if (cnt == 1 && s.dof >= 3 && s.dof < 6) {
      t.jacRecs.push_back(JacRec{s.joint, s.dof, p, s.tin, s.tout, s.parent, s.weight, 1});
    } else {
      t.multiCols.push_back(p);
    }
  }
  std::stable_sort(t.jacRecs.begin(), t.jacRecs.end(), [](const JacRec& a, const JacRec& b) {
    return a.joint != b.joint ? a.joint < b.joint : a.dof < b.dof;
  });
  // pad to a multiple of 4 with copies of the last record (re-writing a column with the same
  // values is idempotent)
  while (!t.jacRecs.empty() && t.jacRecs.size() % 4 != 0) {
    t.jacRecs.push_back(t.jacRecs.back());
  }
  return MMX_OK;
}

int32_t buildFusedTables(
    const mmx_rig_desc* d,
    const HostTables& t,
    int32_t Kp,
    const int32_t* posParent,
    int32_t Ko,
    const int32_t* oriParent,
    const uint8_t* forceSolve,
    const std::vector<int32_t>* unionPos,
    const std::vector<int32_t>* unionOri,
    FusedTables& f,
    std::string& err) {
  const int32_t J = t.J, P = t.P;
  f.U = Kp + 3 * Ko;
  f.dfsJoint.assign(J, 0);
  f.subSize.assign(J, 1);
  for (int32_t j = 0; j < J; ++j) {
    f.dfsJoint[t.tin[j]] = j;
    f.subSize[t.tin[j]] = t.tout[j] - t.tin[j];
  }
  f.maxDepth = 0;
  for (int32_t j = 0; j < J; ++j) {
    f.maxDepth = std::max(f.maxDepth, t.level[j]);
  }
  f.unitJoint.assign(size_t(std::max(f.U, 1)), 0);
  for (int32_t c = 0; c < Kp; ++c) {
    f.unitJoint[c] = posParent[c];
  }
  for (int32_t c = 0; c < Ko; ++c) {
    for (int k = 0; k < 3; ++k) {
      f.unitJoint[Kp + 3 * c + k] = oriParent[c];
    }
  }
  // units per DFS position (ascending unit index within a joint: deterministic summation order)
  f.posUnitStart.assign(J + 1, 0);
  for (int32_t u = 0; u < f.U; ++u) {
    f.posUnitStart[t.tin[f.unitJoint[u]] + 1]++;
  }
  for (int32_t k = 0; k < J; ++k) {
    f.posUnitStart[k + 1] += f.posUnitStart[k];
  }
  f.posUnits.assign(size_t(std::max(f.U, 1)), 0);
  {
    std::vector<int32_t> cur(f.posUnitStart.begin(), f.posUnitStart.end() - 1);
    for (int32_t u = 0; u < f.U; ++u) {
      f.posUnits[cur[t.tin[f.unitJoint[u]]]++] = u;
    }
  }
  // a joint is "loaded" if some unit sits in its subtree (in some element of the batch, when the
  // constraint parents are per instance)
  std::vector<uint8_t> loaded(J, 0);
  auto markLoaded = [&](int32_t a) {
    while (a >= 0 && !loaded[a]) {
      loaded[a] = 1;
      a = d->parent[a];
    }
  };
  for (int32_t u = 0; u < f.U; ++u) {
    markLoaded(f.unitJoint[u]);
  }
  for (const std::vector<int32_t>* lst : {unionPos, unionOri}) {
    if (lst != nullptr) {
      for (int32_t j : *lst) {
        markLoaded(j);
      }
    }
  }
  // Orientation constraints only see rotation dofs, position constraints see all: a column is
  // structurally non-zero iff it has a source (a,dof) with a unit below a that the dof acts on.
  std::vector<uint8_t> hasPoint(J, 0);
  auto markPoint = [&](int32_t a) {
    while (a >= 0 && !hasPoint[a]) {
      hasPoint[a] = 1;
      a = d->parent[a];
    }
  };
  for (int32_t c = 0; c < Kp; ++c) {
    markPoint(posParent[c]);
  }
  if (unionPos != nullptr) {
    for (int32_t j : *unionPos) {
      markPoint(j);
    }
  }
  f.solveList.clear();
  f.srcStart.assign(1, 0);
  f.srcs.clear();
  f.structNonZero.assign(size_t(P), 0);
  for (int32_t p : t.eliminationList) {
    bool nz = false;
    for (int32_t e = t.colStart[p]; e < t.colStart[p + 1]; ++e) {
      const ColumnSource& s = t.colSources[e];
      const bool rot = s.dof >= 3 && s.dof < 6;
      if (rot ? loaded[s.joint] : hasPoint[s.joint]) {
        nz = true;
      }
    }
    f.structNonZero[size_t(p)] = nz ? 1 : 0;
    if (!nz && !(forceSolve != nullptr && forceSolve[p] != 0)) {
      continue;
    }
    f.solveList.push_back(p);
    // only the sources that can move a unit: a rotation dof with a unit in its joint's subtree, a
    // translation / scale dof with a POINT there (the others have zero moments below them, so every term
    // they take part in is exactly zero)
    for (int32_t e = t.colStart[p]; nz && e < t.colStart[p + 1]; ++e) {
      const ColumnSource& s = t.colSources[e];
      const bool rot = s.dof >= 3 && s.dof < 6;
      if (rot ? loaded[s.joint] : hasPoint[s.joint]) {
        f.srcs.push_back(s);
      }
    }
    f.srcStart.push_back(int32_t(f.srcs.size()));
  }
  (void)err;
  return MMX_OK;
}

TileMasks eliminationTileMasks(int32_t n, const std::vector<uint8_t>& related, bool dense) {
  TileMasks m;
  const int32_t NB = (n + 15) / 16;
  m.NB = NB;
  if (NB > 32) {
    return m;
  }
  for (int32_t I = 0; I < NB; ++I) {
    m.rowMask[I] = dense ? (I == 31 ? 0xffffffffu : ((1u << (I + 1)) - 1u)) : (1u << I);
  }
  if (!dense) {
    for (int32_t row = 0; row < n; ++row) {
      for (int32_t col = 0; col < row; ++col) {
        if (related[size_t(row) * size_t(n) + size_t(col)]) {
          m.rowMask[row >> 4] |= 1u << (col >> 4);
        }
      }
    }
    for (int32_t k = 0; k < NB; ++k) { // fill: the rows that hold a tile in column k become mutually coupled
      for (int32_t a = k + 1; a < NB; ++a) {
        if (!(m.rowMask[a] >> k & 1u)) {
          continue;
        }
        for (int32_t bb = k + 1; bb <= a; ++bb) {
          if (m.rowMask[bb] >> k & 1u) {
            m.rowMask[a] |= 1u << bb;
          }
        }
      }
    }
  }
  for (int32_t I = 0; I < NB; ++I) {
    for (int32_t k = 0; k <= I; ++k) {
      if (m.rowMask[I] >> k & 1u) {
        m.colMask[k] |= 1u << I;
        m.tiles.push_back(I | (k << 8));
        const uint32_t below = k == 0 ? 0u : ((1u << k) - 1u);
        m.products += __builtin_popcount(m.rowMask[I] & m.rowMask[k] & below);
      }
    }
  }
  // level schedule (TileMasks::levelSteps)
  {
    int32_t level[32] = {};
    int32_t maxLevel = -1;
    for (int32_t k = 0; k < NB; ++k) {
      int32_t lv = 0;
      for (int32_t j = 0; j < k; ++j) {
        if (m.rowMask[k] >> j & 1u) {
          lv = std::max(lv, level[j] + 1);
        }
      }
      level[k] = lv;
      maxLevel = std::max(maxLevel, lv);
    }
    std::vector<int32_t> steps;
    for (int32_t lv = 0; lv <= maxLevel; ++lv) {
      int32_t used = 0, inStep = 0;
      auto flush = [&]() {
        while (inStep > 0 && inStep < 4) {
          steps.push_back(-1);
          ++inStep;
        }
        used = 0, inStep = 0;
      };
      // the widest panels first: they take the low waves, the narrow ones fill up
      std::vector<int32_t> cols;
      for (int32_t k = 0; k < NB; ++k) {
        if (level[k] == lv) {
          cols.push_back(k);
        }
      }
      std::stable_sort(cols.begin(), cols.end(), [&](int32_t a, int32_t b2) { return __builtin_popcount(m.colMask[a]) > __builtin_popcount(m.colMask[b2]); });
      for (int32_t k : cols) {
        const int32_t nt = __builtin_popcount(m.colMask[k]);
        const int32_t rowsBelow = 16 * (nt - 1);
        if (rowsBelow > 192) { // the whole workgroup (tiledPanelFactor's tail substitution takes the rows beyond 208)
          flush();
          steps.push_back(k | (0 << 8) | (0xf << 12));
          inStep = 1;
          flush();
          continue;
        }
        const int32_t nw = std::max(1, (rowsBelow + 47) / 48);
        if (used + nw > 4 || inStep == 4) {
          flush();
        }
        steps.push_back(k | (used << 8) | (nw << 12));
        used += nw, ++inStep;
      }
      flush();
    }
    m.levelSteps.clear();
    m.levelSteps.push_back(int32_t(steps.size() / 4));
    m.levelSteps.insert(m.levelSteps.end(), steps.begin(), steps.end());
  }
  return m;
}

bool buildF64AssemblyListHost(
    const HostTables& t, const std::vector<int32_t>& solveList, const int32_t* posParent, int32_t Kp, const int32_t* oriParent, int32_t Ko,
    int32_t uc, F64AssemblyListHost& out) {
  const int32_t n = int32_t(solveList.size()), U = Kp + 3 * Ko;
  std::vector<int32_t> prefix(size_t(n) + 1, 0);
  for (int32_t c = 0; c < n; ++c) {
    const int32_t p = solveList[size_t(c)];
    prefix[size_t(c) + 1] = prefix[size_t(c)] + (t.colStart[size_t(p) + 1] - t.colStart[size_t(p)]);
  }
  std::vector<int32_t> unitTin(size_t(std::max(U, 1)));
  for (int32_t c = 0; c < Kp; ++c) {
    unitTin[size_t(c)] = t.tin[size_t(posParent[size_t(c)])];
  }
  for (int32_t c = 0; c < Ko; ++c) {
    for (int k = 0; k < 3; ++k) {
      unitTin[size_t(Kp + 3 * c + k)] = t.tin[size_t(oriParent[size_t(c)])];
    }
  }
  out.groups.clear(), out.extra.clear(), out.chunkStart.clear();
  std::vector<int32_t> blockMasks;
  for (int32_t u0 = 0; u0 < U; u0 += uc) {
    out.chunkStart.push_back(int32_t(out.groups.size() / 2));
    blockMasks.push_back(0);
    for (int32_t u = u0; u < std::min(U, u0 + uc); ++u) {
      const bool isPoint = u < Kp;
      for (int32_t c = 0; c < n; ++c) {
        const int32_t p = solveList[size_t(c)];
        int32_t count = 0, first = -1;
        const size_t extraAt = out.extra.size();
        for (int32_t e = t.colStart[size_t(p)]; e < t.colStart[size_t(p) + 1]; ++e) {
          const ColumnSource& cs = t.colSources[size_t(e)];
          const bool rot = cs.dof >= 3 && cs.dof < 6;
          if (cs.tin <= unitTin[size_t(u)] && unitTin[size_t(u)] < cs.tout && (rot || isPoint)) {
            const int32_t k = prefix[size_t(c)] + (e - t.colStart[size_t(p)]);
            if (count == 0) {
              first = k;
            }
            out.extra.push_back(k);
            ++count;
          }
        }
        if (count == 0) {
          continue;
        }
        if (count > 8191) {
          return false;
        }
        if (count == 1) {
          out.extra.resize(extraAt); // (a single source rides in the group word)
        }
        blockMasks.back() |= int32_t(1u << std::min(c >> 4, 31)); // (blocks beyond 31 share the last bit: n <= 208 has 13)
        out.groups.push_back(uint32_t(c) | uint32_t(u - u0) << 12 | uint32_t(count) << 18);
        out.groups.push_back(uint32_t(count == 1 ? first : int32_t(extraAt)));
      }
    }
  }
  out.chunkStart.push_back(int32_t(out.groups.size() / 2));
  out.chunkStart.insert(out.chunkStart.end(), blockMasks.begin(), blockMasks.end());
  return true;
}

} // namespace mmx
