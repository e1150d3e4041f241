"""Bit-exact checks of the integer bookkeeping (north_star: 'bit-exact on joint-index
bookkeeping'): the library's host tables vs the oracle's parent-chasing restatement.  CPU only."""
import numpy as np
import pytest

from momentum_amd import capi, make_humanoid72, make_rig300, make_test_character
from momentum_amd import build as mbuild


@pytest.fixture(scope="module", autouse=True)
def _built():
    mbuild.build()


RIGS = {
    "chain24": lambda: make_test_character(24),
    "humanoid72": lambda: make_humanoid72(),
    "humanoid72_p219": lambda: make_humanoid72(variant="p219"),
    "rig300": lambda: make_rig300(),
}


@pytest.mark.parametrize("name", list(RIGS))
def test_dfs_intervals_equal_parent_chasing(orc, name):
    rig = RIGS[name]()
    t = capi.host_tables(rig)
    J = rig.num_joints
    anc = orc.ancestor_matrix(rig)  # anc[a, j] = a is j or an ancestor of j (while-loop of the reference)
    tin, tout = t["tin"], t["tout"]
    mine = (tin[:, None] <= tin[None, :]) & (tin[None, :] < tout[:, None])
    assert np.array_equal(mine, anc.astype(bool))
    assert sorted(tin.tolist()) == list(range(J))  # a permutation: pre-order numbering
    assert np.array_equal(t["level"], rig.depth())


@pytest.mark.parametrize("name", list(RIGS))
def test_enabled_list_and_active_joint_params(orc, name):
    rig = RIGS[name]()
    rng = np.random.default_rng(5)
    for trial in range(3):
        en = (rng.uniform(size=rig.num_params) < (1.0 if trial == 0 else 0.6)).astype(np.uint8)
        t = capi.host_tables(rig, en)
        assert np.array_equal(t["enabled_list"], np.flatnonzero(en).astype(np.int32))
        assert np.array_equal(t["active_joint_params"], orc.active_joint_params(rig, en))


def test_humanoid_shapes():
    r = make_humanoid72()
    assert (r.num_joints, r.num_params, int(r.depth().max())) == (72, 128, 12)
    r = make_humanoid72(variant="p219")
    assert (r.num_joints, r.num_params) == (72, 219)
    r = make_rig300()
    assert (r.num_joints, r.num_params) == (300, 300) and int(r.depth().max()) <= 16
    r = make_test_character(24)
    assert (r.num_joints, r.num_params) == (24, 31)


# ---- the solvers' column order and the tile structure of their factor (mmx_host_elimination_order / _tile_structure)
def _param_joints(rig, en=None):
    """joints each (enabled) parameter drives, from the parameter transform's CSR rows."""
    out = [set() for _ in range(rig.num_params)]
    for r in range(7 * rig.num_joints):
        for k in range(rig.pt_outer[r], rig.pt_outer[r + 1]):
            out[int(rig.pt_inner[k])].add(r // 7)
    return out


def _related(rig, order, anc):
    """H(row, col) can be non-zero iff a joint of the one parameter is an ancestor-or-self of a joint of the other."""
    pj = _param_joints(rig)
    n = len(order)
    rel = np.zeros((n, n), np.uint8)
    for a in range(n):
        for b in range(a):
            rel[a, b] = any(anc[x, y] or anc[y, x] for x in pj[order[a]] for y in pj[order[b]])
    return rel


@pytest.mark.parametrize("name", list(RIGS))
def test_elimination_order_is_a_post_order_of_the_enabled_parameters(orc, name):
    rig = RIGS[name]()
    rng = np.random.default_rng(8)
    anc = orc.ancestor_matrix(rig).astype(bool)
    pj = _param_joints(rig)
    for trial in range(2):
        en = (rng.uniform(size=rig.num_params) < (1.0 if trial == 0 else 0.7)).astype(np.uint8)
        t = capi.host_tables(rig, en)
        order = t["elimination_order"]
        assert sorted(order.tolist()) == np.flatnonzero(en).tolist()  # a permutation of the enabled list
        # children before parents: a parameter that drives ONE joint never precedes a single-joint parameter of a
        # strict descendant of that joint
        single = [p for p in order if len(pj[p]) == 1]
        pos = {int(p): i for i, p in enumerate(order)}
        for p in single:
            (a,) = pj[p]
            for q in single:
                (b,) = pj[q]
                if a != b and anc[a, b]:  # a strict ancestor of b
                    assert pos[int(q)] < pos[int(p)], (p, q)


def _numeric_fill(rel, seed=0):
    """Pattern of the Cholesky factor of a random SPD matrix with the given pattern (dense double arithmetic)."""
    n = rel.shape[0]
    rng = np.random.default_rng(seed)
    A = np.tril(rel, -1) * rng.uniform(0.5, 1.0, size=(n, n))
    A = A + A.T + np.eye(n) * (n + 1.0)
    L = np.linalg.cholesky(A)
    return np.abs(np.tril(L)) > 1e-14


@pytest.mark.parametrize("name", ["humanoid72", "humanoid72_p219", "rig300"])
def test_tile_structure_covers_the_numerical_factor_and_is_sparse_in_elimination_order(orc, name):
    rig = RIGS[name]()
    anc = orc.ancestor_matrix(rig).astype(bool)
    t = capi.host_tables(rig)
    dense_products = lambda NB: NB * (NB * NB - 1) // 6
    results = {}
    for label, order in (("elimination", t["elimination_order"]), ("parameter", t["enabled_list"])):
        rel = _related(rig, [int(p) for p in order], anc)
        n = rel.shape[0]
        NB = (n + 15) // 16
        ts = capi.host_tile_structure(rel)
        Lnz = _numeric_fill(rel)
        tiles = 0
        for I in range(NB):
            for Jc in range(I + 1):
                has = bool(Lnz[16 * I : 16 * I + 16, 16 * Jc : 16 * Jc + 16].any())
                bit = bool(ts["row_mask"][I] >> Jc & 1)
                assert bit or not has, (label, I, Jc)  # every numerically non-zero tile is in the structure
                assert bit == bool(ts["col_mask"][Jc] >> I & 1)
                tiles += bit
        # the products the masked factorisation performs, recounted from the masks
        prod = sum(
            bin(int(ts["row_mask"][I]) & int(ts["row_mask"][k]) & ((1 << k) - 1)).count("1")
            for I in range(NB)
            for k in range(I + 1)
            if ts["row_mask"][I] >> k & 1
        )
        assert prod == ts["products"]
        results[label] = (tiles, ts["products"], NB)
    (te, pe, NB), (tp, pp, _) = results["elimination"], results["parameter"]
    assert pp == dense_products(NB) and tp == NB * (NB + 1) // 2  # root first: the factor fills completely
    assert 3 * pe <= pp and te < tp, results  # leaves first: at most a third of the tile products


def test_tile_structure_fill_in():
    """An arrow pattern pointing the wrong way (first row block coupled to all) fills completely; the same arrow with the
    hub last has no fill at all."""
    n = 96
    rel = np.zeros((n, n), np.uint8)
    rel[16:, :16] = 1  # every later parameter coupled to the first block
    ts = capi.host_tile_structure(rel)
    assert all(int(ts["row_mask"][I]) == (1 << (I + 1)) - 1 for I in range(6))
    rel = np.zeros((n, n), np.uint8)
    rel[80:, :80] = 1  # the hub last
    ts = capi.host_tile_structure(rel)
    assert [int(ts["row_mask"][I]) for I in range(6)] == [1, 2, 4, 8, 16, 63]
    assert ts["products"] == 5  # the hub's diagonal tile: one product per earlier column


@pytest.mark.parametrize("shape", ["bushy", "star", "chain"])
@pytest.mark.parametrize("seed", range(4))
def test_tile_structure_holds_the_real_normal_equations_of_random_rigs(orc, seed, shape):
    """Random trees with shared parameters (one parameter driving rotations of several unrelated joints -- the case the
    pattern's 'some joint of the one is an ancestor-or-self of some joint of the other' is written for): J^T J of the
    oracle's double Jacobian at a random pose, in elimination order, has no entry outside the pattern, and its numerical
    Cholesky factor no tile outside the symbolic structure."""
    from tests.helpers import make_problem
    from tests.test_gpu_fuzz import random_rig

    rng = np.random.default_rng(1000 * seed + len(shape))
    J = int(rng.integers(20, 90))
    rig = random_rig(rng, J, shape)
    anc = orc.ancestor_matrix(rig).astype(bool)
    order = [int(p) for p in capi.host_tables(rig)["elimination_order"]]
    assert sorted(order) == list(range(rig.num_params))
    rel = _related(rig, order, anc)
    ts = capi.host_tile_structure(rel)
    n = len(order)
    NB = (n + 15) // 16
    pp = rng.choice(J, size=min(J, 12), replace=False)
    op = rng.choice(J, size=min(J, 5), replace=False)
    cons, th0, _ = make_problem(rig, pp, op, 1, seed=seed, perturb=0.3)
    theta = rng.uniform(-0.3, 0.3, size=rig.num_params)
    Jm, r, e = orc.eval_jacobian(rig, cons.instance(0), theta, dtype="f64")
    H = (Jm.T @ Jm)[np.ix_(order, order)]
    nz = np.abs(np.tril(H, -1)) > 1e-12 * max(1.0, np.abs(H).max())
    assert not np.any(nz & ~rel.astype(bool))  # entry level: nothing outside the ancestor pattern
    Lnz = np.abs(np.tril(np.linalg.cholesky(H + 0.05 * np.eye(n)))) > 1e-13 * max(1.0, np.abs(H).max())
    for I in range(NB):
        for Jc in range(I + 1):
            if Lnz[16 * I : 16 * I + 16, 16 * Jc : 16 * Jc + 16].any():
                assert ts["row_mask"][I] >> Jc & 1, (I, Jc)
    if shape == "chain":  # every joint an ancestor of the ones below: wherever the parameters reach, the structure is dense
        assert ts["products"] > 0 or NB == 1


def test_tile_structure_argument_checks():
    """n outside 0..512 is refused with a message (32 mask bits per block row); n = 0 is the empty structure."""
    import ctypes as C

    row, col = np.zeros(32, np.uint32), np.zeros(32, np.uint32)
    prod = C.c_int64(-1)
    rel = np.zeros((1, 1), np.uint8)
    rc = capi.lib().mmx_host_tile_structure(C.c_int32(513), capi.as_ptr(rel, C.c_uint8), capi.as_ptr(row, C.c_uint32), capi.as_ptr(col, C.c_uint32), C.byref(prod))
    assert rc != 0 and b"512" in capi.lib().mmx_last_error()
    rc = capi.lib().mmx_host_tile_structure(C.c_int32(0), None, capi.as_ptr(row, C.c_uint32), capi.as_ptr(col, C.c_uint32), C.byref(prod))
    assert rc == 0 and prod.value == 0 and not row.any() and not col.any()
    ts = capi.host_tile_structure(np.zeros((512, 512), np.uint8))  # the largest system: 32 diagonal tiles, nothing else
    assert [int(x) for x in ts["row_mask"]] == [1 << i for i in range(32)] and ts["products"] == 0


@pytest.mark.parametrize("seed", range(6))
def test_tile_structure_is_closed_under_elimination(seed):
    """What the masked factorisation relies on (tiledFactorPairs, choleskyFactorResidentKernel): if two tiles (I, k) and
    (J, k), I >= J > k, of a block column exist, tile (I, J) exists -- so the product L(I,k) L(J,k)^T always has a home, and
    column k + 1's missing term from column k's panel only ever touches rows column k + 1 holds.  Random patterns, and the
    masked left-looking algorithm on the tile grid (only listed tiles are read or written) against a dense Cholesky."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(40, 200))
    NB = (n + 15) // 16
    rel = np.tril((rng.uniform(size=(n, n)) < rng.uniform(0.005, 0.05)).astype(np.uint8), -1)
    ts = capi.host_tile_structure(rel)
    M = np.array([[bool(ts["row_mask"][I] >> Jc & 1) for Jc in range(NB)] for I in range(NB)])
    for k in range(NB):
        rows = [I for I in range(k + 1, NB) if M[I, k]]
        for I in rows:
            for Jc in rows:
                if Jc <= I:
                    assert M[I, Jc], (k, I, Jc)
        assert M[k, k] and bool(ts["col_mask"][k] >> k & 1)
    # masked block algorithm in double on a padded SPD matrix with that pattern
    NP = 16 * NB
    A = np.zeros((NP, NP))
    A[:n, :n] = np.tril(rel, -1) * rng.uniform(0.5, 1.0, size=(n, n))
    A = A + A.T + np.eye(NP) * (n + 1.0)
    tile = lambda X, I, Jc: X[16 * I : 16 * I + 16, 16 * Jc : 16 * Jc + 16]
    L = np.full((NP, NP), np.nan)  # tiles outside the structure are never written: NaN would poison a wrong read
    products = 0
    for k in range(NB):
        for I in range(k, NB):
            if not M[I, k]:
                continue
            C = tile(A, I, k).copy()
            for j in range(k):
                if M[I, j] and M[k, j]:
                    C -= tile(L, I, j) @ tile(L, k, j).T
                    products += 1
            if I == k:
                tile(L, k, k)[:] = np.linalg.cholesky(C)
            else:
                tile(L, I, k)[:] = np.linalg.solve(tile(L, k, k), C.T).T
    assert products == ts["products"]
    Ld = np.linalg.cholesky(A)
    for I in range(NB):
        for Jc in range(I + 1):
            if M[I, Jc]:
                assert np.abs(tile(L, I, Jc) - tile(Ld, I, Jc)).max() <= 1e-10
            else:
                assert np.abs(tile(Ld, I, Jc)).max() <= 1e-12


def _check_level_schedule(rel):
    """The resident factor kernel's level schedule (TileMasks::levelSteps, mmx_host_tile_level_schedule): every block column
    exactly once; a column only after every column it has a tile in the row of (its left-looking update reads those); the
    columns of a step on disjoint waves of the 4-wave workgroup, each with enough of them for its panel (16 + 48 x waves
    rows); a panel beyond 208 rows alone in its step."""
    ts = capi.host_tile_structure(rel)
    steps = capi.host_tile_level_schedule(rel)
    n = rel.shape[0]
    NB = (n + 15) // 16
    done_before = {}
    seen = []
    for s, step in enumerate(steps):
        assert 1 <= len(step) <= 4
        waves_used = set()
        for k, w0, nw in step:
            assert 0 <= k < NB
            seen.append(k)
            nt = bin(int(ts["col_mask"][k])).count("1")
            if nw == 15:
                assert len(step) == 1 and 16 * nt > 208
            else:
                assert nw >= 1 and w0 + nw <= 4 and 16 + 48 * nw >= 16 * nt
                ws = set(range(w0, w0 + nw))
                assert not (ws & waves_used)
                waves_used |= ws
            deps = [j for j in range(k) if int(ts["row_mask"][k]) >> j & 1]
            assert all(j in done_before and done_before[j] < s for j in deps), (k, deps, s)
        for k, _, _ in step:
            done_before[k] = s
    assert sorted(seen) == list(range(NB))
    return len(steps), NB


@pytest.mark.parametrize("name", ["humanoid72", "humanoid72_p219", "rig300"])
def test_level_schedule_of_the_rigs(orc, name):
    rig = RIGS[name]()
    anc = orc.ancestor_matrix(rig).astype(bool)
    t = capi.host_tables(rig)
    rel = _related(rig, [int(p) for p in t["elimination_order"]], anc)
    steps, NB = _check_level_schedule(rel)
    assert steps <= NB
    if name == "rig300":
        assert steps <= NB - 5  # the finger chains go side by side with the spine's: 19 block columns in at most 14 steps


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_level_schedule_of_random_patterns(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(17, 512))
    rel = np.zeros((n, n), np.uint8)
    # a random forest on the 16-blocks: a block couples to its ancestors' blocks (plus a few stray entries)
    NB = (n + 15) // 16
    parent = [-1] + [int(rng.integers(0, i)) if rng.random() < 0.8 else -1 for i in range(1, NB)]
    order = list(range(NB))[::-1]  # children before parents: block i's ancestors have smaller index -> reverse numbering
    pos = {b: i for i, b in enumerate(order)}
    for b in range(NB):
        a = parent[b]
        while a >= 0:
            lo, hi = sorted((pos[b], pos[a]))
            rel[16 * hi : min(16 * hi + 16, n), 16 * lo : min(16 * lo + 16, n)] = 1
            a = parent[a]
    for _ in range(3):
        i, j = sorted(rng.integers(0, n, size=2))
        if i != j:
            rel[j, i] = 1
    _check_level_schedule(np.tril(rel, -1))
    _check_level_schedule(np.tril(np.ones((n, n), np.uint8), -1))  # dense: one column per step


@pytest.mark.parametrize("name,uc", [("humanoid72", 16), ("test_character_8", 5), ("rig300", 16)])
def test_f64_assembly_list_against_a_numpy_restatement(orc, name, uc):
    """mmx_solve_f64's assembly list (mmx_host_f64_assembly_list / buildF64AssemblyListHost): exactly the entries (column,
    unit) of J that have an applicable source -- the source's joint an ancestor-or-self of the unit's joint; translation and
    scale dofs only for points (joint_error_function-inl.h:248-291) --, each with its sources' positions in the packed
    table, and per chunk the mask of 16-column blocks that have an entry.  Restated from the parameter transform and the
    ancestor matrix."""
    rig = RIGS[name]() if name in RIGS else make_test_character(8)
    J, P = rig.num_joints, rig.num_params
    anc = orc.ancestor_matrix(rig).astype(bool)  # anc[a, j]: a is an ancestor-or-self of j
    rng = np.random.default_rng(3)
    pos_parent = rng.integers(0, J, size=min(J, 21)).astype(np.int32)
    ori_parent = rng.integers(0, J, size=min(J, 9)).astype(np.int32)
    solve_list = np.sort(rng.choice(P, size=max(1, (3 * P) // 4), replace=False)).astype(np.int32)
    chunks = capi.host_f64_assembly_list(rig, solve_list, pos_parent, ori_parent, uc)
    # sources of a parameter: the non-zeros of its column of the parameter transform, in row order (= colSources order)
    src = [[] for _ in range(P)]
    for r in range(7 * J):
        for k in range(rig.pt_outer[r], rig.pt_outer[r + 1]):
            src[int(rig.pt_inner[k])].append((r // 7, r % 7))
    prefix = np.concatenate([[0], np.cumsum([len(src[p]) for p in solve_list])])
    Kp, U = len(pos_parent), len(pos_parent) + 3 * len(ori_parent)
    unit_joint = [int(pos_parent[u]) if u < Kp else int(ori_parent[(u - Kp) // 3]) for u in range(U)]
    assert len(chunks) == (U + uc - 1) // uc
    for ch, chunk in enumerate(chunks):
        want, mask = [], 0
        for ul in range(min(uc, U - ch * uc)):
            u = ch * uc + ul
            for c, p in enumerate(solve_list):
                ks = []
                for e, (joint, dof) in enumerate(src[int(p)]):
                    if anc[joint, unit_joint[u]] and (3 <= dof < 6 or u < Kp):
                        ks.append(int(prefix[c]) + e)
                if ks:
                    want.append((c, ul, ks))
                    mask |= 1 << min(c >> 4, 31)
        assert chunk["entries"] == want, (name, ch)
        assert chunk["block_mask"] == mask
