"""Validates the tree-structured formulas (tests/tree_algebra_np.py: J^T J from subtree moments,
J^T y by an adjoint pass, J d by a tangent pass) against the oracle's explicit Jacobian.  CPU only."""
import numpy as np
import pytest

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character
from tests import tree_algebra_np as ta
from tests.helpers import make_problem, quat_rot


def _units(orc, rig, cons, theta):
    """Units exactly as the kernels define them (momentum_amd/csrc/mmx_device.hpp evalUnit)."""
    st = orc.skeleton_state(rig, theta, "f64")
    W = st["world"]
    units = []
    for c in range(cons.Kp):
        j = cons.pos_parent[c]
        w = W[j]
        p = w[:3] + quat_rot(w[3:7], w[7] * cons.pos_offset[c].astype(np.float64))
        wt = float(cons.pos_weight[c]) * float(np.float32(cons.pos_function_weight))
        units.append(dict(joint=j, point=True, p=p, sigma=np.sqrt(wt), f=p - cons.pos_target[c]))
    def qmat(q):
        q = q / np.linalg.norm(q)
        return np.stack([quat_rot(q, e) for e in np.eye(3)], axis=1)
    for c in range(cons.Ko):
        j = cons.ori_parent[c]
        w = W[j]
        Ro, Rt = qmat(cons.ori_offset[c].astype(np.float64)), qmat(cons.ori_target[c].astype(np.float64))
        wt = float(cons.ori_weight[c]) * float(np.float32(cons.ori_function_weight))
        for k in range(3):
            p = quat_rot(w[3:7], Ro[:, k])
            units.append(dict(joint=j, point=False, p=p, sigma=np.sqrt(wt), f=p - Rt[:, k]))
    return st, units


CASES = {
    "chain8": (lambda: make_test_character(8), [7, 3, 1, 3], [6, 2]),
    "humanoid": (lambda: make_humanoid72(unit=0.01), "lm", "lm"),
    "humanoid_dense": (lambda: make_humanoid72(unit=0.01), list(range(0, 72, 3)), list(range(1, 72, 4))),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("subset", [False, True])
def test_tree_formulas_match_explicit_jacobian(orc, name, subset):
    mk, pp, op = CASES[name]
    rig = mk()
    if pp == "lm":
        pp = op = humanoid72_landmark_joints(rig)
    cons, th0, ths = make_problem(rig, pp, op, 1, seed=31, perturb=0.4, random_offsets=True, weights="random")
    c = cons.instance(0)
    c.pos_function_weight, c.ori_function_weight = 0.7, 1.9
    rng = np.random.default_rng(4)
    theta = rng.uniform(-0.4, 0.4, rig.num_params)
    en = (rng.uniform(size=rig.num_params) < 0.7).astype(np.uint8) if subset else None
    J, r, err = orc.eval_jacobian(rig, c, theta, enabled=en, dtype="f64")
    st, units = _units(orc, rig, c, theta)
    tree = ta.Tree(rig, st, en)
    # residual rows and error from the units
    rr = np.concatenate([u["sigma"] * u["f"] for u in units])
    assert np.abs(rr - r).max() <= 1e-12 * max(1, np.abs(r).max())
    # J^T r via the adjoint pass
    y = [u["sigma"] * (u["sigma"] * u["f"]) for u in units]
    S = ta.subtree_sums(tree, units, y)
    g = ta.jt_times(tree, S)
    gref = J.T @ r
    assert np.abs(g - gref).max() <= 1e-11 * max(1, np.abs(gref).max())
    # J d via the tangent pass
    d = rng.normal(size=rig.num_params)
    if en is not None:
        d[en == 0] = 0
    Jd = ta.j_times(tree, units, d)
    assert np.abs(Jd - J @ d).max() <= 1e-11 * max(1, np.abs(J @ d).max())
    # J^T J from the subtree moments
    H = ta.jtj(tree, S)
    Href = J.T @ J
    assert np.abs(H - Href).max() <= 1e-10 * max(1, np.abs(Href).max())
