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
