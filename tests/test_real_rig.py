"""The reference's real character assets end to end: momentum's test character with stored motion
(examples/convert_model/test_data/character_with_motion.glb + character.model, test/resources/model_with_motion.glb)
-- skeleton, parameter transform, limits, motion frames and identity offsets read from the FB_momentum extension
(momentum/io/gltf/gltf_io.cpp, gltf_animation_io.cpp:72-112) by momentum_amd.model_io.  The stored motion frames are
theta*, the targets FK(theta*), and the solve from theta = 0 is held to the oracle's double solve.

The committed fixtures (tests/golden/real_rig_*.npz, tests/golden/make_real_rig_fixture.py) hold what the loader
extracted from the assets plus the problem and the expected answer: the GPU box has no reference checkout."""
import os

import numpy as np
import pytest

from momentum_amd import model_io
from momentum_amd._abi import GnOptions
from momentum_amd.rigs import Rig

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["real_rig_character_with_motion.npz", "real_rig_model_with_motion.npz"]
OPT = dict(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)


def fixture_rig(g) -> Rig:
    return Rig(g["parent"], g["pre_rotation"], g["translation_offset"], g["pt_outer"], g["pt_inner"], g["pt_value"], g["pt_offsets"],
               len(g["param_names"]), [str(x) for x in g["joint_names"]], [str(x) for x in g["param_names"]])  # fmt: skip


def fixture_cons(orc, g):
    return orc.Constraints(g["pos_parent"], g["pos_offset"], g["pos_target"], g["pos_weight"],
                           g["ori_parent"], g["ori_offset"], g["ori_target"], g["ori_weight"])  # fmt: skip


@pytest.mark.parametrize("name", FIXTURES)
def test_assets_load_to_the_committed_fixture(name):
    """Where the reference checkout is present: loading the GLB again gives the fixture's rig, limits and motion."""
    from tests.golden import make_real_rig_fixture as mk

    path = mk.ASSETS[name]
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    g = np.load(os.path.join(HERE, name))
    rig, limits, motion = mk.load_asset(path)
    assert rig.joint_names == ["root", "joint1", "joint2"] and list(rig.parent) == [-1, 0, 1]
    assert rig.param_names == [str(x) for x in g["param_names"]] and rig.num_params == 10
    for f in ("parent", "pre_rotation", "translation_offset", "pt_outer", "pt_inner", "pt_value", "pt_offsets"):
        assert np.array_equal(getattr(rig, f), g[f]), f
    # metres in the file, centimetres in momentum: the test character's bones are one unit long
    assert np.allclose(rig.translation_offset, [[0, 0, 0], [0, 1, 0], [0, 1, 0]], atol=1e-6)
    assert np.array_equal(motion["poses"], g["theta_star"]) and motion["fps"] == float(g["fps"])
    # one MinMax limit on root_tx, [-0.1, 0.1] (both encodings of the limits array: [lo, hi] and [[lo, hi]])
    assert len(limits) == 1 and limits[0].index0 == 0 and np.allclose(list(limits[0].v)[:2], [-0.1, 0.1])
    assert np.array_equal(np.array([[l.type, l.index0, l.index1, l.weight, *list(l.v)] for l in limits], np.float64), g["limits"])


def test_model_file_describes_the_same_parameter_transform():
    from tests.golden import make_real_rig_fixture as mk

    if not os.path.exists(mk.MODEL_TEXT):
        pytest.skip("reference checkout not present")
    g = np.load(os.path.join(HERE, FIXTURES[0]))
    sections = model_io.load_momentum_model(open(mk.MODEL_TEXT).read())
    pnames, triplets, offsets = model_io.parse_parameter_transform(sections["ParameterTransform"], [str(x) for x in g["joint_names"]])
    outer, inner, value = model_io._csr(triplets, 7 * len(g["joint_names"]))
    assert pnames == [str(x) for x in g["param_names"]]
    assert np.array_equal(outer, g["pt_outer"]) and np.array_equal(inner, g["pt_inner"]) and np.array_equal(value, g["pt_value"])
    assert not offsets.any()
    # the shared parameter drives two joints with 0.5 each (momentum/test/character/character_helpers.cpp:137-138)
    A = fixture_rig(g).dense_transform()
    k = [str(x) for x in g["param_names"]].index("shared_rz")
    assert np.flatnonzero(A[:, k]).tolist() == [7 * 1 + 5, 7 * 2 + 5] and np.allclose(A[[12, 19], k], 0.5)


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_reproduces_the_real_rig_fixture(orc, name):
    g = np.load(os.path.join(HERE, name))
    rig = fixture_rig(g)
    for b in range(g["theta_star"].shape[0]):  # the stored states are FK of the stored motion
        w = orc.skeleton_state(rig, g["theta_star"][b].astype(np.float64), "f64")["world"]
        assert np.abs(w - g["state_star"][b]).max() <= 1e-12
    ref = orc.solve_batch(rig, fixture_cons(orc, g), g["theta0"], GnOptions.make(**OPT), dtype="f64")
    assert np.abs(ref["theta"] - g["theta_final"]).max() <= 1e-10 and np.array_equal(ref["iterations"], g["iterations"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_real_rig_solves_on_the_gpu(torch_cuda, orc, name):
    from momentum_amd import capi

    torch = torch_cuda
    g = np.load(os.path.join(HERE, name))
    rig = fixture_rig(g)
    B = g["theta_star"].shape[0]
    pb = capi.Problem(capi.RigHandle(rig, 0), B, g["pos_parent"], g["ori_parent"])
    dev = pb.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    pb.set_constraints(t(g["pos_offset"]), t(g["pos_target"]), t(g["pos_weight"]), t(g["ori_offset"]), t(g["ori_target"]), t(g["ori_weight"]))
    # forward pass at the stored motion: the asset's poses through mmx_eval_skeleton_state against the oracle's double FK
    st = pb.skeleton_state(t(g["theta_star"])).cpu().numpy().astype(np.float64)
    ref_st = g["state_star"]
    sign = np.sign((st[..., 3:7] * ref_st[..., 3:7]).sum(-1, keepdims=True))
    assert np.abs(st[..., :3] - ref_st[..., :3]).max() <= 5e-6 * max(1.0, np.abs(ref_st[..., :3]).max())
    assert np.abs(st[..., 3:7] - sign * ref_st[..., 3:7]).max() <= 5e-6 and np.abs(st[..., 7] - ref_st[..., 7]).max() <= 5e-6
    for route in ("auto", "wide"):
        pb.set_route(route)
        out = pb.solve(t(g["theta0"]), GnOptions.make(**OPT), want_history=True)
        torch.cuda.synchronize()
        th = out["theta"].cpu().numpy().astype(np.float64)
        # (the first stored frame of character_with_motion.glb is the rest pose: its answer is 0, held absolutely)
        rel = np.linalg.norm(th - g["theta_final"], axis=1) / np.maximum(np.linalg.norm(g["theta_final"], axis=1), 1e-2)
        assert rel.max() <= 1e-5, (name, route, rel)
        assert np.array_equal(out["iterations"].cpu().numpy(), g["iterations"]) and np.all(out["status"].cpu().numpy() & 3 == 0)
        h = out["error_history"].cpu().numpy()
        assert np.all(np.abs(h - g["error_history"]) <= 1e-4 * np.abs(g["error_history"]) + 1e-7 * g["error_history"][:, :1] + 1e-12)  # (fp32 noise floor of a converged fit; the rest-pose frame starts AT its solution)
    # the double instantiation: 1e-10
    out = pb.solve_f64(torch.from_numpy(g["theta0"].astype(np.float64)).to(dev), GnOptions.make(**OPT))
    torch.cuda.synchronize()
    th = out["theta"].cpu().numpy()
    assert (np.linalg.norm(th - g["theta_final"], axis=1) / np.maximum(np.linalg.norm(g["theta_final"], axis=1), 1e-2)).max() <= 1e-10
