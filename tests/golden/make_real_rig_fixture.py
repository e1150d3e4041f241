#!/usr/bin/env python3
"""The reference's real character assets as fixtures: momentum's test character with its stored motion
(momentum/examples/convert_model/test_data/character_with_motion.glb, momentum/test/resources/model_with_motion.glb:
glTF binaries with the FB_momentum extension -- skeleton nodes, parameter transform, parameter limits, motion frames and
identity offsets) read IN PLACE from the reference checkout by momentum_amd.model_io (load_gltf / load_gltf_motion), and the
parameter transform of character.model next to it cross-checked against the extension's.  What is committed is what the
loader extracted (a few hundred numbers per asset) plus an IK problem built on it:

    theta* = the stored motion frames (one batch element per frame), targets = FK(theta*) through the double oracle,
    a position + orientation constraint on every joint, start at 0, ten Gauss-Newton iterations at lambda = 0.05,
    the oracle's double solve as the expected answer.

tests/test_real_rig.py checks (here, where /root/reference exists) that loading the assets again reproduces the
fixture, and (on the GPU) that the HIP path solves it to 1e-5 of the stored answer; tests/cpp/test_real_rig.cpp does
the same through the C++ shell with SkeletonState.  /root/reference is not read at test time on the GPU box.

    python tests/golden/make_real_rig_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from momentum_amd import model_io  # noqa: E402
from momentum_amd._abi import GnOptions  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/momentum"
ASSETS = {
    "real_rig_character_with_motion.npz": os.path.join(REF, "examples/convert_model/test_data/character_with_motion.glb"),
    "real_rig_model_with_motion.npz": os.path.join(REF, "test/resources/model_with_motion.glb"),
}
MODEL_TEXT = os.path.join(REF, "examples/convert_model/test_data/character.model")
OPT = dict(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)


def load_asset(path):
    """(rig with the stored identity as transform offsets, limits, motion dict) of one GLB"""
    data = open(path, "rb").read()
    rig, limits = model_io.load_gltf(data)
    motion = model_io.load_gltf_motion(data)
    assert motion and motion["parameter_names"] == rig.param_names and motion["joint_names"] == rig.joint_names
    rig.pt_offsets = motion["identity"].astype(np.float32).copy()  # IdentityParameters = joint-parameter offsets
    return rig, limits, motion


def build(path):
    rig, limits, motion = load_asset(path)
    theta_star = motion["poses"].astype(np.float32)
    B, J = theta_star.shape[0], rig.num_joints
    allj = np.arange(J, dtype=np.int32)
    rng = np.random.default_rng(20240926)
    pos_offset = rng.uniform(-1.0, 1.0, size=(B, J, 3)).astype(np.float32)  # centimetres, like the asset's locators
    ori_offset = np.zeros((B, J, 4), np.float32)
    ori_offset[..., 3] = 1.0
    pos_target = np.zeros((B, J, 3), np.float32)
    ori_target = np.zeros((B, J, 4), np.float32)
    state_star = np.zeros((B, J, 8), np.float64)
    from tests.helpers import quat_rot

    for b in range(B):
        w = orc.skeleton_state(rig, theta_star[b].astype(np.float64), "f64")["world"]
        state_star[b] = w
        for j in range(J):
            pos_target[b, j] = w[j, :3] + quat_rot(w[j, 3:7], w[j, 7] * pos_offset[b, j].astype(np.float64))
            ori_target[b, j] = w[j, 3:7]
    ones = lambda k: np.ones((B, k), np.float32)
    cons = orc.Constraints(allj, pos_offset, pos_target, ones(J), allj, ori_offset, ori_target, ones(J))
    theta0 = np.zeros_like(theta_star)
    ref = orc.solve_batch(rig, cons, theta0, GnOptions.make(**OPT), dtype="f64")
    return dict(
        parent=rig.parent, pre_rotation=rig.pre_rotation, translation_offset=rig.translation_offset, pt_outer=rig.pt_outer,
        pt_inner=rig.pt_inner, pt_value=rig.pt_value, pt_offsets=rig.pt_offsets, joint_names=np.array(rig.joint_names),
        param_names=np.array(rig.param_names), fps=np.float64(motion["fps"]),
        limits=np.array([[l.type, l.index0, l.index1, l.weight, *list(l.v)] for l in limits], np.float64).reshape(-1, 8),
        theta_star=theta_star, state_star=state_star, pos_parent=allj, ori_parent=allj, pos_offset=pos_offset, pos_target=pos_target,
        pos_weight=ones(J), ori_offset=ori_offset, ori_target=ori_target, ori_weight=ones(J), theta0=theta0,
        theta_final=ref["theta"], final_error=ref["error"], error_history=ref["error_history"], iterations=ref["iterations"],
    )  # fmt: skip


def main():
    # the .model text next to the GLB describes the same parameter transform as the GLB's FB_momentum extension
    rig, _, _ = load_asset(ASSETS["real_rig_character_with_motion.npz"])
    sections = model_io.load_momentum_model(open(MODEL_TEXT).read())
    pnames, triplets, _ = model_io.parse_parameter_transform(sections["ParameterTransform"], rig.joint_names)
    outer, inner, value = model_io._csr(triplets, 7 * rig.num_joints)
    assert list(pnames) == rig.param_names and np.array_equal(outer, rig.pt_outer) and np.array_equal(inner, rig.pt_inner)
    assert np.array_equal(np.asarray(value, np.float32), rig.pt_value)
    for name, path in ASSETS.items():
        np.savez_compressed(os.path.join(HERE, name), **build(path))
        print("wrote", name)


if __name__ == "__main__":
    main()
