#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle (double
precision).  The reference itself cannot be built or imported in this container (no Eigen / GSL /
fmt / spdlog / dispenso), so these are NOT outputs of Meta's binary: they are
outputs of the oracle restatement, which is pinned to the reference's own golden vectors by
tests/test_oracle_golden.py.  They freeze the expected answers of BASELINE configs[0] and
configs[1] so that (a) the oracle cannot drift silently and (b) the HIP path is checked against
bytes in the repository, not only against code that runs next to it.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character  # noqa: E402
from momentum_amd._abi import GnOptions  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import make_problem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def limits_to_array(limits):
    """[N,8] float64 rows (type, index0, index1, weight, v0..v3) of _abi.ParameterLimit"""
    return np.array([[l.type, l.index0, l.index1, l.weight, *list(l.v)] for l in limits], dtype=np.float64).reshape(-1, 8)


def blocks_to_dict(blocks):
    """flat npz entries jb<i>_<field> of a list of _abi.JointBlock"""
    out = {"num_joint_blocks": np.int64(len(blocks))}
    for i, b in enumerate(blocks):
        out[f"jb{i}_type"] = np.int64(b.type)
        out[f"jb{i}_parent"] = b.parent
        out[f"jb{i}_fw_loss"] = np.array([b.function_weight, b.loss[0], b.loss[1]], np.float64)
        for f in ("weight", "global_", "local_point", "local_dir", "plane_d"):
            a = getattr(b, f)
            if a is not None:
                out[f"jb{i}_{f}"] = np.asarray(a, np.float32)
    return out


def dump(name, rig, pos_parent, ori_parent, batch, seed, perturb, extras=False, joint_blocks=False):
    cons, th0, ths = make_problem(rig, pos_parent, ori_parent, batch, seed=seed, perturb=perturb)
    extra = {}
    if joint_blocks:
        # one block of every further JointErrorFunction specialisation
        from tests.test_oracle_joint_blocks import TYPES, make_block

        rng = np.random.default_rng(seed + 5)
        blocks = []
        for i, ty in enumerate(TYPES.values()):
            loss = (0.0, 0.5) if i == 2 else (2.0, 1.0)
            blocks.append(make_block(ty, rng.choice(rig.num_joints, size=3), rng, weight=1.0, batch=batch, function_weight=0.5 + 0.1 * i, loss=loss))
        cons = orc.Constraints(
            cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
            joint_blocks=blocks,
        )  # fmt: skip
        extra = blocks_to_dict(blocks)
    if extras:
        # parameter limits, a model-parameter prior and a Cauchy loss on the position block
        from momentum_amd._abi import ParameterLimit

        P = rig.num_params
        rng = np.random.default_rng(seed + 99)
        limits = [
            ParameterLimit.minmax(7, -0.05, 0.05, 2.0),
            ParameterLimit.minmax(20, -0.1, 0.02, 1.0),
            ParameterLimit.linear(9, 12, 0.5, 0.05, weight=1.5),
            ParameterLimit.linear(30, 31, 1.0, -0.1, -0.1, float(np.finfo(np.float32).max), weight=0.5),
            ParameterLimit.halfplane(15, 16, 0.6, 0.8, 0.05, 1.0),
        ]
        mt = rng.uniform(-0.1, 0.1, size=(batch, P)).astype(np.float32)
        mw = rng.uniform(-0.2, 1.0, size=(batch, P)).astype(np.float32)
        cons = orc.Constraints(
            cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
            limits=limits, limit_function_weight=0.8, model_target=mt, model_weights=mw, model_function_weight=1.2, pos_loss=(0.0, 0.1),
        )  # fmt: skip
        extra = dict(limits=limits_to_array(limits), limit_function_weight=0.8, model_target=mt, model_weights=mw,
                     model_function_weight=1.2, pos_loss=np.array([0.0, 0.1]), ori_loss=np.array([2.0, 1.0]))  # fmt: skip
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    J0, r0, e0 = orc.eval_jacobian(rig, cons.instance(0), th0[0].astype(np.float64), dtype="f64")
    st0 = orc.skeleton_state(rig, ths[0].astype(np.float64), "f64")["world"]
    np.savez_compressed(
        os.path.join(HERE, name),
        pos_parent=cons.pos_parent, ori_parent=cons.ori_parent,
        pos_offset=cons.pos_offset, pos_target=cons.pos_target, pos_weight=cons.pos_weight,
        ori_offset=cons.ori_offset, ori_target=cons.ori_target, ori_weight=cons.ori_weight,
        theta0=th0, theta_star=ths, theta_final=ref["theta"], final_error=ref["error"], error_history=ref["error_history"],
        iterations=ref["iterations"], jac0=J0.astype(np.float32), res0=r0, err0=np.float64(e0), state_star0=st0, **extra,
    )  # fmt: skip
    print(name, "batch", batch, "final error", ref["error"])


if __name__ == "__main__":
    # BASELINE configs[0]: 24-joint chain, 3 position constraints (parents 23, 12, 5)
    dump("cfg1_chain24.npz", make_test_character(24), [23, 12, 5], [], 4, 12345, 0.1)
    # BASELINE configs[1]: 72-joint humanoid, position + orientation on 16 landmark joints
    rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
    lm = humanoid72_landmark_joints(rig)
    dump("cfg2_humanoid72.npz", rig, lm, lm, 4, 12345, 0.3)
    # the same with parameter limits, a model-parameter prior and a robust loss
    dump("cfg2_limits_prior_cauchy.npz", rig, lm, lm, 4, 4321, 0.3, extras=True)
    # the same with Plane / HalfPlane / Aim / FixedAxis / Normal constraint blocks (explicit-Jacobian path)
    dump("cfg2_joint_blocks.npz", rig, lm, lm, 4, 2468, 0.3, joint_blocks=True)
