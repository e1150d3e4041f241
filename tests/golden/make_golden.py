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


def dump(name, rig, pos_parent, ori_parent, batch, seed, perturb):
    cons, th0, ths = make_problem(rig, pos_parent, ori_parent, batch, seed=seed, perturb=perturb)
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
        iterations=ref["iterations"], jac0=J0.astype(np.float32), res0=r0, err0=np.float64(e0), state_star0=st0,
    )  # fmt: skip
    print(name, "batch", batch, "final error", ref["error"])


if __name__ == "__main__":
    # BASELINE configs[0]: 24-joint chain, 3 position constraints (parents 23, 12, 5)
    dump("cfg1_chain24.npz", make_test_character(24), [23, 12, 5], [], 4, 12345, 0.1)
    # BASELINE configs[1]: 72-joint humanoid, position + orientation on 16 landmark joints
    rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
    lm = humanoid72_landmark_joints(rig)
    dump("cfg2_humanoid72.npz", rig, lm, lm, 4, 12345, 0.3)
