"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
double-precision oracle): the oracle must keep reproducing them (CPU), and the HIP path must match
them through the C ABI (GPU)."""
import os

import numpy as np
import pytest

from momentum_amd import make_humanoid72, make_test_character
from momentum_amd._abi import GnOptions

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {
    "cfg1_chain24.npz": lambda: make_test_character(24),
    "cfg2_humanoid72.npz": lambda: make_humanoid72(seed=12345, variant="p128", unit=0.01),
    "cfg2_limits_prior_cauchy.npz": lambda: make_humanoid72(seed=12345, variant="p128", unit=0.01),
    "cfg2_joint_blocks.npz": lambda: make_humanoid72(seed=12345, variant="p128", unit=0.01),
}
OPT = dict(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)


def _extras(g):
    """keyword arguments of the optional blocks stored in a fixture (limits, model prior, losses)"""
    if "num_joint_blocks" in g.files:
        from momentum_amd._abi import JointBlock

        blocks = []
        for i in range(int(g["num_joint_blocks"])):
            opt = lambda f: g[f"jb{i}_{f}"] if f"jb{i}_{f}" in g.files else None
            fw, la, lc = g[f"jb{i}_fw_loss"]
            blocks.append(JointBlock(int(g[f"jb{i}_type"]), g[f"jb{i}_parent"], opt("weight"), opt("global_"), opt("local_point"),
                                     opt("local_dir"), opt("plane_d"), float(fw), (float(la), float(lc))))  # fmt: skip
        return dict(joint_blocks=blocks)
    if "limits" not in g.files:
        return {}
    from momentum_amd._abi import ParameterLimit

    limits = []
    for row in g["limits"]:
        l = ParameterLimit(int(row[0]), int(row[1]), int(row[2]), float(row[3]))
        for k in range(4):
            l.v[k] = float(row[4 + k])
        limits.append(l)
    return dict(limits=limits, limit_function_weight=float(g["limit_function_weight"]), model_target=g["model_target"],
                model_weights=g["model_weights"], model_function_weight=float(g["model_function_weight"]),
                pos_loss=tuple(g["pos_loss"]), ori_loss=tuple(g["ori_loss"]))  # fmt: skip


def _cons(orc, g):
    return orc.Constraints(g["pos_parent"], g["pos_offset"], g["pos_target"], g["pos_weight"],
                           g["ori_parent"], g["ori_offset"], g["ori_target"], g["ori_weight"], **_extras(g))  # fmt: skip


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_fixture(orc, name):
    g = np.load(os.path.join(HERE, name))
    rig = CASES[name]()
    cons = _cons(orc, g)
    ref = orc.solve_batch(rig, cons, g["theta0"], GnOptions.make(**OPT), dtype="f64")
    assert np.abs(ref["theta"] - g["theta_final"]).max() <= 1e-10
    assert np.array_equal(ref["iterations"], g["iterations"])
    assert np.abs(ref["error_history"] - g["error_history"]).max() <= 1e-9 * max(1.0, np.abs(g["error_history"]).max())
    J0, r0, e0 = orc.eval_jacobian(rig, cons.instance(0), g["theta0"][0].astype(np.float64), dtype="f64")
    assert np.abs(J0 - g["jac0"]).max() <= 1e-6 * max(1.0, np.abs(J0).max())  # stored as fp32
    assert np.abs(r0 - g["res0"]).max() <= 1e-12 * max(1.0, np.abs(r0).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_path_matches_fixture(orc, name):
    import torch

    from momentum_amd import capi

    g = np.load(os.path.join(HERE, name))
    rig = CASES[name]()
    B = g["theta0"].shape[0]
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, g["pos_parent"], g["ori_parent"])
    dev = pb.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    Kp, Ko = len(g["pos_parent"]), len(g["ori_parent"])
    ex = _extras(g)
    if "model_target" in ex:
        ex["model_target"], ex["model_weights"] = t(ex["model_target"]), t(ex["model_weights"])
    if "joint_blocks" in ex:
        from momentum_amd._abi import JointBlock

        d = lambda a: None if a is None else t(a)
        ex["joint_blocks"] = [JointBlock(k.type, k.parent, d(k.weight), d(k.global_), d(k.local_point), d(k.local_dir), d(k.plane_d),
                                         k.function_weight, k.loss) for k in ex["joint_blocks"]]  # fmt: skip
    pb.set_constraints(t(g["pos_offset"]).reshape(B, Kp, 3), t(g["pos_target"]).reshape(B, Kp, 3), t(g["pos_weight"]).reshape(B, Kp),
                       t(g["ori_offset"]).reshape(B, Ko, 4), t(g["ori_target"]).reshape(B, Ko, 4), t(g["ori_weight"]).reshape(B, Ko), **ex)  # fmt: skip
    # world transforms at theta*
    st = pb.skeleton_state(t(g["theta_star"])).cpu().numpy()[0]
    assert np.abs(st - g["state_star0"]).max() <= 5e-6 * max(1.0, np.abs(g["state_star0"]).max())
    # dense Jacobian / residual at theta0 of instance 0
    jac, res, err = pb.eval_jacobian(t(g["theta0"]))
    assert np.abs(jac[0].cpu().numpy().T - g["jac0"]).max() <= 2e-5 * max(1.0, np.abs(g["jac0"]).max())
    assert np.abs(res[0].cpu().numpy() - g["res0"]).max() <= 2e-5 * max(1.0, np.abs(g["res0"]).max())
    assert abs(err[0].item() - float(g["err0"])) <= 2e-5 * max(1.0, float(g["err0"]))
    # ten Gauss-Newton iterations
    out = pb.solve(t(g["theta0"].copy()), GnOptions.make(**OPT), want_history=True)
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - g["theta_final"], axis=1) / np.linalg.norm(g["theta_final"], axis=1)
    tol = 1e-5 if name.startswith("cfg2") else 5e-5  # the chain fixture is ill-conditioned (tests/test_gpu_parity.py)
    if "joint_blocks" in name:
        # large-residual problem (final error 2..4.6: planes / aims that cannot all be met, a half-plane
        # block whose active set can flip), so the solution moves with dJ^T r and the fp32 storage of
        # the dense J bounds parity: measured against the oracle's own float instantiation, which is
        # 2e-5..9e-5 from its double one on this fixture
        r32 = orc.solve_batch(rig, _cons(orc, g), g["theta0"], GnOptions.make(**OPT), dtype="f32")["theta"]
        rel32 = np.linalg.norm(r32 - g["theta_final"], axis=1) / np.linalg.norm(g["theta_final"], axis=1)
        tol = max(1e-4, 3.0 * rel32.max())
    assert rel.max() <= tol, rel
    assert np.array_equal(out["iterations"].cpu().numpy(), g["iterations"])
    h = out["error_history"].cpu().numpy()
    assert np.abs(h - g["error_history"]).max() <= 1e-4 * max(1.0, np.abs(g["error_history"]).max())
