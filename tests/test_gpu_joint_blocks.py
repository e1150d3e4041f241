"""GPU parity of the further JointErrorFunctionT specialisations (Plane / HalfPlane / AimDist / AimDir /
FixedAxisDiff / Cos / Angle / Normal) against the CPU oracle, through the C ABI.
While they fit the fused solve carries these rows as a small dense block (kGen instantiations of fusedSolveKernel);
MMX_FUSED_GENERAL=0 or larger problems take the explicit-Jacobian kernels (J assembly -> J^T J -> Cholesky step)."""
import numpy as np
import pytest

from momentum_amd import _abi, humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import GnOptions, JointBlock
from tests.helpers import make_problem
from tests.test_oracle_joint_blocks import TYPES, make_block

pytestmark = pytest.mark.gpu
UNIT = 0.01


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m gpu on the MI355X box)")
    return torch


def _device_block(torch, blk: JointBlock, dev) -> JointBlock:
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    return JointBlock(blk.type, blk.parent, t(blk.weight), t(blk.global_), t(blk.local_point), t(blk.local_dir), t(blk.plane_d),
                      blk.function_weight, blk.loss)  # fmt: skip


def _problem(torch, orc, rig, pp, op, B, seed, blocks, device_payload=True, pos_w=1.0, ori_w=1.0):
    from momentum_amd import capi

    cons, th0, _ = make_problem(rig, pp, op, B, seed=seed, perturb=0.3)
    full = orc.Constraints(
        cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
        pos_function_weight=pos_w, ori_function_weight=ori_w, joint_blocks=blocks,
    )  # fmt: skip
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    dev = pb.device
    if device_payload:
        t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
        gb = [_device_block(torch, blk, dev) for blk in blocks]
    else:
        t = lambda a, shp: np.ascontiguousarray(a, np.float32).reshape(shp)
        gb = blocks
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
        pos_w, ori_w, joint_blocks=gb,
    )  # fmt: skip
    assert pb.M == full.rows
    return rh, pb, full, th0


def _compare_jacobian(torch, orc, rig, pb, full, theta, enabled=None, tol=3e-5):
    B = theta.shape[0]
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), enabled=enabled, dtype="f64")
        Jg = jac[b].T
        assert Jg.shape == J.shape
        scale = max(1.0, np.abs(J).max())
        assert np.abs(Jg - J).max() <= tol * scale, (b, np.abs(Jg - J).max(), scale)
        assert np.abs(Jg[:, np.abs(J).max(axis=0) == 0]).max(initial=0.0) == 0  # structurally zero columns are exact zeros
        assert np.abs(res[b] - r).max() <= tol * max(1.0, np.abs(r).max())
        assert abs(err[b] - e) <= tol * max(1.0, e)


@pytest.mark.parametrize("name", list(TYPES))
def test_block_rows_of_jacobian_match_oracle(torch_cuda, orc, name):
    """One block type at a time on the 72-joint rig, next to position + orientation constraints."""
    torch = torch_cuda
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    B = 3
    rng = np.random.default_rng(100 + TYPES[name])
    parents = rng.choice(rig.num_joints, size=11, replace=True)
    blk = make_block(TYPES[name], parents, rng, weight=1.3, batch=B, function_weight=0.8)
    blk.weight[:, 2] = 0.0
    rh, pb, full, th0 = _problem(torch, orc, rig, lm[:5], lm[5:8], B, 77, [blk])
    theta = rng.uniform(-0.4, 0.4, size=(B, rig.num_params)).astype(np.float32)
    en = np.ones(rig.num_params, np.uint8)
    en[[1, 7, 40]] = 0
    for enabled in (None, en):
        if enabled is not None:
            pb.set_enabled(enabled)
        _compare_jacobian(torch, orc, rig, pb, full, theta, enabled)


@pytest.mark.parametrize("which", ["chain8", "humanoid72", "blocks_only"])
def test_mixed_blocks_match_oracle(torch_cuda, orc, which):
    """All eight types at once, a robust loss on one block, a disabled block, host payload ingest."""
    torch = torch_cuda
    if which == "humanoid72":
        rig = make_humanoid72(unit=UNIT)
        lm = humanoid72_landmark_joints(rig)
        pp, op, B = lm, lm, 4
    elif which == "chain8":
        rig, pp, op, B = make_test_character(8), [7, 3], [6], 5
    else:
        rig, pp, op, B = make_test_character(8), [], [], 5
    rng = np.random.default_rng(5)
    J = rig.num_joints
    blocks = []
    for i, ty in enumerate(TYPES.values()):
        cnt = int(rng.integers(1, 6))
        loss = (0.0, 0.7) if i == 3 else ((1.0, 0.5) if i == 6 else (2.0, 1.0))
        fw = 0.0 if i == 5 else float(rng.uniform(0.5, 1.5))
        blocks.append(make_block(ty, rng.choice(J, size=cnt), rng, weight=1.0, batch=B, function_weight=fw, loss=loss))
    rh, pb, full, th0 = _problem(torch, orc, rig, pp, op, B, 31, blocks, device_payload=(which != "chain8"))
    theta = rng.uniform(-0.5, 0.5, size=(B, rig.num_params)).astype(np.float32)
    _compare_jacobian(torch, orc, rig, pb, full, theta)


@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
@pytest.mark.parametrize("mode", ["gn", "line_search"])
def test_solve_with_blocks_matches_oracle(torch_cuda, orc, which, mode):
    """SolverT::solve with GaussNewtonSolver on position + orientation + plane / aim / fixed-axis /
    normal constraints: theta within the parity bar of the oracle's double solve."""
    from tests.test_gpu_parity import _sensitivity

    torch = torch_cuda
    if which == "chain8":
        rig, pp, op, B = make_test_character(8), [7, 3], [6], 4
    else:
        rig = make_humanoid72(unit=UNIT)
        lm = humanoid72_landmark_joints(rig)
        pp, op, B = lm, lm, 4
    rng = np.random.default_rng(9)
    J = rig.num_joints
    blocks = [
        make_block(_abi.MMX_JC_HALF_PLANE, rng.choice(J, size=4), rng, weight=1.0, batch=B),
        make_block(_abi.MMX_JC_AIM_DIST, rng.choice(J, size=2), rng, weight=0.5, batch=B),
        make_block(_abi.MMX_JC_FIXED_AXIS_DIFF, rng.choice(J, size=3), rng, weight=0.5, batch=B, function_weight=0.6),
        make_block(_abi.MMX_JC_NORMAL, rng.choice(J, size=3), rng, weight=1.0, batch=B, loss=(0.0, 1.0)),
    ]
    rh, pb, full, th0 = _problem(torch, orc, rig, pp, op, B, 12345, blocks)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, do_line_search=(mode == "line_search"))
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    th = out["theta"].cpu().numpy()
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    assert (out["iterations"].cpu().numpy() == ref["iterations"]).all()
    assert (out["status"].cpu().numpy() & 3 == 0).all()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    sens = _sensitivity(orc, rig, full, th0, opt, ref)
    # explicit-J path on a large-residual problem (unsatisfiable planes / aims): fp32 storage of J bounds
    # parity at a few 1e-5 (the oracle's own float instantiation is at 2e-5..1e-4 here)
    tol = np.maximum(5e-5, 3.0 * sens)
    assert (rel <= tol).all(), (rel, tol)
    hist = out["error_history"].cpu().numpy()
    assert np.abs(hist - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())


def test_failed_set_constraints_does_not_poison_the_next_call(torch_cuda, orc):
    """A set_constraints call that fails its validation (here: an ellipsoid limit with a joint index out of
    range, next to NEW joint blocks) must leave the problem usable: nothing is modified before everything
    has been validated, so the retry with valid data rebuilds the block tables and J / r match the oracle."""
    from momentum_amd import capi
    from momentum_amd._abi import EllipsoidLimit

    torch = torch_cuda
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    B = 2
    rng = np.random.default_rng(9)
    blk = make_block(TYPES["plane"] if "plane" in TYPES else list(TYPES.values())[0], rng.choice(rig.num_joints, size=7), rng, weight=1.1, batch=B, function_weight=0.9)
    cons, th0, _ = make_problem(rig, lm[:4], lm[4:6], B, seed=5, perturb=0.3)
    full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
                           joint_blocks=[blk])  # fmt: skip
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    args = (t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
            t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)), 1.0, 1.0)  # fmt: skip
    gb = [_device_block(torch, blk, dev)]
    bad = EllipsoidLimit.make(rig.num_joints + 3, (0.0, 0.1, 0.0), 0, (0, 0, 0), (0, 0, 0), (1, 1, 1))
    rows_before = pb.M
    with pytest.raises(capi.MmxError):
        pb.set_constraints(*args, joint_blocks=gb, ellipsoid_limits=[bad])
    assert int(capi.lib().mmx_problem_num_rows(pb._h)) == rows_before  # nothing was modified
    pb.set_constraints(*args, joint_blocks=gb)
    assert pb.M == full.rows
    theta = rng.uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)
    _compare_jacobian(torch, orc, rig, pb, full, theta)
