"""GPU parity of the LimitType::Ellipsoid rows (mmx_ellipsoid_limit) against the CPU oracle: J / r /
error next to the other limit types and joint constraints, and Gauss-Newton / line-search solves
(explicit-Jacobian kernels: jointBlocksKernel, stepUpdateKernel)."""
import numpy as np
import pytest

from momentum_amd import capi  # noqa: E402  (default_route: which kernels the problems of a test run)

from momentum_amd import _abi, humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import EllipsoidLimit, GnOptions, ParameterLimit
from tests.helpers import make_problem
from tests.test_gpu_joint_blocks import _device_block
from tests.test_oracle_joint_blocks import make_block

pytestmark = pytest.mark.gpu


def _setup(torch, orc, which, B, seed, with_blocks):

    if which == "chain8":
        rig, pp, op = make_test_character(8), [7, 3], [6]
        ells = [EllipsoidLimit.make(6, [0.1, 0.2, -0.1], 2, [0.0, 0.5, 0.0], [10.0, 20.0, 30.0], [0.6, 1.2, 0.8], 3.0),
                EllipsoidLimit.make(3, [0.0, 0.1, 0.0], 5, [0.1, 0.0, 0.2], [0.0, 0.0, 45.0], [1.0, 0.5, 0.7], 1.0)]  # fmt: skip
    else:
        rig = make_humanoid72(unit=0.01)
        lm = humanoid72_landmark_joints(rig)
        pp, op = lm, lm
        rng0 = np.random.default_rng(seed)
        ells = []
        for _ in range(5):
            parent = int(rng0.integers(1, rig.num_joints))
            chain = [parent]
            while rig.parent[chain[-1]] >= 0:
                chain.append(int(rig.parent[chain[-1]]))
            ep = int(rng0.choice(chain[1:])) if len(chain) > 1 and rng0.uniform() < 0.7 else int(rng0.integers(0, rig.num_joints))
            ells.append(EllipsoidLimit.make(parent, rng0.uniform(-0.05, 0.05, 3), ep, rng0.uniform(-0.1, 0.1, 3), rng0.uniform(-90, 90, 3),
                                            rng0.uniform(0.05, 0.3, 3), float(rng0.uniform(0.5, 4.0))))  # fmt: skip
    cons, th0, _ = make_problem(rig, pp, op, B, seed=seed, perturb=0.3)
    rng = np.random.default_rng(seed + 1)
    limits = [ParameterLimit.minmax(3, -0.05, 0.05, 1.0), ParameterLimit.linear(4, 5, 1.0, 0.0, weight=0.5)]
    blocks = [make_block(_abi.MMX_JC_HALF_PLANE, rng.choice(rig.num_joints, size=3), rng, weight=1.0, batch=B)] if with_blocks else []
    wl = 25.0  # ellipsoid rows carry kPositionWeight = 1e-4: a visible weight for the test
    full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset,
                           cons.ori_target, cons.ori_weight, limits=limits, limit_function_weight=wl, joint_blocks=blocks,
                           ellipsoid_limits=ells)  # fmt: skip
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
                       limits=limits, limit_function_weight=wl, joint_blocks=[_device_block(torch, k, pb.device) for k in blocks],
                       ellipsoid_limits=ells)  # fmt: skip
    assert pb.M == full.rows
    return rig, pb, full, th0


@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
@pytest.mark.parametrize("with_blocks", [False, True])
def test_ellipsoid_rows_match_oracle(orc, which, with_blocks):
    import torch

    B = 3
    rig, pb, full, th0 = _setup(torch, orc, which, B, 17, with_blocks)
    rng = np.random.default_rng(5)
    theta = rng.uniform(-0.4, 0.4, size=(B, rig.num_params)).astype(np.float32)
    en = np.ones(rig.num_params, np.uint8)
    en[[2, 6]] = 0
    for enabled in (None, en):
        if enabled is not None:
            pb.set_enabled(enabled)
        jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
        jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
        for b in range(B):
            J, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), enabled=enabled, dtype="f64")
            assert np.abs(jac[b].T - J).max() <= 3e-5 * max(1.0, np.abs(J).max())
            assert np.abs(res[b] - r).max() <= 3e-5 * max(1.0, np.abs(r).max())
            assert abs(err[b] - e) <= 3e-5 * max(1.0, e)
            nE = 3 * len(full.ellipsoid_limits)
            rowsE = slice(full.rows - len(full.limits) - nE, full.rows - len(full.limits))
            assert np.abs(J[rowsE]).max() > 0  # the rows under test are not trivially zero


@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
@pytest.mark.parametrize("line_search", [False, True])
def test_solve_with_ellipsoid_limits_matches_oracle(orc, which, line_search):
    import torch

    from tests.test_gpu_parity import _sensitivity

    B = 4
    rig, pb, full, th0 = _setup(torch, orc, which, B, 23, False)
    opt = GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05, do_line_search=line_search)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    tol = np.maximum(3e-5, 3.0 * _sensitivity(orc, rig, full, th0, opt, ref))
    assert (rel <= tol).all(), (rel, tol)
    h = out["error_history"].cpu().numpy()
    assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())
    assert (out["status"].cpu().numpy() & 3 == 0).all()


@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
def test_fused_solve_carries_blocks_and_ellipsoids(orc, which, monkeypatch):
    """The one-launch solve with a half-plane block, ellipsoid limits and parameter limits next to the position /
    orientation constraints (the marker tracker's shape, marker_tracker.cpp:916-960): its normal equations
    (parity hook of the fused kernel -- only answered when the problem takes the fused path) against the oracle's
    J^T J / J^T r in double, then the solve under three step rules against the oracle and against the
    explicit-Jacobian kernels (MMX_ROUTE_EXPLICIT_JACOBIAN)."""
    import torch

    from momentum_amd._abi import MMX_STEP_LM_SCHEDULE
    from tests.test_gpu_parity import _sensitivity

    B = 4
    rig, pb, full, th0 = _setup(torch, orc, which, B, 29, True)
    rng = np.random.default_rng(6)
    theta = rng.uniform(-0.4, 0.4, size=(B, rig.num_params)).astype(np.float32)
    lst, jtj, jtr = pb.fused_normal_equations(torch.from_numpy(theta).to(pb.device))
    jtj, jtr = jtj.cpu().numpy(), jtr.cpu().numpy()
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), dtype="f64")
        Je = J[:, lst]
        H, g = Je.T @ Je, Je.T @ r
        assert np.abs(jtj[b] - H).max() <= 5e-5 * max(1.0, np.abs(H).max()), np.abs(jtj[b] - H).max() / np.abs(H).max()
        assert np.abs(jtr[b] - g).max() <= 5e-5 * max(1.0, np.abs(g).max())
        other = np.setdiff1d(np.arange(rig.num_params), lst)
        if other.size:  # what left the solve list is structurally zero
            assert np.abs(J[: full.rows - len(full.limits), other]).max() == 0.0
    for opt in (
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05),
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05, do_line_search=2),
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE),
    ):
        ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
        tol = np.maximum(3e-5, 3.0 * _sensitivity(orc, rig, full, th0, opt, ref))
        outs = []
        for general in ("fused", "explicit_jacobian"):
            pb.set_route(general)
            out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
            th = out["theta"].cpu().numpy()
            rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
            assert (rel <= tol).all(), (general, rel, tol)
            assert (out["status"].cpu().numpy() & 3 == 0).all()
            h = out["error_history"].cpu().numpy()
            assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())
            assert pb.last_route() == general
            outs.append(th)
        pb.set_route("auto")
        assert not np.array_equal(outs[0], outs[1])  # two different routes really ran (fp32 rounding differs)


@pytest.mark.parametrize("route", ["tree", "dense"])
def test_wide_solve_carries_blocks_and_ellipsoids(orc, route, monkeypatch):
    """The 300-joint rig (wide path) with a half-plane block, an aim block, a fixed-axis block, ellipsoid limits and
    parameter limits next to 160 position / orientation constraints: the tree kernels keep those rows as a dense block
    J_g (tree normal equations: g, rank-k update of the tiles; tree refinement: r_g - J_g d), the dense route assembles
    them into J (jointBlocksKernel)."""
    import ctypes as C

    import torch

    from momentum_amd import make_rig300
    from tests.test_gpu_parity import _sensitivity

    if route == "dense":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_rig300(seed=12345, unit=0.01)
    rng = np.random.default_rng(83)
    J, B = rig.num_joints, 3
    pp = rng.choice(J, size=120, replace=False)
    op = rng.choice(J, size=40, replace=False)
    cons, th0, _ = make_problem(rig, pp, op, B, seed=97, perturb=0.2)
    blocks = [
        make_block(_abi.MMX_JC_HALF_PLANE, rng.choice(J, size=6), rng, weight=1.0, batch=B),
        make_block(_abi.MMX_JC_AIM_DIST, rng.choice(J, size=1), rng, weight=0.5, batch=B),
        make_block(_abi.MMX_JC_FIXED_AXIS_DIFF, rng.choice(J, size=1), rng, weight=0.5, batch=B, function_weight=0.7),
    ]  # (15 rows with the ellipsoid limit: what fits next to this rig's 125 KB of tree tables in LDS; more rows take the dense route)
    ells = []
    for _ in range(1):
        parent = int(rng.integers(1, J))
        chain = [parent]
        while rig.parent[chain[-1]] >= 0:
            chain.append(int(rig.parent[chain[-1]]))
        ep = int(rng.choice(chain[1:])) if len(chain) > 1 else 0
        ells.append(EllipsoidLimit.make(parent, rng.uniform(-0.05, 0.05, 3), ep, rng.uniform(-0.1, 0.1, 3), rng.uniform(-90, 90, 3),
                                        rng.uniform(0.05, 0.3, 3), float(rng.uniform(0.5, 4.0))))  # fmt: skip
    limits = [ParameterLimit.minmax(3, -0.05, 0.05, 1.0), ParameterLimit.linear(4, 5, 1.0, 0.0, weight=0.5)]
    wl = 25.0
    full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset,
                           cons.ori_target, cons.ori_weight, limits=limits, limit_function_weight=wl, joint_blocks=blocks,
                           ellipsoid_limits=ells)  # fmt: skip
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
                       limits=limits, limit_function_weight=wl, joint_blocks=[_device_block(torch, k, pb.device) for k in blocks],
                       ellipsoid_limits=ells)  # fmt: skip
    assert pb.M == full.rows
    if route == "tree":  # the tree normal equations with J_g against the oracle's J^T J / J^T r (parity hook)
        buf, nn = np.zeros(rig.num_params, np.int32), C.c_int32(0)
        capi._check(capi.lib().mmx_debug_fused_normal_equations(pb._h, None, None, None, capi.as_ptr(buf, C.c_int32), C.byref(nn), None))
        lst = np.sort(buf[: nn.value])  # (elimination order -> parameter order, which tree_normal_equations reports in)
        en = np.zeros(rig.num_params, np.uint8)
        en[lst] = 1
        pb.set_enabled(en)  # only structurally non-zero columns: the enabled system IS the solve-list system
        theta = rng.uniform(-0.2, 0.2, size=(B, rig.num_params)).astype(np.float32)
        Ht, gt = pb.tree_normal_equations(torch.from_numpy(theta).to(pb.device))
        Ht, gt = Ht.cpu().numpy(), gt.cpu().numpy()
        for b in range(B):
            Jm, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), enabled=en, dtype="f64")
            Je = Jm[:, lst]
            H, g = Je.T @ Je, Je.T @ r
            assert np.abs(np.tril(Ht[b]) - np.tril(H)).max() <= 5e-5 * max(1.0, np.abs(H).max())
            assert np.abs(gt[b] - g).max() <= 5e-5 * max(1.0, np.abs(g).max())
        pb.set_enabled(np.ones(rig.num_params, np.uint8))
    for opt in (
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05),
        GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05, do_line_search=2),
    ):
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
        rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        tol = np.maximum(3e-5, 3.0 * _sensitivity(orc, rig, full, th0, opt, ref))
        assert (rel <= tol).all(), (route, rel, tol)
        assert (out["status"].cpu().numpy() & 3 == 0).all()
        h = out["error_history"].cpu().numpy()
        assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())
