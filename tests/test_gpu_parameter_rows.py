"""GPU parity of the parameter-space error functions (LimitErrorFunction on model parameters,
ModelParametersErrorFunction) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

from momentum_amd import capi  # noqa: E402  (default_route: which kernels the problems of a test run)

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import GnOptions, ParameterLimit
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu
UNIT = 0.01
FLT_MAX = float(np.finfo(np.float32).max)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m gpu on the MI355X box)")
    return torch


def _limits(P, rng, count, rig=None):
    out = []
    for k in range(count):
        kind = k % (6 if rig is not None else 4)
        a, b = rng.choice(P, size=2, replace=False)
        if kind == 0:
            out.append(ParameterLimit.minmax(a, -0.08, 0.12, rng.uniform(0.5, 2.0)))
        elif kind == 1:
            out.append(ParameterLimit.linear(a, b, rng.uniform(-1, 1), rng.uniform(-0.2, 0.2), weight=rng.uniform(0.5, 2.0)))
        elif kind == 2:
            out.append(ParameterLimit.linear(a, b, 1.0, -0.1, -0.1, FLT_MAX, weight=0.5))  # piecewise: applies above -0.1 only
        elif kind == 3:
            n = rng.normal(size=2)
            n /= np.linalg.norm(n)
            out.append(ParameterLimit.halfplane(a, b, n[0], n[1], 0.1, rng.uniform(0.5, 2.0)))
        else:
            # joint-parameter limits on rotation rows that are driven by at least one model parameter
            rows = [r for r in range(7 * rig.num_joints) if r % 7 in (3, 4, 5) and rig.pt_outer[r + 1] > rig.pt_outer[r]]
            r0, r1 = rng.choice(rows, size=2, replace=False)
            if kind == 4:
                out.append(ParameterLimit.minmax_joint(r0 // 7, r0 % 7, -0.05, 0.08, rng.uniform(0.5, 2.0)))
            else:
                out.append(ParameterLimit.linear_joint(r0 // 7, r0 % 7, r1 // 7, r1 % 7, rng.uniform(-1, 1), 0.05, weight=rng.uniform(0.5, 2.0)))
    return out


def _problem(torch, orc, rig, pp, op, B, seed, with_limits=True, with_model=True):

    cons, th0, ths = make_problem(rig, pp, op, B, seed=seed, perturb=0.3)
    P = rig.num_params
    rng = np.random.default_rng(seed + 7)
    limits = _limits(P, rng, 19, rig) if with_limits else []
    mt = rng.uniform(-0.2, 0.2, size=(B, P)).astype(np.float32) if with_model else None
    mw = rng.uniform(-0.3, 1.5, size=(B, P)).astype(np.float32) if with_model else None  # some weights <= 0: rows dropped
    full = orc.Constraints(
        cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
        limits=limits, limit_function_weight=0.6, model_target=mt, model_weights=mw, model_function_weight=1.4,
    )  # fmt: skip
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
        1.0, 1.0, limits=limits, limit_function_weight=0.6,
        model_target=None if mt is None else t(mt, (B, P)), model_weights=None if mw is None else t(mw, (B, P)), model_function_weight=1.4,
    )  # fmt: skip
    assert pb.M == full.rows
    return rh, pb, full, th0


@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
@pytest.mark.parametrize("blocks", ["limits", "model", "both"])
def test_parameter_rows_of_jacobian_match_oracle(torch_cuda, orc, which, blocks):
    torch = torch_cuda
    if which == "chain8":
        rig, pp, op, B = make_test_character(8), [7, 3], [6], 4
    else:
        rig = make_humanoid72(unit=UNIT)
        pp = op = humanoid72_landmark_joints(rig)
        B = 3
    rh, pb, full, th0 = _problem(torch, orc, rig, pp, op, B, 300, blocks != "model", blocks != "limits")
    rng = np.random.default_rng(3)
    theta = rng.uniform(-0.4, 0.4, size=(B, rig.num_params)).astype(np.float32)
    en = np.ones(rig.num_params, np.uint8)
    en[[2, 5]] = 0
    for enabled in (None, en):
        if enabled is not None:
            pb.set_enabled(enabled)
        jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
        jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
        for b in range(B):
            J, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), enabled=enabled, dtype="f64")
            Jg = jac[b].T
            assert Jg.shape == J.shape
            scale = max(1.0, np.abs(J).max())
            assert np.abs(Jg - J).max() <= 2e-5 * scale
            assert np.abs(Jg[np.abs(J) == 0]).max() == 0  # structural zeros (and unused rows) are exact zeros
            assert np.abs(res[b] - r).max() <= 2e-5 * max(1.0, np.abs(r).max())
            assert abs(err[b] - e) <= 2e-5 * max(1.0, e)


@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
@pytest.mark.parametrize("mode", ["gn", "line_search", "lm_schedule", "three_kernel"])
def test_solve_with_limits_and_model_prior_matches_oracle(torch_cuda, orc, which, mode, monkeypatch):
    """The fused kernel folds the rows into g / H / the refinement / the line-search error on the fly;
    MMX_ROUTE_EXPLICIT_JACOBIAN (three_kernel) goes through the dense J instead -- both must match the oracle."""
    from momentum_amd._abi import MMX_STEP_LM_SCHEDULE
    from tests.test_gpu_parity import _sensitivity

    torch = torch_cuda
    if which == "chain8":
        rig, pp, op, B = make_test_character(8), [7, 3], [6], 4
    else:
        rig = make_humanoid72(unit=UNIT)
        pp = op = humanoid72_landmark_joints(rig)
        B = 4
    if mode == "three_kernel":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rh, pb, full, th0 = _problem(torch, orc, rig, pp, op, B, 12345)
    kw = dict(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
    if mode == "line_search":
        kw["do_line_search"] = True
    elif mode == "lm_schedule":
        kw["step_rule"] = MMX_STEP_LM_SCHEDULE
    opt = GnOptions.make(**kw)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    th = out["theta"].cpu().numpy()
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    # the model-parameter prior regularises every parameter, so even the chain fixture is well
    # conditioned: strict 1e-5, except under the LM schedule (lambda shrinks; bounded by the measured
    # sensitivity of the oracle's own double solve like in test_gpu_parity)
    tol = np.full(B, 1e-5)
    if mode == "lm_schedule":
        tol = np.maximum(tol, 3.0 * _sensitivity(orc, rig, full, th0, opt, ref))
    assert np.all(rel <= tol), (rel, tol)
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
    assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())


@pytest.mark.parametrize("route", ["tree", "dense"])
def test_wide_solve_with_limits_and_the_model_prior(torch_cuda, orc, route, monkeypatch):
    """The 300-joint rig (wide path) with parameter limits of every type and the model-parameter prior next to its
    position / orientation constraints: the tree kernels evaluate these rows on the fly from theta (normal equations:
    g, diagonal and shared off-diagonal entries of H; refinement residual), the dense route assembles them into J."""
    from momentum_amd import make_rig300

    torch = torch_cuda
    if route == "dense":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_rig300(seed=12345, unit=UNIT)
    rng = np.random.default_rng(79)
    pp = rng.choice(rig.num_joints, size=120, replace=False)
    op = rng.choice(rig.num_joints, size=40, replace=False)
    B = 3
    rh, pb, full, th0 = _problem(torch, orc, rig, pp, op, B, 91)
    if route == "tree":  # the tree-moment normal equations with the parameter-space rows against the oracle's J^T J / J^T r
        import ctypes as C


        buf, nn = np.zeros(rig.num_params, np.int32), C.c_int32(0)
        capi._check(capi.lib().mmx_debug_fused_normal_equations(pb._h, None, None, None, capi.as_ptr(buf, C.c_int32), C.byref(nn), None))
        lst = np.sort(buf[: nn.value])  # (elimination order -> parameter order, which tree_normal_equations reports in)
        assert nn.value == rig.num_params  # the model prior keeps every parameter in the solve list
        theta = rng.uniform(-0.2, 0.2, size=(B, rig.num_params)).astype(np.float32)
        Ht, gt = pb.tree_normal_equations(torch.from_numpy(theta).to(pb.device))
        Ht, gt = Ht.cpu().numpy(), gt.cpu().numpy()
        for b in range(B):
            Jm, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), dtype="f64")
            Je = Jm[:, lst]
            H, g = Je.T @ Je, Je.T @ r
            assert np.abs(np.tril(Ht[b]) - np.tril(H)).max() <= 5e-5 * max(1.0, np.abs(H).max())
            assert np.abs(gt[b] - g).max() <= 5e-5 * max(1.0, np.abs(g).max())
    from tests.test_gpu_parity import _sensitivity

    for opt in (
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05),
        GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05, do_line_search=2),
    ):
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
        th = out["theta"].cpu().numpy()
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        tol = np.maximum(1e-5, 3.0 * _sensitivity(orc, rig, full, th0, opt, ref))
        assert np.all(rel <= tol), (route, rel, tol)
        assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
        h = out["error_history"].cpu().numpy()
        assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())
