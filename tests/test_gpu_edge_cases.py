"""Edge cases of the C ABI on the GPU: empty constraint sets, empty enabled set, single joint,
single instance, parameter-space rows only -- compared with the oracle where there is something to
compare, otherwise checked for the reference's documented behaviour."""
import numpy as np
import pytest

from momentum_amd import capi  # noqa: E402  (default_route: which kernels the problems of a test run)

from momentum_amd import make_test_character
from momentum_amd._abi import GnOptions, ParameterLimit
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m gpu on the MI355X box)")
    return torch


def _upload(torch, pb, cons, B, **kw):
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
        cons.pos_function_weight, cons.ori_function_weight, **kw,
    )  # fmt: skip


@pytest.mark.parametrize("solver", ["fused", "v1"])
def test_no_enabled_parameters_keeps_theta(torch_cuda, orc, solver, monkeypatch):
    # SolverT::setEnabledParameters with an empty set: the compacted system has size 0, theta is unchanged

    torch = torch_cuda
    if solver == "v1":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_test_character(6)
    B = 3
    cons, th0, _ = make_problem(rig, [5, 2], [4], B, seed=1, theta0_scale=0.2)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, B)
    pb.set_enabled(np.zeros(rig.num_params, np.uint8))
    opt = GnOptions.make(min_iterations=3, max_iterations=3, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)
    assert np.array_equal(out["theta"].cpu().numpy(), th0)
    ref = orc.solve_batch(rig, cons, th0, opt, enabled=np.zeros(rig.num_params, np.uint8), dtype="f64")
    assert np.abs(out["error"].cpu().numpy() - ref["error"]).max() <= 1e-5 * max(1.0, np.abs(ref["error"]).max())
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])


def test_single_joint_single_instance(torch_cuda, orc):
    from momentum_amd.rigs import Rig

    torch = torch_cuda
    full = make_test_character(3)
    # a rig of one joint: root with its 6 rigid parameters + scale (first joint of the test character)
    J, P = 1, 7
    outer = np.arange(8, dtype=np.int32)
    rig = Rig(
        parent=np.array([-1], np.int32), pre_rotation=np.array([[0, 0, 0, 1]], np.float32),
        translation_offset=np.zeros((1, 3), np.float32), pt_outer=outer, pt_inner=np.arange(7, dtype=np.int32),
        pt_value=np.ones(7, np.float32), pt_offsets=np.zeros(7, np.float32), num_params=P,
        joint_names=["root"], param_names=[f"p{i}" for i in range(7)],
    )  # fmt: skip
    assert rig.num_joints == J
    cons, th0, ths = make_problem(rig, [0, 0], [0], 1, seed=5, perturb=0.3, random_offsets=True)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, 1, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, 1)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"]) / np.linalg.norm(ref["theta"])
    assert rel <= 1e-5, rel
    del full


@pytest.mark.parametrize("solver", ["fused", "v1"])
def test_parameter_rows_only_no_joint_constraints(torch_cuda, orc, solver, monkeypatch):
    # Kp = Ko = 0: only the limit and model-parameter blocks; the solution of the regularised
    # quadratic is the oracle's

    torch = torch_cuda
    if solver == "v1":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_test_character(5)
    P, B = rig.num_params, 2
    cons, _, _ = make_problem(rig, [], [], B, seed=2)
    rng = np.random.default_rng(8)
    limits = [ParameterLimit.minmax(1, -0.05, 0.05, 2.0), ParameterLimit.linear(2, 6, 0.5, 0.1), ParameterLimit.halfplane(3, 4, 0.6, 0.8, 0.3)]
    mt = rng.uniform(-0.3, 0.3, (B, P)).astype(np.float32)
    mw = rng.uniform(0.2, 1.5, (B, P)).astype(np.float32)
    full = orc.Constraints(
        cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
        limits=limits, limit_function_weight=1.0, model_target=mt, model_weights=mw, model_function_weight=1.0,
    )  # fmt: skip
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    dev = pb.device
    _upload(torch, pb, cons, B, limits=limits, model_target=torch.from_numpy(mt).to(dev), model_weights=torch.from_numpy(mw).to(dev))
    assert pb.M == len(limits) + P
    th0 = rng.uniform(-0.5, 0.5, (B, P)).astype(np.float32)
    opt = GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(dev), opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert np.all(rel <= 1e-5), rel
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())


def test_unsupported_limit_type_is_a_loud_error(torch_cuda):
    from momentum_amd._abi import ParameterLimit as PL

    torch = torch_cuda
    rig = make_test_character(4)
    cons, _, _ = make_problem(rig, [3], [], 1, seed=3)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, 1, cons.pos_parent, cons.ori_parent)
    bad = PL.minmax(0, -1, 1)
    bad.type = 5  # LimitType::Ellipsoid needs joint transforms: not implemented
    with pytest.raises(capi.MmxError) as ei:
        _upload(torch, pb, cons, 1, limits=[bad])
    assert "Ellipsoid" in str(ei.value)
    oob = PL.linear(0, rig.num_params, 1.0, 0.0)
    with pytest.raises(capi.MmxError):
        _upload(torch, pb, cons, 1, limits=[oob])


@pytest.mark.parametrize("count", [1, 15, 16, 17, 33, 100, 176, 191, 192, 200, 219])
def test_three_kernel_path_system_sizes(torch_cuda, orc, count, monkeypatch):
    """Explicit-Jacobian solver over the sizes of the dense system: partial 16-blocks of the blocked
    LDS Cholesky (panel rows / MFMA tiles beyond n), the 48-rows-per-wave limit of its panel (n <= 208),
    and the hand-over to the in-HBM factorisation once the factor no longer fits LDS."""
    from momentum_amd import make_humanoid72
    torch = torch_cuda
    monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_humanoid72(seed=12345, variant="p219", unit=0.01)
    P, J = rig.num_params, rig.num_joints
    assert P == 219
    allj = np.arange(J, dtype=np.int32)
    B = 2
    cons, th0, _ = make_problem(rig, allj, allj, B, seed=40 + count, perturb=0.25)
    rng = np.random.default_rng(count)
    en = np.zeros(P, np.uint8)
    en[rng.choice(P, size=count, replace=False)] = 1
    pb = capi.Problem(capi.RigHandle(rig, 0), B, allj, allj)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, J, 3)), t(cons.pos_target, (B, J, 3)), t(cons.pos_weight, (B, J)),
                       t(cons.ori_offset, (B, J, 4)), t(cons.ori_target, (B, J, 4)), t(cons.ori_weight, (B, J)))  # fmt: skip
    pb.set_enabled(en)
    for ls in (False, True):
        opt = GnOptions.make(min_iterations=5, max_iterations=5, regularization=0.05, do_line_search=ls)
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        ref = orc.solve_batch(rig, cons, th0, opt, enabled=en, dtype="f64")
        th = out["theta"].cpu().numpy()
        assert np.array_equal(th[:, en == 0], th0[:, en == 0])  # disabled parameters never move
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-12)
        # conditioning of the whole map: the oracle's own answer under an fp32-epsilon perturbation of theta0
        pert = orc.solve_batch(rig, cons, th0.astype(np.float64) + 1e-7 * rng.normal(size=th0.shape) * en, opt, enabled=en, dtype="f64")
        sens = np.linalg.norm(pert["theta"] - ref["theta"], axis=1) / np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-12)
        tol = np.maximum(1e-5, 3.0 * sens)
        assert (rel <= tol).all(), (count, ls, rel, tol)
        assert (out["status"].cpu().numpy() & 3 == 0).all()
        h = out["error_history"].cpu().numpy()
        assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())


@pytest.mark.parametrize("case", ["one_chunk", "many_units"])
def test_store_pattern_probe_writes_every_element(torch_cuda, case):
    """mmx_debug_store_pattern (the J-assembly kernel's stores without kinematics, bench.py's
    roofline.store_pattern_gbs): every element of the [B][P][M] block is written, for the one-chunk
    layout (<= 64 units, streaming stores) and the chunked one (plain stores, column-outer); problems
    with further row blocks are refused."""
    from momentum_amd import make_humanoid72
    from momentum_amd import humanoid72_landmark_joints

    torch = torch_cuda
    rig = make_humanoid72(unit=0.01)
    joints = humanoid72_landmark_joints(rig) if case == "one_chunk" else np.arange(rig.num_joints, dtype=np.int32)
    B = 5
    cons, _, _ = make_problem(rig, joints, joints, B, seed=1, perturb=0.1)
    pb = capi.Problem(capi.RigHandle(rig, 0), B, joints, joints)
    _upload(torch, pb, cons, B)
    M, P = pb.M, rig.num_params
    assert (M // 3 <= 64) == (case == "one_chunk")
    buf = torch.full((B, P, M), float("nan"), dtype=torch.float32, device=pb.device)
    ms = pb.store_pattern_kernel_ms(buf)
    assert ms > 0.0
    got = buf.cpu().numpy()
    assert not np.isnan(got).any()
    assert np.array_equal(got, np.broadcast_to(np.arange(B, dtype=np.float32)[:, None, None], got.shape))


def test_passive_limits_are_ignored_like_in_the_reference(torch_cuda, orc):
    """LimitType::MinMaxJointPassive gets neither an error term nor a row from LimitErrorFunctionT
    (limit_error_function.cpp:836-837,1051-1052): a limit list with passive entries gives the J / r / error
    and the solve of the same list without them (rows compacted)."""
    from momentum_amd._abi import ParameterLimit as PL
    from oracle import oracle as o

    torch = torch_cuda
    rig = make_test_character(8)
    B = 3
    cons, th0, _ = make_problem(rig, [7, 3], [6], B, seed=11, theta0_scale=0.3)
    active = [PL.minmax(1, -0.1, 0.1, 2.0), PL.minmax_joint(2, 4, -0.05, 0.05), PL.linear(3, 4, 0.5, 0.1)]
    passive = PL.minmax_joint_passive(1, 3, -0.01, 0.01, 5.0)
    mixed = [passive, active[0], active[1], passive, active[2]]
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, B, limits=mixed, limit_function_weight=0.7)
    assert pb.M == 3 * 2 + 9 + len(active)
    full = o.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target, cons.ori_weight,
                         limits=active, limit_function_weight=0.7)  # fmt: skip
    theta = np.random.default_rng(5).uniform(-0.4, 0.4, size=(B, rig.num_params)).astype(np.float32)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64), dtype="f64")
        assert np.abs(jac[b].cpu().numpy().T - J).max() <= 2e-5 * max(1.0, np.abs(J).max())
        assert np.abs(res[b].cpu().numpy() - r).max() <= 2e-5 * max(1.0, np.abs(r).max())
        assert abs(float(err[b]) - e) <= 2e-5 * max(1.0, e)
    opt = GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    assert np.abs(out["error"].cpu().numpy() - ref["error"]).max() <= 1e-4 * max(1.0, np.abs(ref["error"]).max())


def test_row_major_jacobian_is_the_transpose(torch_cuda):
    """MMX_LAYOUT_ROW_MAJOR: J[b][i * P + p], bit-identical to the transposed column-major result (sizes
    that are not multiples of the 32 x 32 transposition tile; limits add rows after the joint rows)."""
    from momentum_amd._abi import ParameterLimit as PL

    torch = torch_cuda
    rig = make_test_character(9)
    B = 5
    cons, th0, _ = make_problem(rig, [8, 3, 5], [6, 2], B, seed=21, theta0_scale=0.3)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, B, limits=[PL.minmax(1, -0.1, 0.1, 2.0)])
    theta = torch.from_numpy(th0.copy()).to(pb.device)
    jc, rc, ec = pb.eval_jacobian(theta)
    jr, rr, er = pb.eval_jacobian(theta, row_major=True)
    assert tuple(jr.shape) == (B, pb.M, pb.P)
    assert torch.equal(jr, jc.transpose(1, 2).contiguous()) and torch.equal(rc, rr) and torch.equal(ec, er)
    # host-buffer entry point with the row-major layout
    import ctypes as C

    jh = np.zeros((B, pb.M, pb.P), np.float32)
    capi._check(capi.lib().mmx_eval_jacobian_host(pb._h, th0.ctypes.data_as(C.c_void_p), jh.ctypes.data_as(C.c_void_p), None, None, 1))
    assert np.array_equal(jh, jr.cpu().numpy())


@pytest.mark.parametrize("solver", ["fused", "v1"])
def test_parameter_history_matches_the_iterates(torch_cuda, orc, solver, monkeypatch):
    """SolverT::setStoreHistory(true) (solver.cpp:53-72,101-110): iterationHistory_["parameters"].col(i) is the
    parameter vector after iteration i, untouched columns stay zero.  Checked against solves truncated at i + 1
    iterations (the solve is deterministic) and, for an element that converges early, against the zero rows."""

    torch = torch_cuda
    if solver == "v1":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig = make_test_character(8)
    B = 4
    cons, th0, _ = make_problem(rig, [7, 3], [6], B, seed=5, theta0_scale=0.2)
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    _upload(torch, pb, cons, B)
    opt = GnOptions.make(min_iterations=2, max_iterations=6, threshold=1e9, regularization=0.05)  # huge threshold: stops after min_iterations + 1
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True, want_parameter_history=True)
    hist = out["parameter_history"].cpu().numpy()
    iters = out["iterations"].cpu().numpy()
    th = out["theta"].cpu().numpy()
    assert hist.shape == (B, 6, rig.num_params) and iters.min() >= 3 and iters.min() < 6  # some element stops early
    for b in range(B):
        assert np.array_equal(hist[b, iters[b] - 1], th[b])
        assert np.all(hist[b, iters[b]:] == 0)
    for i in range(2):
        o2 = GnOptions.make(min_iterations=i + 1, max_iterations=i + 1, threshold=1e9, regularization=0.05)
        ti = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), o2)["theta"].cpu().numpy()
        assert np.array_equal(hist[:, i], ti)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    assert np.array_equal(iters, ref["iterations"])
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert rel.max() <= 5e-5


def test_jtj_history_is_the_damped_lower_triangle_of_every_iteration(torch_cuda, orc):
    """GaussNewtonSolverT's iterationHistory_["jtj"] (gauss_newton_solver.cpp:262-279): block i of the history is
    hessianApprox_ of iteration i -- the lower triangle of J^T J over the enabled parameters with the regularisation already on
    its diagonal (:249 adds it in place), the upper triangle never written.  Problem.jtj_history rebuilds it from the
    parameter history (mmx_eval_normal_equations at the parameters each iteration started from); checked against the
    oracle's compacted system of solves truncated at i + 1 iterations, enabled subset, an element that stops early."""
    torch = torch_cuda
    rig = make_test_character(8)
    B, lam = 4, 0.05
    cons, th0, _ = make_problem(rig, [7, 3], [6], B, seed=5, theta0_scale=0.2)
    enabled = np.ones(rig.num_params, np.uint8)
    enabled[[1, 4]] = 0
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    pb.set_enabled(enabled)
    _upload(torch, pb, cons, B)
    opt = GnOptions.make(min_iterations=2, max_iterations=6, threshold=1e9, regularization=lam)
    t0 = torch.from_numpy(th0.copy()).to(pb.device)
    out = pb.solve(t0.clone(), opt, want_history=True, want_parameter_history=True)
    iters = out["iterations"].cpu().numpy()
    H = pb.jtj_history(t0, out["parameter_history"], out["iterations"], lam).cpu().numpy()
    n = int(enabled.sum())
    assert H.shape == (B, 6, n, n) and iters.min() < 6
    for b in range(B):
        assert np.all(H[b, iters[b] :] == 0)
        sub = cons.subset(np.array([b]))
        for i in range(iters[b]):
            o2 = GnOptions.make(min_iterations=i + 1, max_iterations=i + 1, threshold=1e9, regularization=lam)
            ref = orc.solve(rig, sub, th0[b], o2, enabled=enabled, dtype="f64")["jtj"]
            want = np.tril(ref) + lam * np.eye(n)
            assert np.all(np.triu(H[b, i], 1) == 0)
            assert np.abs(H[b, i] - want).max() <= 2e-5 * np.abs(want).max(), (b, i)
