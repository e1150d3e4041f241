"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Tolerances: bit-exact for integer outputs; floating point within north_star's 1e-5 relative on pose
parameters (stated at each assert)."""
import numpy as np
import pytest

from momentum_amd import capi  # noqa: E402  (default_route: which kernels the problems of a test run)

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import GnOptions
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu

UNIT = 0.01  # humanoid offsets U[2,30] cm expressed in metres


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("these tests need a GPU (run with -m gpu on the MI355X box)")
    return torch


def _gpu_problem(torch, rig, cons, B):

    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
        cons.pos_function_weight, cons.ori_function_weight,
    )  # fmt: skip
    return rh, pb


CASES = {
    # name: (rig factory, pos parents, ori parents, batch)
    "chain24_cfg1": (lambda: make_test_character(24), [23, 12, 5], [], 3),
    "chain8_mixed": (lambda: make_test_character(8), [7, 3, 1], [6, 2], 5),
    "humanoid72_cfg2": (lambda: make_humanoid72(unit=UNIT), "lm", "lm", 6),
    "humanoid72_ori_only": (lambda: make_humanoid72(unit=UNIT), [], [6, 26, 51, 12], 2),
    "humanoid72_many_units": (lambda: make_humanoid72(unit=UNIT), list(range(0, 72, 2)), list(range(1, 72, 3)), 2),
}


def _case(name):
    mk, pp, op, B = CASES[name]
    rig = mk()
    if pp == "lm":
        pp = humanoid72_landmark_joints(rig)
    if op == "lm":
        op = humanoid72_landmark_joints(rig)
    return rig, pp, op, B


def test_skeleton_state_matches_oracle(torch_cuda, orc):
    torch = torch_cuda
    for name in ("chain24_cfg1", "humanoid72_cfg2"):
        rig, pp, op, B = _case(name)
        cons, th0, ths = make_problem(rig, pp, op, B, seed=21, perturb=0.5)
        rh, pb = _gpu_problem(torch, rig, cons, B)
        st = pb.skeleton_state(torch.from_numpy(ths).to(pb.device)).cpu().numpy()
        for b in range(B):
            ref = orc.skeleton_state(rig, ths[b].astype(np.float64), "f64")["world"]
            scale = max(1.0, np.abs(ref[:, :3]).max())
            assert np.abs(st[b, :, :3] - ref[:, :3]).max() <= 5e-6 * scale  # fp32 FK chain of depth <= 24
            assert np.abs(st[b, :, 3:] - ref[:, 3:]).max() <= 5e-6
    # the reference's FK golden value (forward_kinematics_test.cpp:80-86) through the GPU path
    rig = make_test_character(24)
    cons, _, _ = make_problem(rig, [2], [], 1)
    rh, pb = _gpu_problem(torch, rig, cons, 1)
    th = np.zeros((1, rig.num_params), np.float32)
    th[0, :10] = [1, 1, 1, np.pi, 0, -np.pi, 0.1, np.pi, np.pi, -np.pi]
    w = pb.skeleton_state(torch.from_numpy(th).to(pb.device)).cpu().numpy()[0, 2].astype(np.float64)
    from tests.helpers import quat_rot

    p = w[:3] + quat_rot(w[3:7], w[7] * np.ones(3))
    assert np.abs(p - [-1.14354682, 3.14354706, -0.0717732906]).max() <= 2e-6


@pytest.mark.parametrize("name", list(CASES))
def test_jacobian_residual_error_match_oracle(torch_cuda, orc, name):
    torch = torch_cuda
    rig, pp, op, B = _case(name)
    cons, th0, ths = make_problem(rig, pp, op, B, seed=100, perturb=0.4, random_offsets=True, weights="random")
    cons.pos_function_weight, cons.ori_function_weight = 0.8, 1.25
    if cons.Kp:
        cons.pos_weight[0, 0] = 0.0  # a zero-weight constraint keeps zero rows
    rh, pb = _gpu_problem(torch, rig, cons, B)
    rng = np.random.default_rng(9)
    theta = rng.uniform(-0.4, 0.4, size=(B, rig.num_params)).astype(np.float32)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, cons.instance(b), theta[b].astype(np.float64), dtype="f64")
        Jg = jac[b].T  # [M,P]
        scale = max(1.0, np.abs(J).max())
        assert np.abs(Jg - J).max() <= 2e-5 * scale, (name, b)
        assert np.array_equal(Jg == 0, np.abs(J) == 0) or np.abs(Jg[np.abs(J) == 0]).max() == 0  # structural zeros are exact zeros
        assert np.abs(res[b] - r).max() <= 2e-5 * max(1.0, np.abs(r).max())
        assert abs(err[b] - e) <= 2e-5 * max(1.0, e)


def test_jacobian_with_disabled_parameters(torch_cuda, orc):
    torch = torch_cuda
    rig, pp, op, B = _case("humanoid72_cfg2")
    cons, th0, ths = make_problem(rig, pp, op, B, seed=5, perturb=0.3)
    rh, pb = _gpu_problem(torch, rig, cons, B)
    rng = np.random.default_rng(1)
    en = (rng.uniform(size=rig.num_params) < 0.7).astype(np.uint8)
    pb.set_enabled(en)
    theta = ths
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    jac = jac.cpu().numpy()
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, cons.instance(b), theta[b].astype(np.float64), enabled=en, dtype="f64")
        assert np.abs(jac[b].T - J).max() <= 2e-5 * max(1.0, np.abs(J).max())
        assert np.all(jac[b].T[:, en == 0] == 0)
    # normal equations of the compacted system (validateIdentical compares JtJ / Jtr, error_function_helpers.cpp:347-368)
    jtj, jtr, _ = pb.normal_equations(torch.from_numpy(theta).to(pb.device))
    jtj, jtr = jtj.cpu().numpy(), jtr.cpu().numpy()
    keep = np.flatnonzero(en)
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, cons.instance(b), theta[b].astype(np.float64), enabled=en, dtype="f64")
        H = J[:, keep].T @ J[:, keep]
        g = J[:, keep].T @ r
        assert np.abs(jtj[b] - H).max() <= 5e-5 * max(1.0, np.abs(H).max())
        assert np.abs(jtr[b] - g).max() <= 5e-5 * max(1.0, np.abs(g).max())
        assert np.array_equal(jtj[b], jtj[b].T)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("subset", [False, True])
def test_fused_kernel_normal_equations_match_oracle(torch_cuda, orc, name, subset):
    """The fused solve kernel never forms J: its J^T J (from per-joint subtree moments) and J^T r
    (adjoint pass) must equal the oracle's J^T J / J^T r on the kernel's solve list, and every
    parameter it drops from the dense system must have a structurally zero Jacobian column."""
    torch = torch_cuda
    rig, pp, op, B = _case(name)
    cons, th0, ths = make_problem(rig, pp, op, B, seed=100, perturb=0.4, random_offsets=True, weights="random")
    cons.pos_function_weight, cons.ori_function_weight = 0.8, 1.25
    rh, pb = _gpu_problem(torch, rig, cons, B)
    rng = np.random.default_rng(9)
    en = None
    if subset:
        en = (rng.uniform(size=rig.num_params) < 0.7).astype(np.uint8)
        pb.set_enabled(en)
    theta = rng.uniform(-0.4, 0.4, size=(B, rig.num_params)).astype(np.float32)
    lst, jtj, jtr = pb.fused_normal_equations(torch.from_numpy(theta).to(pb.device))
    jtj, jtr = jtj.cpu().numpy(), jtr.cpu().numpy()
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, cons.instance(b), theta[b].astype(np.float64), enabled=en, dtype="f64")
        dropped = np.setdiff1d(np.arange(rig.num_params), lst)
        assert np.all(J[:, dropped] == 0)  # integer bookkeeping: exact
        H = J[:, lst].T @ J[:, lst]
        g = J[:, lst].T @ r
        assert np.abs(jtj[b] - H).max() <= 2e-5 * max(1.0, np.abs(H).max()), (name, b)
        assert np.abs(jtr[b] - g).max() <= 2e-5 * max(1.0, np.abs(g).max()), (name, b)


def _sensitivity(orc, rig, cons, th0, opt, ref, eps=1e-7):
    """Relative change of the oracle's double-precision solution under an eps-sized perturbation of
    theta0 (fp32 epsilon): the conditioning of the whole 10-iteration map, per instance."""
    rng = np.random.default_rng(0)
    worst = np.zeros(th0.shape[0])
    for _ in range(3):
        rp = orc.solve_batch(rig, cons, th0.astype(np.float64) + eps * rng.normal(size=th0.shape), opt, dtype="f64")
        d = np.linalg.norm(rp["theta"] - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        worst = np.maximum(worst, d)
    return worst


@pytest.mark.parametrize("name", ["chain24_cfg1", "chain8_mixed", "humanoid72_cfg2", "humanoid72_many_units"])
def test_solve_matches_oracle_pose_parameters(torch_cuda, orc, name):
    """north_star: pose parameters within 1e-5 relative of momentum's CPU solver (here: the oracle's
    GaussNewtonSolverT<double> restatement) after 10 GN iterations, lambda = 0.05."""
    torch = torch_cuda
    rig, pp, op, B = _case(name)
    cons, th0, ths = make_problem(rig, pp, op, B, seed=12345, perturb=0.3)
    rh, pb = _gpu_problem(torch, rig, cons, B)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
    theta = torch.from_numpy(th0.copy()).to(pb.device)
    out = pb.solve(theta, opt, want_history=True)
    th = out["theta"].cpu().numpy()
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    tol = np.full(B, 1e-5)
    if name.startswith("chain"):
        # The under-determined chain fixtures (31 parameters, 9 rows) amplify a 1e-7 perturbation of
        # their fp32 inputs to > 1e-5 in the reference's own double solve, so no implementation
        # with fp32 storage can be pinned at 1e-5 there: bound those instances by 3x their measured
        # sensitivity instead.  The BASELINE configs (humanoid) are held to the strict 1e-5.
        tol = np.maximum(tol, 3.0 * _sensitivity(orc, rig, cons, th0, opt, ref))
    assert np.all(rel <= tol), (rel, tol)
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])  # integer bookkeeping: exact
    assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
    e, eref = out["error"].cpu().numpy(), ref["error"]
    assert np.abs(e - eref).max() <= 1e-4 * np.maximum(1e-3, np.abs(eref)).max()
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())


def _vacuous_limit():
    """A MinMax limit that is never active (no error, zero Jacobian row): a problem that carries it has a parameter-space
    row, so the fused solve takes its GENERAL instantiation instead of the one compiled per step rule."""
    from momentum_amd._abi import ParameterLimit

    return [ParameterLimit.minmax(0, -1e9, 1e9, 1.0)]


def _gpu_problem_with_vacuous_limit(torch, rig, cons, B):
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
        cons.pos_function_weight, cons.ori_function_weight, limits=_vacuous_limit(),
    )  # fmt: skip
    return rh, pb


@pytest.mark.parametrize("name", ["chain24_cfg1", "humanoid72_cfg2", "humanoid72_many_units"])
def test_plain_gauss_newton_instantiation_matches_the_general_one(torch_cuda, orc, name):
    """Problems without parameter-space rows run the fused kernel's instantiation per step rule (for GaussNewtonSolverT
    without a line search: the LM schedule, the backtracking loops and the on-the-fly limit rows compiled out); with a
    (vacuous) limit row the same problem runs the general instantiation.  The same arithmetic on the same path: the same
    iterates up to the compiler's scheduling, and the same parity with the oracle."""
    torch = torch_cuda
    rig, pp, op, B = _case(name)
    cons, th0, ths = make_problem(rig, pp, op, B, seed=777, perturb=0.3)
    rh, pb = _gpu_problem(torch, rig, cons, B)
    rhg, pbg = _gpu_problem_with_vacuous_limit(torch, rig, cons, B)
    pb.set_route("fused"), pbg.set_route("fused")  # (a statement about the one-launch solve's instantiations)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
    general = pbg.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    plain = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    plain2 = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert pb.last_route() == "fused" and pbg.last_route() == "fused"
    assert torch.equal(plain["theta"], plain2["theta"])  # deterministic
    assert torch.equal(plain["iterations"], general["iterations"]) and torch.equal(plain["status"], general["status"])
    a, b = plain["theta"].cpu().numpy(), general["theta"].cpu().numpy()
    h, hg = plain["error_history"].cpu().numpy(), general["error_history"].cpu().numpy()
    assert np.abs(h - hg).max() <= 1e-5 * max(1.0, np.abs(hg).max())
    if name.startswith("humanoid"):  # (the under-determined chain amplifies a rounding difference beyond any fixed bound)
        assert (np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)).max() <= 5e-6
        ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
        for x in (a, b):
            rel = np.linalg.norm(x - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
            assert rel.max() <= 1e-5, rel


@pytest.mark.parametrize("path", ["fused", "three_kernel", "fused_general"])
@pytest.mark.parametrize("mode", ["line_search", "line_search_directional", "lm_schedule"])
def test_line_search_and_lm_schedule_match_oracle(torch_cuda, orc, mode, path, monkeypatch):
    """GaussNewtonSolverT with doLineSearch (gauss_newton_solver.cpp:283-313) and the LM gain-ratio
    schedule of BASELINE configs[2] (the lambda form of trust_region_qr.cpp:244-268; no direct
    reference implementation, so parity is against the build's own oracle restatement)."""
    from momentum_amd._abi import MMX_STEP_LM_SCHEDULE

    torch = torch_cuda
    if path == "three_kernel":  # explicit J -> J^T J -> Cholesky step -> stepUpdateKernel
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig, pp, op, B = _case("humanoid72_cfg2")
    cons, th0, ths = make_problem(rig, pp, op, B, seed=4242, perturb=0.3)
    if path == "fused_general":  # (path "fused": the LM schedule runs the instantiation compiled for it alone; here the general one)
        rh, pb = _gpu_problem_with_vacuous_limit(torch, rig, cons, B)
    else:
        rh, pb = _gpu_problem(torch, rig, cons, B)
    if mode == "line_search":
        opt = GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05, do_line_search=True)
    elif mode == "line_search_directional":  # SubsetGaussNewtonSolverT / GaussNewtonSolverQRT rule
        opt = GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05, do_line_search=2)
    else:
        opt = GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    tol = np.full(B, 1e-5)
    if mode == "lm_schedule":
        # the schedule shrinks lambda (0.05 -> ~1e-4 after ten good steps), which raises the
        # conditioning of the late steps: bound by the measured sensitivity of the oracle's own
        # double solve to an fp32-epsilon input perturbation, like the chain fixtures
        tol = np.maximum(tol, 3.0 * _sensitivity(orc, rig, cons, th0, opt, ref))
    assert np.all(rel <= tol), (rel, tol)
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-4 * max(1.0, np.abs(href).max())
    if mode.startswith("line_search"):
        assert np.all(np.diff(h, axis=1) <= 1e-6 * np.abs(h[:, :-1]) + 1e-12)  # monotone
    assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])


@pytest.mark.parametrize("path", ["fused", "three_kernel"])
@pytest.mark.parametrize("rule", [1, 2])
def test_line_search_backtracking_matches_oracle(torch_cuda, orc, rule, path, monkeypatch):
    """Starts from which the full Gauss-Newton step overshoots (3 rad away, lambda = 0.01): the
    backtracking itself -- trial errors, accept tests of GaussNewtonSolverT (rule 1,
    gauss_newton_solver.cpp:283-313) and of SubsetGaussNewtonSolverT / GaussNewtonSolverQRT (rule 2,
    subset_gauss_newton_solver.cpp:117-142) -- against the oracle.  An accept test is a branch on a
    difference of errors: where the oracle's own float build takes another branch than its double
    build the instance is only required to have decreased its error."""
    torch = torch_cuda
    if path == "three_kernel":
        monkeypatch.setattr(capi, "default_route", "explicit_jacobian")
    rig, pp, op, _ = _case("humanoid72_cfg2")
    B = 48
    cons, th0, ths = make_problem(rig, pp, op, B, seed=4242, perturb=3.0)
    rh, pb = _gpu_problem(torch, rig, cons, B)
    opt = GnOptions.make(min_iterations=3, max_iterations=3, regularization=0.01, do_line_search=rule)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    ref32 = orc.solve_batch(rig, cons, th0, opt, dtype="f32")
    plain = orc.solve_batch(rig, cons, th0, GnOptions.make(min_iterations=3, max_iterations=3, regularization=0.01), dtype="f64")
    th = out["theta"].cpu().numpy()
    scale = np.maximum(1.0, np.abs(ref["theta"]).max(axis=1))
    d = np.abs(th - ref["theta"]).max(axis=1) / scale
    stable = np.abs(ref32["theta"] - ref["theta"]).max(axis=1) / scale <= 1e-3  # same branches in float and double
    backtracked = np.abs(plain["theta"] - ref["theta"]).max(axis=1) > 1e-6
    assert np.count_nonzero(stable & backtracked) >= 8  # the case does exercise the backtracking
    # lambda = 0.01 and steps of several radians: bounded by the float oracle's own distance
    tol = np.maximum(2e-4, 4.0 * np.abs(ref32["theta"] - ref["theta"]).max(axis=1) / scale)
    assert np.all(d[stable] <= tol[stable]), (d[stable], tol[stable])
    h = out["error_history"].cpu().numpy()
    assert np.all(h[:, -1] <= h[:, 0])
    e_gpu = np.array([orc.get_error(rig, cons.instance(b), th[b], "f64") for b in range(B)])
    assert np.all(e_gpu[~stable] <= h[~stable, 0])


def test_solve_convergence_bookkeeping_and_determinism(torch_cuda, orc):
    torch = torch_cuda
    rig, pp, op, B = _case("chain8_mixed")
    cons, th0, ths = make_problem(rig, pp, op, B, seed=3, perturb=0.2)
    rh, pb = _gpu_problem(torch, rig, cons, B)
    # loose threshold => instances stop early, each at its own iteration (solver.cpp:98-115)
    opt = GnOptions.make(min_iterations=2, max_iterations=30, threshold=1e5, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    it, itref = out["iterations"].cpu().numpy(), ref["iterations"]
    assert np.all(it >= 3) and np.all(it <= 30)
    assert np.abs(it - itref).max() <= 1  # the stop test compares fp32-level error differences
    th = out["theta"].cpu().numpy()
    same = it == itref
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    tol = np.maximum(1e-5, 3.0 * _sensitivity(orc, rig, cons, th0, opt, ref))
    assert np.all(rel[same] <= tol[same]), (rel, tol)
    # run-to-run determinism (pymomentum/test/test_solver2.py:195-198)
    out2 = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert torch.equal(out["theta"], out2["theta"]) and torch.equal(out["error_history"], out2["error_history"])


def test_nonfinite_input_reverts_to_initial_parameters(torch_cuda):
    # pymomentum/tensor_ik/tensor_ik.cpp:168-173
    torch = torch_cuda
    rig, pp, op, B = _case("chain8_mixed")
    cons, th0, ths = make_problem(rig, pp, op, B, seed=3)
    cons.pos_target[1, 0, 0] = np.nan
    rh, pb = _gpu_problem(torch, rig, cons, B)
    opt = GnOptions.make(min_iterations=3, max_iterations=3)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)
    st = out["status"].cpu().numpy()
    assert st[1] & 3 != 0 and np.all(np.delete(st, 1) & 3 == 0)
    assert np.array_equal(out["theta"].cpu().numpy()[1], th0[1])
    assert np.isfinite(out["theta"].cpu().numpy()).all()


def test_full_size_batch_properties(torch_cuda):
    """BASELINE config 2 at full size (B = 4096): size-independent properties -- every instance
    reduces its error by orders of magnitude, duplicates of one instance give bit-identical
    results wherever they sit in the batch, and |r|^2 == error."""
    torch = torch_cuda
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    Bs = 16
    cons, th0, ths = make_problem(rig, lm, lm, Bs, seed=12345, perturb=0.3)
    B = 4096
    rep = lambda a: np.ascontiguousarray(np.tile(a, (B // Bs,) + (1,) * (a.ndim - 1)))
    from oracle import oracle as o

    big = o.Constraints(cons.pos_parent, rep(cons.pos_offset), rep(cons.pos_target), rep(cons.pos_weight),
                        cons.ori_parent, rep(cons.ori_offset), rep(cons.ori_target), rep(cons.ori_weight))  # fmt: skip
    rh, pb = _gpu_problem(torch, rig, big, B)
    theta0 = torch.from_numpy(rep(th0)).to(pb.device)
    jac, res, err = pb.eval_jacobian(theta0)
    assert torch.allclose((res.double() ** 2).sum(dim=1), err, rtol=1e-5, atol=1e-9)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05)
    out = pb.solve(theta0.clone(), opt, want_history=True)
    th = out["theta"].view(B // Bs, Bs, -1)
    assert torch.equal(th[0].expand_as(th), th)  # position in the batch does not matter
    h = out["error_history"]
    assert torch.all(h[:, -1] < 1e-3 * h[:, 0])
    assert torch.all(out["status"] & 3 == 0) and torch.all(out["iterations"] == 10)


@pytest.mark.parametrize("shape", ["humanoid72", "rig300"])
def test_jacobian_is_the_same_whatever_the_waves_per_instance(torch_cuda, shape):
    """J-assembly deals an instance's columns to four waves below 12 288 instances per launch and to three from there on; the
    LDS-bound large rigs take eight, sixteen from 8192 instances on (launchFkJacobian, mmx_kernels.hip: measured per batch
    size): the same instances give bit-identical J, r and errors in every launch shape."""
    torch = torch_cuda
    if shape == "humanoid72":
        rig = make_humanoid72(unit=UNIT)
        lm = humanoid72_landmark_joints(rig)
        pos, ori, Bs, sizes = lm, lm, 64, (64, 2048, 12288)
    else:
        from momentum_amd import make_rig300

        rig = make_rig300(unit=UNIT)
        rng = np.random.default_rng(3)
        pos = np.sort(rng.choice(rig.num_joints, 150, replace=False))
        ori = np.sort(rng.choice(rig.num_joints, 50, replace=False))
        Bs, sizes = 8, (8, 512, 8192)
    cons, th0, _ = make_problem(rig, pos, ori, Bs, seed=4711, perturb=0.3)
    from oracle import oracle as o

    got = {}
    for B in sizes:
        rep = lambda a: np.ascontiguousarray(np.tile(a, (B // Bs,) + (1,) * (a.ndim - 1)))
        big = o.Constraints(cons.pos_parent, rep(cons.pos_offset), rep(cons.pos_target), rep(cons.pos_weight),
                            cons.ori_parent, rep(cons.ori_offset), rep(cons.ori_target), rep(cons.ori_weight))  # fmt: skip
        rh, pb = _gpu_problem(torch, rig, big, B)
        jac, res, err = pb.eval_jacobian(torch.from_numpy(rep(th0)).to(pb.device))
        torch.cuda.synchronize()
        assert torch.equal(jac[:Bs].expand(B // Bs, *jac[:Bs].shape).reshape(jac.shape), jac)  # (every copy of the 64 instances alike)
        got[B] = (jac[:Bs].cpu().numpy(), res[:Bs].cpu().numpy(), err[:Bs].cpu().numpy())
        del jac, res, err, pb, rh
        torch.cuda.empty_cache()
    for B in sizes[1:]:
        for a, b in zip(got[Bs], got[B]):
            assert np.array_equal(a, b)
    assert np.isfinite(got[Bs][0]).all() and np.abs(got[Bs][0]).max() > 0


@pytest.mark.parametrize("kp,ko", [(60, 2), (130, 90), (250, 100)])
def test_jacobian_unit_chunks_match_oracle(torch_cuda, orc, kp, ko):
    """J-assembly with more units than lanes: 66 units (a second, nearly empty chunk), 400 (the last
    chunk of a six-chunk group is partial, a seventh starts a second group) and 550 (two groups);
    repeated joints among the constraints, random weights with a zero one."""
    from momentum_amd import make_rig300

    torch = torch_cuda
    rig = make_rig300(seed=12345, unit=UNIT)
    rng = np.random.default_rng(kp)
    pp = rng.choice(rig.num_joints, size=kp, replace=True)
    op = rng.choice(rig.num_joints, size=ko, replace=True)
    B = 2
    cons, th0, ths = make_problem(rig, pp, op, B, seed=kp + ko, perturb=0.3, random_offsets=True, weights="random")
    cons.pos_weight[0, 1] = 0.0
    rh, pb = _gpu_problem(torch, rig, cons, B)
    assert pb.M == 3 * kp + 9 * ko
    theta = rng.uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    jac, res, err = jac.cpu().numpy(), res.cpu().numpy(), err.cpu().numpy()
    for b in range(B):
        J, r, e = orc.eval_jacobian(rig, cons.instance(b), theta[b].astype(np.float64), dtype="f64")
        assert np.abs(jac[b].T - J).max() <= 2e-5 * max(1.0, np.abs(J).max())
        assert np.abs(jac[b].T[np.abs(J) == 0]).max() == 0  # structural zeros are exact zeros
        assert np.abs(res[b] - r).max() <= 2e-5 * max(1.0, np.abs(r).max())
        assert abs(err[b] - e) <= 2e-5 * max(1.0, e)


def test_large_rig_config5_solve_matches_oracle(torch_cuda, orc):
    """BASELINE configs[4] shape: 300-joint hand+body rig, P = 300, 150 position + 50 orientation
    constraints (M = 900).  More than 224 solved parameters: the solve takes the three-kernel path
    with the factor in global memory (choleskyStepGlobalKernel)."""
    from momentum_amd import make_rig300

    torch = torch_cuda
    rig = make_rig300(seed=12345, unit=UNIT)
    rng = np.random.default_rng(77)
    pp = rng.choice(rig.num_joints, size=150, replace=False)
    op = rng.choice(rig.num_joints, size=50, replace=False)
    B = 3
    cons, th0, ths = make_problem(rig, pp, op, B, seed=555, perturb=0.2)
    rh, pb = _gpu_problem(torch, rig, cons, B)
    theta = rng.uniform(-0.2, 0.2, size=(B, rig.num_params)).astype(np.float32)
    jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).to(pb.device))
    J, r, e = orc.eval_jacobian(rig, cons.instance(0), theta[0].astype(np.float64), dtype="f64")
    assert np.abs(jac[0].cpu().numpy().T - J).max() <= 2e-5 * max(1.0, np.abs(J).max())
    # the matrix-core normal equations (normalEquationsMfmaKernel, 190 tiles in one pass)
    jtj, jtr, _ = pb.normal_equations(torch.from_numpy(theta).to(pb.device))
    H, g = J.T @ J, J.T @ r
    assert np.abs(jtj[0].cpu().numpy() - H).max() <= 5e-5 * max(1.0, np.abs(H).max())
    assert np.abs(jtr[0].cpu().numpy() - g).max() <= 5e-5 * max(1.0, np.abs(g).max())
    opt = GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    tol = np.maximum(1e-5, 3.0 * _sensitivity(orc, rig, cons, th0, opt, ref))
    assert np.all(rel <= tol), (rel, tol)
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
    assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
    # the driver's default doLineSearch on the large system: Cholesky step in HBM + stepUpdateKernel
    opt = GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05, do_line_search=True)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert np.all(rel <= np.maximum(1e-5, 3.0 * _sensitivity(orc, rig, cons, th0, opt, ref))), rel
    assert np.abs(out["error_history"].cpu().numpy() - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())


def test_config3_full_size_lm_schedule_properties(torch_cuda, orc):
    """BASELINE configs[2] at its full size (65 536 instances, LM damping schedule): the first
    instances are compared with the oracle, the rest through size-independent properties (a tiled
    batch gives bit-identical results wherever an instance sits; every instance ends finite with a
    decreased error and a full iteration count)."""
    from momentum_amd._abi import MMX_STEP_LM_SCHEDULE
    from oracle import oracle as o

    torch = torch_cuda
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    Bs, B = 32, 65536
    cons, th0, ths = make_problem(rig, lm, lm, Bs, seed=777, perturb=0.3)
    rep = lambda a: np.ascontiguousarray(np.tile(a, (B // Bs,) + (1,) * (a.ndim - 1)))
    big = o.Constraints(cons.pos_parent, rep(cons.pos_offset), rep(cons.pos_target), rep(cons.pos_weight),
                        cons.ori_parent, rep(cons.ori_offset), rep(cons.ori_target), rep(cons.ori_weight))  # fmt: skip
    rh, pb = _gpu_problem(torch, rig, big, B)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE)
    out = pb.solve(torch.from_numpy(rep(th0)).to(pb.device), opt, want_history=True)
    th = out["theta"].view(B // Bs, Bs, -1)
    assert torch.equal(th[0].expand_as(th), th)
    assert torch.isfinite(out["theta"]).all() and torch.all(out["status"] & 3 == 0) and torch.all(out["iterations"] == 10)
    h = out["error_history"]
    assert torch.all(h[:, -1] < 1e-3 * h[:, 0])
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    rel = np.linalg.norm(th[0].cpu().numpy() - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    tol = np.maximum(1e-5, 3.0 * _sensitivity(orc, rig, cons, th0, opt, ref))
    # The schedule takes discrete decisions (rho against 0.25 / 0.75 / 0).  When a gain ratio sits on
    # a threshold, single and double precision go different ways for an iteration and end on
    # different (equally valid) iterates: on this very batch the oracle's OWN float instantiation
    # differs from its double one by up to 6e-3 on some instances.  The smooth sensitivity estimate
    # does not see that, so at most two instances in 32 may take the other branch; they must still
    # be good solutions (error history check above).
    ref32 = orc.solve_batch(rig, cons, th0, opt, dtype="f32")
    rel32 = np.linalg.norm(ref32["theta"] - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    over = rel > tol
    assert over.sum() <= 2 and rel.max() <= max(2e-2, 3.0 * rel32.max()), (rel[over], tol[over], rel32.max())


def test_tensor_ik_default_options(torch_cuda, orc):
    """The batch driver's defaults (pymomentum/tensor_ik/solver_options.h:28-37): levmar_lambda =
    0.01, minIter = 4, maxIter = 50, threshold = 10, lineSearch = true.  Its default linear solver,
    QR (GaussNewtonSolverQRT), solves the same regularised normal equations as the Cholesky choice
    (SubsetGaussNewtonSolverT), and both use the directional line search (tensor_ik.cpp:142-158;
    MMX_LINE_SEARCH_DIRECTIONAL).  Instances stop at
    their own iteration; the converged poses must agree with the oracle's double solve."""
    torch = torch_cuda
    rig, pp, op, B = _case("humanoid72_cfg2")
    B = 8
    cons, th0, ths = make_problem(rig, pp, op, B, seed=2024, perturb=0.3)
    rh, pb = _gpu_problem(torch, rig, cons, B)
    opt = GnOptions.make(min_iterations=4, max_iterations=50, threshold=10.0, regularization=0.01, do_line_search=2)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    it, itref = out["iterations"].cpu().numpy(), ref["iterations"]
    assert np.all(it >= 5) and np.all(it <= 50)
    # the stop test compares relative error changes of ~1e-6: fp32 may fire an iteration or two apart
    assert np.abs(it - itref).max() <= 3, (it, itref)
    th = out["theta"].cpu().numpy()
    # all instances are converged, so a different stopping iteration moves theta by less than the
    # last step; compare the poses through the joint world positions like test_solver2.py:135-200
    st = pb.skeleton_state(out["theta"]).cpu().numpy()
    for b in range(B):
        sref = orc.skeleton_state(rig, ref["theta"][b], "f64")["world"]
        assert np.abs(st[b][:, :3] - sref[:, :3]).max() <= 1e-4
    e, eref = out["error"].cpu().numpy(), ref["error"]
    assert np.all(np.abs(e - eref) <= 1e-3 * eref + 1e-9), (e, eref)
    h = out["error_history"].cpu().numpy()
    assert np.all(np.diff(h[:, : it.min()], axis=1) <= 1e-6 * np.abs(h[:, : it.min() - 1]) + 1e-12)  # line search: monotone
    assert np.all(out["status"].cpu().numpy() & 3 == 0)
    del th


def test_tree_moment_normal_equations_match_the_dense_product_on_the_wide_rig(torch_cuda, orc, monkeypatch):
    """BASELINE configs[4] shape (300 joints, 150 position + 50 orientation constraints): H = J^T J and g = J^T r
    built from the tree moments (treeNormalEquationsKernel, what the explicit-Jacobian solver uses for wide
    systems) against the matrix-core product of the dense J (normalEquationsMfmaKernel) and against the oracle's
    double J; then the solve through either, same pose parameters."""
    from momentum_amd import make_rig300

    torch = torch_cuda
    rig = make_rig300(seed=12345, unit=UNIT)
    rng = np.random.default_rng(77)
    pp = rng.choice(rig.num_joints, size=150, replace=False)
    op = rng.choice(rig.num_joints, size=50, replace=False)
    B = 3
    cons, th0, ths = make_problem(rig, pp, op, B, seed=555, perturb=0.2, weights="random")
    rh, pb = _gpu_problem(torch, rig, cons, B)
    import ctypes as C


    buf = np.zeros(rig.num_params, np.int32)
    nn = C.c_int32(0)
    capi._check(capi.lib().mmx_debug_fused_normal_equations(pb._h, None, None, None, capi.as_ptr(buf, C.c_int32), C.byref(nn), None))
    lst = np.sort(buf[: nn.value])  # (the solve list is in elimination order; the hooks below report in parameter order)
    en = np.zeros(rig.num_params, np.uint8)
    en[lst] = 1  # only structurally non-zero columns enabled: the enabled system IS the solve-list system
    pb.set_enabled(en)
    theta = rng.uniform(-0.2, 0.2, size=(B, rig.num_params)).astype(np.float32)
    td = torch.from_numpy(theta).to(pb.device)
    Hd, gd, _ = pb.normal_equations(td)
    Ht, gt = pb.tree_normal_equations(td)
    Hd, Ht, gd, gt = Hd.cpu().numpy(), Ht.cpu().numpy(), gd.cpu().numpy(), gt.cpu().numpy()
    for b in range(B):
        Jm, r, e = orc.eval_jacobian(rig, cons.instance(b), theta[b].astype(np.float64), enabled=en, dtype="f64")
        Je = Jm[:, lst]
        H, g = Je.T @ Je, Je.T @ r
        scale = max(1.0, np.abs(H).max())
        assert np.abs(np.tril(Ht[b]) - np.tril(H)).max() <= 5e-5 * scale, np.abs(np.tril(Ht[b]) - np.tril(H)).max() / scale
        assert np.abs(np.tril(Hd[b]) - np.tril(H)).max() <= 5e-5 * scale
        assert np.abs(gt[b] - g).max() <= 5e-5 * max(1.0, np.abs(g).max())
    opt = GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.05)
    out_tree = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)["theta"].cpu().numpy()
    assert pb.last_route() == "wide"
    pb.set_route("explicit_jacobian")
    out_dense = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)["theta"].cpu().numpy()
    assert pb.last_route() == "explicit_jacobian"
    ref = orc.solve_batch(rig, cons, th0, opt, enabled=en, dtype="f64")
    for th in (out_tree, out_dense):
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        assert rel.max() <= 2e-5, rel


@pytest.mark.parametrize("variant", ["wide", "explicit_jacobian"])
def test_wide_solve_variants_agree_with_the_oracle(torch_cuda, orc, variant, monkeypatch):
    """The two routes of a wide solve (300-joint rig) -- tree normal equations + left-looking factor + refinement through
    the tree (no dense J), and the explicit-Jacobian kernels (dense J^T J on the matrix cores, the refinement streaming
    J) -- under the LM schedule and with elements that converge at different iterations: same pose parameters, iteration
    counts and statuses as the oracle."""
    from momentum_amd import make_rig300
    from momentum_amd._abi import MMX_STEP_LM_SCHEDULE

    torch = torch_cuda
    monkeypatch.setattr(capi, "default_route", variant)
    rig = make_rig300(seed=12345, unit=UNIT)
    rng = np.random.default_rng(78)
    pp = rng.choice(rig.num_joints, size=150, replace=False)
    op = rng.choice(rig.num_joints, size=50, replace=False)
    B = 4
    cons, th0, ths = make_problem(rig, pp, op, B, seed=556, perturb=0.2)
    th0[1] = ths[1]  # starts at its solution: converges at once, sits out the other elements' iterations
    rh, pb = _gpu_problem(torch, rig, cons, B)
    for opt in (
        GnOptions.make(min_iterations=1, max_iterations=8, threshold=1e3, regularization=0.05),
        GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE),
    ):
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
        th = out["theta"].cpu().numpy()
        den = np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
        rel = np.linalg.norm(th - ref["theta"], axis=1) / den
        tol = np.maximum(1e-5, 3.0 * _sensitivity(orc, rig, cons, th0, opt, ref))
        assert np.all(rel <= tol), (variant, rel, tol)
        assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"])
        if opt.step_rule != MMX_STEP_LM_SCHEDULE:
            assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"]), (out["iterations"], ref["iterations"])


def test_wide_path_on_a_small_skeleton_with_many_units(torch_cuda, orc, monkeypatch):
    """The wide path's other shapes: the 72-joint humanoid with constraints on EVERY joint (P = 219, n = 219: an even
    number of 16-blocks; U = 288 units > J joints, so the tree kernels take their per-joint loops) pinned to the
    wide route (MMX_ROUTE_WIDE)."""
    import bench

    torch = torch_cuda
    monkeypatch.setattr(capi, "default_route", "wide")
    rig, parents, _, _, _ = bench.build_rig("cfg2_all")
    B = 48
    db = bench.DeviceBatch(rig, parents, B, 0, 31337)
    for ls in (0, 2):
        opt = GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05, do_line_search=ls)
        out = db.pb.solve(db.theta0.clone(), opt, want_history=True)
        ref = orc.solve_batch(rig, db.host_constraints(B), db.theta0.cpu().numpy(), opt, dtype="f64")
        th = out["theta"].cpu().numpy()
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        assert rel.max() <= 1e-5, (ls, rel.max())
        assert int((out["status"] & 3 != 0).sum()) == 0
        h = out["error_history"].cpu().numpy()
        assert np.abs(h - ref["error_history"]).max() <= 1e-4 * max(1.0, np.abs(ref["error_history"]).max())
    # iterationHistory_["parameters"] on this route: row i = the parameters after iteration i (the solve is deterministic)
    opt = GnOptions.make(min_iterations=3, max_iterations=3, threshold=1.0, regularization=0.05)
    out = db.pb.solve(db.theta0.clone(), opt, want_parameter_history=True)
    hist = out["parameter_history"].cpu().numpy()
    assert np.array_equal(hist[:, 2], out["theta"].cpu().numpy())
    o2 = GnOptions.make(min_iterations=2, max_iterations=2, threshold=1.0, regularization=0.05)
    assert np.array_equal(hist[:, 1], db.pb.solve(db.theta0.clone(), o2)["theta"].cpu().numpy())
