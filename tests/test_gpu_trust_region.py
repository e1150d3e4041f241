"""MMX_STEP_TRUST_REGION = TrustRegionQRT::doIteration (momentum/character_solver/trust_region_qr.cpp:52-270)
in the fused kernel and -- driven from the host, several kernels per trust step -- on the wide route (systems beyond
the fused instantiations, further joint error functions / ellipsoid limits), against the oracle's line-by-line
restatement (pinned by the reference's own TrustRegionTest shapes in tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from momentum_amd import capi  # noqa: E402  (default_route: which kernels the problems of a test run)

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import MMX_STEP_TRUST_REGION, GnOptions
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu
UNIT = 0.01


def _gpu(torch, rig, cons, B):

    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)), 1.0, 1.0,
    )  # fmt: skip
    return rh, pb


@pytest.mark.parametrize("route", ["fused", "wide"])
@pytest.mark.parametrize("radius", [1.0, 0.3])
def test_trust_region_matches_oracle_on_the_reference_fixture(torch_cuda, orc, radius, route):
    """The reference's TrustRegionTest.SanityCheck shape (solver_test.cpp:178-230): position + orientation
    constraint on every joint of createTestCharacter, targets from a random pose in [-1, 1]^P, start at 0.
    J has full column rank there, so the (almost) undamped steps are well defined in single precision:
    the error history follows the oracle's double-precision run (same trial decisions) and the pose
    parameters agree to 1e-4 after 12 iterations (the Newton updates of lambda divide two fp32 quadratic
    forms, which is where single and double precision part beyond 1e-5)."""
    torch = torch_cuda
    rig = make_test_character(5)
    J = rig.num_joints
    B = 16
    cons, th0, ths = make_problem(rig, list(range(J)), list(range(J)), B, seed=900, perturb=1.0)
    rh, pb = _gpu(torch, rig, cons, B)
    pb.set_route(route)
    opt = GnOptions.make(min_iterations=12, max_iterations=12, threshold=1000.0, step_rule=MMX_STEP_TRUST_REGION, trust_region_radius=radius)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert pb.last_route() == route
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    assert np.array_equal(out["status"].cpu().numpy() & 3, ref["status"]) and np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    same = np.all(np.abs(h - href) <= 2e-3 * np.abs(href) + 1e-6 * href[:, :1], axis=1)
    assert same.mean() >= 0.8, (same.mean(), np.abs(h - href).max(axis=1))
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert rel[same].max() <= 1e-4, rel
    # whatever path an element took, it is a good solution: the reference's own acceptance bound against GN
    gn = orc.solve_batch(rig, cons, th0, GnOptions.make(min_iterations=12, max_iterations=12, threshold=1000.0, regularization=0.05), dtype="f64")
    for b in range(B):
        e_tr = orc.get_error(rig, cons.instance(b), th[b].astype(np.float64), "f64")
        e_gn = orc.get_error(rig, cons.instance(b), gn["theta"][b], "f64")
        assert e_tr <= 1.001 * e_gn + 0.001, (b, e_tr, e_gn)


@pytest.mark.parametrize("route", ["fused", "wide"])
def test_trust_region_on_the_humanoid_does_at_least_as_well_as_gauss_newton(torch_cuda, orc, route):
    """BASELINE configs[1]'s rig has redundant rotation dofs (J^T J is singular), where the reference's
    undamped first steps are defined by rounding; parity there is the reference's own criterion
    (solver_test.cpp:228: err_tr <= 1.001 err_gn + 0.001), every element, plus determinism."""
    torch = torch_cuda
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    B = 64
    cons, th0, ths = make_problem(rig, lm, lm, B, seed=31, perturb=0.3)
    rh, pb = _gpu(torch, rig, cons, B)
    pb.set_route(route)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, step_rule=MMX_STEP_TRUST_REGION)
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert pb.last_route() == route
    out2 = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
    assert torch.equal(out["theta"], out2["theta"]) and torch.equal(out["error_history"], out2["error_history"])
    assert int((out["status"] & 3 != 0).sum()) == 0  # (bit 4, MMX_SOLVE_DAMPING_FLOORED, is set: the rule starts from lambda = 1e-10)
    th = out["theta"].cpu().numpy()
    gn = orc.solve_batch(rig, cons, th0, GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05), dtype="f64")
    h = out["error_history"].cpu().numpy()
    assert np.all(np.diff(h, axis=1) <= 1e-6 * np.abs(h[:, :-1]) + 1e-12)  # accepted steps only ever decrease the error
    for b in range(B):
        e_tr = orc.get_error(rig, cons.instance(b), th[b].astype(np.float64), "f64")
        e_gn = orc.get_error(rig, cons.instance(b), gn["theta"][b], "f64")
        assert e_tr <= 1.001 * e_gn + 0.001, (b, e_tr, e_gn)


@pytest.mark.parametrize("config,B,route", [("cfg2_all", 24, "wide"), ("cfg2_all", 24, "fused"), ("cfg5", 12, "auto")])
def test_trust_region_on_systems_beyond_the_fused_solve(torch_cuda, orc, config, B, route):
    """tensor_ik.cpp:150-152 selects TrustRegionQR for ANY problem: the 219-parameter humanoid (every joint constrained;
    J has full column rank, so the double run is a meaningful reference; it fits the fused solve's largest instantiation and
    is run on both routes) and BASELINE configs[4]'s 300-joint rig, which takes the rule on the wide route automatically
    (the fused solve ends at 224 solved parameters).  Error history on the
    oracle's double run where the trial decisions agree, the reference's acceptance bound against Gauss-Newton
    everywhere, determinism."""
    import bench

    torch = torch_cuda
    rig, parents, _, _, _ = bench.build_rig(config)
    db = bench.DeviceBatch(rig, parents, B, 0, 4321)
    its = 8
    opt = GnOptions.make(min_iterations=its, max_iterations=its, threshold=1.0, step_rule=MMX_STEP_TRUST_REGION)
    db.pb.set_route(route)
    out = db.pb.solve(db.theta0.clone(), opt, want_history=True)
    assert db.pb.last_route() == ("wide" if route == "auto" else route)
    out2 = db.pb.solve(db.theta0.clone(), opt, want_history=True)
    assert torch.equal(out["theta"], out2["theta"]) and torch.equal(out["error_history"], out2["error_history"])
    assert int((out["status"] & 1 != 0).sum()) == 0
    cons = db.host_constraints(B)
    th0 = np.zeros((B, rig.num_params), np.float32)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=bench.usable_cores())
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.all(np.diff(h, axis=1) <= 1e-6 * np.abs(h[:, :-1]) + 1e-12)  # accepted steps only ever decrease the error
    same = np.all(np.abs(h - href) <= 2e-3 * np.abs(href) + 1e-6 * href[:, :1], axis=1)
    th = out["theta"].cpu().numpy()
    if config == "cfg2_all":
        assert same.mean() >= 0.7, (same.mean(), np.abs(h - href).max(axis=1))
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        assert rel[same].max() <= 1e-4, rel
    gn = orc.solve_batch(rig, cons, th0, GnOptions.make(min_iterations=its, max_iterations=its, threshold=1.0, regularization=0.05), dtype="f64", nthreads=bench.usable_cores())
    for b in range(B):
        e_tr = orc.get_error(rig, cons.instance(b), th[b].astype(np.float64), "f64")
        e_gn = orc.get_error(rig, cons.instance(b), gn["theta"][b], "f64")
        assert e_tr <= 1.001 * e_gn + 0.001, (b, e_tr, e_gn)


def test_trust_region_with_further_joint_blocks_and_limits(torch_cuda, orc):
    """A tracker-shaped problem under the trust region: plane block + parameter limits next to the landmark constraints.
    The fused solve's trust-region instantiation has no general rows, so the rule runs on the wide route; the oracle's
    TrustRegionQRT restatement carries the same rows."""
    import bench

    torch = torch_cuda
    rig, parents, _, _, _ = bench.build_rig("cfg2_tracker")
    B = 32
    db = bench.DeviceBatch(rig, parents, B, 0, 99, tracker=True)
    opt = GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, step_rule=MMX_STEP_TRUST_REGION)
    out = db.pb.solve(db.theta0.clone(), opt, want_history=True)
    assert db.pb.last_route() == "wide"
    assert int((out["status"] & 1 != 0).sum()) == 0
    cons = db.host_constraints(B)
    th0 = np.zeros((B, rig.num_params), np.float32)
    h = out["error_history"].cpu().numpy()
    assert np.all(np.diff(h, axis=1) <= 1e-6 * np.abs(h[:, :-1]) + 1e-12)
    th = out["theta"].cpu().numpy()
    gn = orc.solve_batch(rig, cons, th0, GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05), dtype="f64")
    tr = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    for b in range(B):
        e_tr = orc.get_error(rig, cons.instance(b), th[b].astype(np.float64), "f64")
        e_gn = orc.get_error(rig, cons.instance(b), gn["theta"][b], "f64")
        e_or = orc.get_error(rig, cons.instance(b), tr["theta"][b], "f64")
        assert e_tr <= 1.001 * e_gn + 0.001 and e_tr <= 1.001 * e_or + 0.001, (b, e_tr, e_gn, e_or)


def test_trust_region_is_refused_on_the_explicit_jacobian_route(torch_cuda):
    torch = torch_cuda
    rig = make_test_character(5)
    cons, th0, _ = make_problem(rig, [4], [3], 2, seed=1)
    rh, pb = _gpu(torch, rig, cons, 2)
    pb.set_route("explicit_jacobian")
    with pytest.raises(capi.MmxError) as ei:
        pb.solve(torch.from_numpy(th0.copy()).to(pb.device), GnOptions.make(step_rule=MMX_STEP_TRUST_REGION))
    assert "MMX_STEP_TRUST_REGION" in str(ei.value)


@pytest.mark.parametrize("precision", ["auto", "mixed"])
def test_trust_region_under_auto_and_mixed_holds_1e5_on_the_reference_fixture(torch_cuda, orc, precision):
    """TrustRegionQR is one of the three solvers solveTensorIKProblem can pick (tensor_ik.cpp:150-152).  The single-precision
    instantiation of the rule is held to 1e-4 on >= 80 % same-path elements (above): the Newton updates of lambda divide two fp32
    quadratic forms.  The rule's elements are marginal by construction (it starts from lambda = 1e-10: the factor's damping floor
    engages on every one) and the mixed instantiation does not carry the rule, so both policies run it in the double kernel:
    north_star's 1e-5 on EVERY element of the reference's TrustRegionTest shape (solver_test.cpp:178-230), iteration counts and
    error histories the oracle's double run's."""
    from momentum_amd._abi import MMX_PRECISION_AUTO, MMX_PRECISION_MIXED, MMX_SOLVE_MIXED

    torch = torch_cuda
    rig = make_test_character(5)
    J = rig.num_joints
    B = 64
    cons, th0, ths = make_problem(rig, list(range(J)), list(range(J)), B, seed=900, perturb=1.0)
    rh, pb = _gpu(torch, rig, cons, B)
    for radius in (1.0, 0.3):
        opt = GnOptions.make(min_iterations=12, max_iterations=12, threshold=1000.0, step_rule=MMX_STEP_TRUST_REGION, trust_region_radius=radius,
                             precision=MMX_PRECISION_AUTO if precision == "auto" else MMX_PRECISION_MIXED)  # fmt: skip
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
        st = out["status"].cpu().numpy()
        assert np.all(st & MMX_SOLVE_MIXED == 0) and np.all(st & 3 == 0)
        assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
        h, href = out["error_history"].cpu().numpy(), ref["error_history"]
        assert np.all(np.abs(h - href) <= 1e-6 * np.abs(href) + 1e-9 * href[:, :1])
        th = out["theta"].cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        assert rel.max() <= 1e-5, rel.max()


def test_trust_region_under_auto_holds_1e5_on_the_all_joints_humanoid(torch_cuda, orc):
    """The same statement on SURVEY 8(d)'s stress variant (P = 219, both constraints on all 72 joints: J has full column rank, so the
    double run is a meaningful reference): MMX_PRECISION_AUTO runs the rule in the double kernel."""
    import bench
    from momentum_amd._abi import MMX_PRECISION_AUTO

    torch = torch_cuda
    rig, parents, _, _, _ = bench.build_rig("cfg2_all")
    B = 24
    db = bench.DeviceBatch(rig, parents, B, 0, 4321)
    opt = GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, step_rule=MMX_STEP_TRUST_REGION, precision=MMX_PRECISION_AUTO)
    out = db.pb.solve(db.theta0.clone(), opt, want_history=True)
    st = out["status"].cpu().numpy()
    assert np.all(st & 3 == 0)
    ref = orc.solve_batch(rig, db.host_constraints(B), np.zeros((B, rig.num_params), np.float32), opt, dtype="f64", nthreads=bench.usable_cores())
    th = out["theta"].cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    same = np.all(np.abs(h - href) <= 1e-6 * np.abs(href) + 1e-9 * href[:, :1], axis=1)  # (a trial decision on its threshold may differ between two double implementations)
    assert same.mean() >= 0.9, same.mean()
    assert rel[same].max() <= 1e-5, rel
