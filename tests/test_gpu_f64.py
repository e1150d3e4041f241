"""mmx_solve_f64 = SolverT<double>::solve with GaussNewtonSolverT<double> per element (gauss_newton_solver.cpp:
315-316): against the oracle's double instantiation at 1e-10 relative on the pose parameters -- both run the
same algorithm in the same precision, so what separates them is summation order only."""
import numpy as np
import pytest

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import MMX_STEP_LM_SCHEDULE, GnOptions
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu
UNIT = 0.01


def _gpu(torch, rig, cons, B, **kw):
    from momentum_amd import capi

    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
        cons.pos_function_weight, cons.ori_function_weight, **kw,
    )  # fmt: skip
    return rh, pb


@pytest.mark.parametrize("mode", ["gn", "line_search_gn", "line_search_directional", "lm_schedule"])
@pytest.mark.parametrize("which", ["humanoid72", "chain24"])
def test_f64_solve_matches_oracle_double(torch_cuda, orc, which, mode):
    torch = torch_cuda
    if which == "humanoid72":
        rig = make_humanoid72(unit=UNIT)
        pp = op = humanoid72_landmark_joints(rig)
        B, perturb = 24, 0.3
    else:
        rig = make_test_character(24)
        pp, op, B, perturb = [23, 12, 5], [20], 8, 0.5
    cons, th0, ths = make_problem(rig, pp, op, B, seed=77, perturb=perturb, random_offsets=True, weights="random")
    rh, pb = _gpu(torch, rig, cons, B)
    kw = dict(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
    if mode == "line_search_gn":
        kw["do_line_search"] = 1
    elif mode == "line_search_directional":
        kw["do_line_search"] = 2
    elif mode == "lm_schedule":
        kw["step_rule"] = MMX_STEP_LM_SCHEDULE
    opt = GnOptions.make(**kw)
    theta = torch.from_numpy(th0.astype(np.float64)).to(pb.device)
    out = pb.solve_f64(theta, opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    # the under-determined chain fixture amplifies rounding-order differences (cond ~ 1e4): 1e-8 there
    assert rel.max() <= (1e-10 if which == "humanoid72" else 1e-8), rel
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"]) and np.array_equal(out["status"].cpu().numpy(), ref["status"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    etol = 1e-9 if which == "humanoid72" else 1e-8  # (the same amplification on the chain fixture: measured 1.8e-9 with the blocked factor)
    assert np.abs(h - href).max() <= etol * max(1.0, np.abs(href).max())
    assert np.abs(out["error"].cpu().numpy() - ref["error"]).max() <= etol * max(1.0, np.abs(ref["error"]).max())


def test_f64_solve_with_robust_loss_disabled_parameters_and_per_instance_parents(torch_cuda, orc):
    torch = torch_cuda
    rig = make_humanoid72(unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    B = 6
    rng = np.random.default_rng(8)
    pos_parents = [rng.choice(rig.num_joints, size=10).astype(np.int32) for _ in range(B)]
    conss = [make_problem(rig, pos_parents[b], lm[:6], 1, seed=50 + b, perturb=0.3, weights="random")[0] for b in range(B)]
    cat = lambda f: np.concatenate([getattr(c, f) for c in conss], axis=0)
    cons = orc.Constraints(pos_parents[0], cat("pos_offset"), cat("pos_target"), cat("pos_weight"), lm[:6], cat("ori_offset"), cat("ori_target"), cat("ori_weight"),
                           pos_function_weight=0.7, ori_function_weight=1.3, pos_loss=(0.0, 0.5), ori_loss=(1.0, 2.0))  # fmt: skip
    rh, pb = _gpu(torch, rig, cons, B, pos_loss=(0.0, 0.5), ori_loss=(1.0, 2.0))
    pb.set_instance_parents(np.stack(pos_parents), None)
    en = np.ones(rig.num_params, np.uint8)
    en[[2, 9, 30]] = 0
    pb.set_enabled(en)
    opt = GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05)
    th0 = np.zeros((B, rig.num_params))
    out = pb.solve_f64(torch.from_numpy(th0.copy()).to(pb.device), opt)
    th = out["theta"].cpu().numpy()
    for b in range(B):
        c = orc.Constraints(pos_parents[b], cons.pos_offset[b], cons.pos_target[b], cons.pos_weight[b], lm[:6], cons.ori_offset[b], cons.ori_target[b],
                            cons.ori_weight[b], pos_function_weight=0.7, ori_function_weight=1.3, pos_loss=(0.0, 0.5), ori_loss=(1.0, 2.0))  # fmt: skip
        ref = orc.solve(rig, c, th0[b], opt, enabled=en, dtype="f64")
        rel = np.linalg.norm(th[b] - ref["theta"]) / np.linalg.norm(ref["theta"])
        assert rel <= 1e-10, (b, rel)
        assert np.all(th[b][[2, 9, 30]] == 0)


@pytest.mark.parametrize("mode", ["gn", "line_search_directional", "lm_schedule"])
@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
def test_f64_solve_with_limits_and_the_model_prior(torch_cuda, orc, which, mode):
    """LimitErrorFunctionT<double> (every limit type on model / joint parameters) and ModelParametersErrorFunctionT<double>
    next to the joint constraints, some parameters disabled, per-element error-function weights: the double instantiation
    against the oracle's at 1e-10 (the reference instantiates everything for double, gauss_newton_solver.cpp:315-316)."""
    from tests.test_gpu_parameter_rows import _problem

    torch = torch_cuda
    if which == "chain8":
        rig, pp, op, B = make_test_character(8), [7, 3], [6], 4
    else:
        rig = make_humanoid72(unit=UNIT)
        pp = op = humanoid72_landmark_joints(rig)
        B = 5
    rh, pb, full, th0 = _problem(torch, orc, rig, pp, op, B, 300, True, True)
    en = np.ones(rig.num_params, np.uint8)
    en[[2, 5]] = 0
    pb.set_enabled(en)
    kw = dict(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05)
    if mode == "line_search_directional":
        kw["do_line_search"] = 2
    elif mode == "lm_schedule":
        kw["step_rule"] = MMX_STEP_LM_SCHEDULE
    opt = GnOptions.make(**kw)
    out = pb.solve_f64(torch.from_numpy(th0.astype(np.float64)).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, enabled=en, dtype="f64")
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert rel.max() <= (1e-10 if which == "humanoid72" else 1e-8), rel
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"]) and np.array_equal(out["status"].cpu().numpy(), ref["status"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-9 * max(1.0, np.abs(href).max())
    assert np.all(th[:, [2, 5]] == th0[:, [2, 5]])


def test_f64_solve_with_parameter_rows_only(torch_cuda, orc):
    """No joint constraint at all: limits and the model prior alone drive the double solve (H starts as lambda I)."""
    from momentum_amd import capi
    from momentum_amd._abi import ParameterLimit as PL

    torch = torch_cuda
    rig = make_test_character(6)
    B, P = 3, rig.num_params
    rng = np.random.default_rng(4)
    limits = [PL.minmax(1, -0.1, 0.1, 2.0), PL.linear(3, 4, 0.5, 0.1, weight=1.5), PL.halfplane(5, 6, 0.6, 0.8, 0.2)]
    mt = rng.uniform(-0.3, 0.3, size=(B, P)).astype(np.float32)
    mw = rng.uniform(0.1, 1.0, size=(B, P)).astype(np.float32)
    th0 = rng.uniform(-0.5, 0.5, size=(B, P))
    z = lambda *shape: np.zeros(shape, np.float32)
    full = orc.Constraints([], z(B, 0, 3), z(B, 0, 3), z(B, 0), [], z(B, 0, 4), z(B, 0, 4), z(B, 0), limits=limits, limit_function_weight=0.8,
                           model_target=mt, model_weights=mw, model_function_weight=1.2)  # fmt: skip
    pb = capi.Problem(capi.RigHandle(rig, 0), B, [], [])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(pb.device)
    pb.set_constraints(t(z(B, 0, 3)), t(z(B, 0, 3)), t(z(B, 0)), t(z(B, 0, 4)), t(z(B, 0, 4)), t(z(B, 0)), limits=limits, limit_function_weight=0.8,
                       model_target=t(mt), model_weights=t(mw), model_function_weight=1.2)  # fmt: skip
    opt = GnOptions.make(min_iterations=5, max_iterations=5, threshold=1.0, regularization=0.05)
    out = pb.solve_f64(torch.from_numpy(th0.copy()).to(pb.device), opt)
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert rel.max() <= 1e-10, rel


@pytest.mark.parametrize("mode", ["gn", "line_search_directional", "lm_schedule"])
@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
def test_f64_solve_with_the_further_joint_error_functions(torch_cuda, orc, which, mode):
    """PlaneErrorFunctionT<double> (plane and half plane), AimDist / AimDir, FixedAxisDiff / Cos / Angle and
    NormalErrorFunctionT<double> next to the position / orientation constraints, one block with a robust loss, one with a
    function weight, per-element error-function weights on top: the double instantiation against the oracle's at 1e-10
    (skeleton_solver_function.cpp:200-261 with T = double; evalJointConstraintF64 in mmx_f64.hip)."""
    from momentum_amd import _abi, capi
    from tests.test_gpu_joint_blocks import _device_block
    from tests.test_oracle_joint_blocks import make_block

    torch = torch_cuda
    rng = np.random.default_rng(21)
    if which == "chain8":
        rig, pp, op, B = make_test_character(8), [7, 3], [6], 4
    else:
        rig = make_humanoid72(unit=UNIT)
        pp = op = humanoid72_landmark_joints(rig)
        B = 5
    J = rig.num_joints
    cons, th0, _ = make_problem(rig, pp, op, B, seed=31, perturb=0.3, weights="random")
    pick = lambda k: rng.choice(np.arange(1, J), size=k, replace=False)
    blocks = [
        make_block(_abi.MMX_JC_PLANE, pick(3), rng, weight=1.0, batch=B),
        make_block(_abi.MMX_JC_HALF_PLANE, pick(2), rng, weight=2.0, batch=B),
        make_block(_abi.MMX_JC_AIM_DIST, pick(2), rng, weight=0.5, batch=B, function_weight=1.7),
        make_block(_abi.MMX_JC_AIM_DIR, pick(2), rng, weight=0.8, batch=B),
        make_block(_abi.MMX_JC_FIXED_AXIS_DIFF, pick(2), rng, weight=1.2, batch=B, loss=(1.0, 0.3)),
        make_block(_abi.MMX_JC_FIXED_AXIS_COS, pick(2), rng, weight=1.0, batch=B),
        make_block(_abi.MMX_JC_FIXED_AXIS_ANGLE, pick(2), rng, weight=0.7, batch=B),
        make_block(_abi.MMX_JC_NORMAL, pick(3), rng, weight=1.5, batch=B),
    ]
    fw = rng.uniform(0.5, 2.0, size=(B, 4 + len(blocks))).astype(np.float32)
    fw[1, 4] = 0.0  # element 1: the plane block switched off
    full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target,
                           cons.ori_weight, joint_blocks=blocks, function_weights=fw)  # fmt: skip
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
                       joint_blocks=[_device_block(torch, k, pb.device) for k in blocks], function_weights=t(fw, fw.shape))  # fmt: skip
    kw = dict(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05)
    if mode == "line_search_directional":
        kw["do_line_search"] = 2
    elif mode == "lm_schedule":
        kw["step_rule"] = MMX_STEP_LM_SCHEDULE
    opt = GnOptions.make(**kw)
    out = pb.solve_f64(torch.from_numpy(th0.astype(np.float64)).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    # (summation order only separates the two: 1e-12 measured.  With the loss scale's 1 / c^2 formed in float instead of
    # double -- the first version -- the error values differed in the ninth digit and theta by 2e-10 / 4.5e-8.)
    assert rel.max() <= (1e-10 if which == "humanoid72" else 1e-8), rel
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"]) and np.array_equal(out["status"].cpu().numpy(), ref["status"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-9 * max(1.0, np.abs(href).max())
    # the blocks matter: without them the answer differs
    plain = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    assert np.abs(plain["theta"] - ref["theta"]).max() > 1e-3


@pytest.mark.parametrize("mode", ["gn", "line_search_directional"])
@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
def test_f64_solve_with_ellipsoid_limits(torch_cuda, orc, which, mode):
    """LimitType::Ellipsoid entries of LimitErrorFunctionT<double> (limit_error_function.cpp:173-195,702-790: the walk from
    the constrained joint stops at the ellipsoid's joint) next to parameter limits and a half-plane block: the double
    instantiation against the oracle's at 1e-10."""
    from tests.test_gpu_ellipsoid import _setup

    torch = torch_cuda
    B = 4
    rig, pb, full, th0 = _setup(torch, orc, which, B, 17, True)
    kw = dict(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05)
    if mode == "line_search_directional":
        kw["do_line_search"] = 2
    opt = GnOptions.make(**kw)
    out = pb.solve_f64(torch.from_numpy(th0.astype(np.float64)).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert rel.max() <= (1e-10 if which == "humanoid72" else 1e-8), rel
    assert np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"]) and np.array_equal(out["status"].cpu().numpy(), ref["status"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-9 * max(1.0, np.abs(href).max())
    # the ellipsoid rows matter: without them the answer differs
    plain = orc.Constraints(full.pos_parent, full.pos_offset, full.pos_target, full.pos_weight, full.ori_parent, full.ori_offset, full.ori_target,
                            full.ori_weight, limits=full.limits, limit_function_weight=25.0, joint_blocks=full.joint_blocks)  # fmt: skip
    ref0 = orc.solve_batch(rig, plain, th0, opt, dtype="f64")
    assert np.abs(ref0["theta"] - ref["theta"]).max() > 1e-4


@pytest.mark.parametrize("radius", [1.0, 0.3])
@pytest.mark.parametrize("which", ["reference_fixture", "humanoid72_all_joints"])
def test_f64_trust_region_follows_the_oracle(torch_cuda, orc, which, radius):
    """TrustRegionQRT<double>::doIteration (trust_region_qr.cpp:52-270) in the double instantiation: the reference's
    TrustRegionTest.SanityCheck shape (solver_test.cpp:178-230: a constraint pair on every joint, random pose in
    [-1, 1]^P, start at 0) and the 72-joint humanoid with every joint constrained -- J has full column rank on both, so
    the LL^T of J^T J + (1e-20 + lambda - 1e-10) I IS the reference's QR with its appended sqrt(lambda) rows up to
    rounding: same trial decisions, error history and pose parameters on the oracle's double run."""
    from momentum_amd._abi import MMX_STEP_TRUST_REGION

    torch = torch_cuda
    if which == "reference_fixture":
        rig = make_test_character(5)
        joints = list(range(rig.num_joints))
        B, its, perturb = 16, 12, 1.0
    else:
        rig = make_humanoid72(variant="p219", unit=UNIT)
        joints = list(range(rig.num_joints))
        B, its, perturb = 6, 8, 0.3
    cons, th0, _ = make_problem(rig, joints, joints, B, seed=900, perturb=perturb)
    rh, pb = _gpu(torch, rig, cons, B)
    opt = GnOptions.make(min_iterations=its, max_iterations=its, threshold=1000.0, step_rule=MMX_STEP_TRUST_REGION, trust_region_radius=radius)
    out = pb.solve_f64(torch.from_numpy(th0.astype(np.float64)).to(pb.device), opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    assert np.array_equal(out["status"].cpu().numpy(), ref["status"]) and np.array_equal(out["iterations"].cpu().numpy(), ref["iterations"])
    h, href = out["error_history"].cpu().numpy(), ref["error_history"]
    assert np.abs(h - href).max() <= 1e-7 * max(1.0, np.abs(href).max()), np.abs(h - href).max(axis=1)
    th = out["theta"].cpu().numpy()
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    assert rel.max() <= 1e-7, rel
    assert np.all(np.diff(h, axis=1) <= 1e-9 * np.abs(h[:, :-1]) + 1e-14)  # accepted steps only ever decrease the error
