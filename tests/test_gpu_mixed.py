"""MMX_PRECISION_MIXED (ABI 11) through the C ABI: the one-launch solve with theta, forward kinematics, residual rows, g = J^T r and
the residual of the linear solve in double around a single-precision factor (momentum_amd/csrc/mmx_mixed.hpp).

The reference instantiates the whole path in double (momentum/solver/gauss_newton_solver.cpp:315-316) and its batched driver
defaults to lambda = 0.01 (pymomentum/tensor_ik/solver_options.h:28-37): the mixed route must follow the oracle's DOUBLE run

  * to ~1e-6 on the pose parameters at BASELINE's and the driver's damping (the single-precision route: 1e-6 ... 2e-6), with the
    iteration counts and (LM schedule) every accept / scale decision of the double run;
  * within north_star's 1e-5 on every stable instance of the marginally determined classes single precision cannot hold
    (BASELINE configs[0] at every damping, configs[1] from lambda = 1e-3 down) WITHOUT a single escalation to the double kernel;
  * bit-identically from run to run; with limits on model / joint parameters and the model-parameter prior aboard; outside its
    scope (trust region, further joint error functions) it is the double instantiation.
"""
import numpy as np
import pytest

from momentum_amd import capi, humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import (
    MMX_PRECISION_F64,
    MMX_PRECISION_MIXED,
    MMX_SOLVE_ESCALATED_F64,
    MMX_SOLVE_MIXED,
    MMX_SOLVE_PRECISION_SUSPECT,
    MMX_STEP_LM_SCHEDULE,
    MMX_STEP_TRUST_REGION,
    GnOptions,
)
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu
UNIT = 0.01
BOUND = 1e-5


def _cores():
    import bench

    return bench.usable_cores()


def _rel(a, ref):
    return np.linalg.norm(a - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)


def _problem(torch, rig, cons, B):
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))  # fmt: skip
    if capi.default_route == "prefer_wide":
        pb.set_route("fused")  # (the forced-wide sweep of scripts/gpu_cross_checks.sh: this file is about the one-launch route's instantiation)
    return pb


def _solve(torch, pb, th0, opt, **kw):
    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items() if v is not None}


def _cfg2(B, seed=777, variant="p128"):
    rig = make_humanoid72(seed=12345, variant=variant, unit=UNIT)
    lm = humanoid72_landmark_joints(rig)
    cons, th0, _ = make_problem(rig, lm, lm, B, seed=seed, perturb=0.3)
    return rig, cons, th0


@pytest.mark.parametrize("lam", [0.05, 0.01])
@pytest.mark.parametrize("line_search", [0, 1, 2])
def test_mixed_follows_the_double_run_at_the_baseline_and_driver_damping(torch_cuda, orc, lam, line_search):
    B = 512
    rig, cons, th0 = _cfg2(B)
    pb = _problem(torch_cuda, rig, cons, B)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=line_search, precision=MMX_PRECISION_MIXED)
    out = _solve(torch_cuda, pb, th0, opt, want_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores())
    assert np.all(out["status"] & ~4 == MMX_SOLVE_MIXED), np.unique(out["status"])  # (4: the factor's damping floor engaged -- informational)
    assert np.array_equal(out["iterations"], ref["iterations"])
    rel = _rel(out["theta"].astype(np.float64), ref["theta"])
    h, href = out["error_history"], ref["error_history"]
    same = np.all(np.abs(h - href) <= 1e-6 * np.abs(href) + 1e-7 * href[:, :1], axis=1)  # (a line-search decision on its threshold may go the other way)
    assert same.sum() >= 0.99 * B, int(same.sum())
    assert rel[same].max() <= 2e-6, (float(rel[same].max()), float(np.median(rel)))
    assert np.median(rel) <= 5e-7
    # the error the solve reports is the double run's
    assert np.allclose(out["error"][same], ref["error"][same], rtol=1e-6, atol=1e-12)


def test_mixed_lm_schedule_takes_the_double_runs_decisions(torch_cuda, orc):
    """configs[2]'s schedule: (lambda, gain ratio) per iteration against the oracle's double run -- the damping is carried in
    double like GaussNewtonSolverT<double>'s, so an element with the double run's decisions has its lambda sequence EXACTLY."""
    import bench

    B = 1024
    rig, cons, th0 = _cfg2(B, seed=99)
    pb = _problem(torch_cuda, rig, cons, B)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE, precision=MMX_PRECISION_MIXED)
    out = _solve(torch_cuda, pb, th0, opt, want_history=True, want_step_history=True)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores(), step_history=True)
    rel = _rel(out["theta"].astype(np.float64), ref["theta"])
    res = bench.lm_branch_analysis(out["step_history"], out["error_history"], ref, rel)
    assert res["same_decisions"] >= 0.995 * B and res["same_decisions_lambda_sequences_equal"], res
    assert res["num_above_bound_with_same_decisions"] == 0 and res["max_rel_same_decisions"] <= 2e-6, res
    assert np.all(out["status"] & MMX_SOLVE_MIXED != 0) and np.all(out["status"] & (3 | MMX_SOLVE_ESCALATED_F64) == 0)


SHAPES = {
    "cfg1": (lambda: make_test_character(24), [23, 12, 5], [], (5e-2, 1e-2, 1e-3)),
    "cfg2": (lambda: make_humanoid72(seed=12345, variant="p128", unit=UNIT), "lm", "lm", (1e-2, 1e-3, 1e-5)),
}


@pytest.mark.parametrize("line_search", [0, 2])
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_mixed_holds_the_bound_where_single_precision_does_not(torch_cuda, orc, name, line_search):
    """Every element whose double run converges, is itself a stable computation (the oracle's double run from theta0 + 1e-12
    ends within 1e-7 of its run from theta0) and whose line-search decisions are the double run's is within 1e-5 -- with ZERO
    escalations to the double kernel (tests/test_gpu_precision.py has the same statement for MMX_PRECISION_AUTO)."""
    mk, pp, op, lams = SHAPES[name]
    rig = mk()
    if pp == "lm":
        pp = op = humanoid72_landmark_joints(rig)
    B = 1024
    cons, th0, _ = make_problem(rig, pp, op, B, seed=777, perturb=0.3)
    pb = _problem(torch_cuda, rig, cons, B)
    e0 = np.array([orc.get_error(rig, cons.instance(b), th0[b].astype(np.float64), "f64") for b in range(0, B, 64)]).max()
    for lam in lams:
        opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=line_search, precision=MMX_PRECISION_MIXED)
        out = _solve(torch_cuda, pb, th0, opt, want_history=True)
        with np.errstate(all="ignore"):
            ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores())
            pert = orc.solve_batch(rig, cons, th0.astype(np.float64) + 1e-12, opt, dtype="f64", nthreads=_cores())
            sane = (ref["status"] == 0) & np.isfinite(ref["theta"]).all(axis=1) & (ref["error"] <= e0)
            stable = _rel(pert["theta"], ref["theta"]) <= 1e-7
        rel = _rel(out["theta"].astype(np.float64), ref["theta"])
        h, href = out["error_history"], ref["error_history"]
        same = np.all(np.abs(h - href) <= 1e-6 * np.abs(href) + 1e-7 * href[:, :1], axis=1) if line_search else np.ones(B, bool)
        held = sane & same & stable & (out["status"] & MMX_SOLVE_PRECISION_SUSPECT == 0)
        assert np.all(out["status"] & MMX_SOLVE_MIXED != 0) and np.all(out["status"] & MMX_SOLVE_ESCALATED_F64 == 0)
        flagged = out["status"] & MMX_SOLVE_PRECISION_SUSPECT != 0
        if name == "cfg2" and lam <= 1e-5:
            # cond(J^T J + lambda I) = 1.4e7 here: cond x eps_f32 ~ 1, the single-precision factor is no preconditioner any more and
            # the conjugate gradients run into their step limit -- the route SAYS so (MMX_SOLVE_PRECISION_SUSPECT on the element;
            # MMX_PRECISION_AUTO then takes it to the double kernel, tests/test_gpu_precision.py); nothing unflagged is above the bound
            assert flagged.mean() >= 0.9, float(flagged.mean())
        else:
            assert held.sum() >= 0.9 * (sane & stable).sum(), (name, lam, line_search, int(sane.sum()), int(stable.sum()), int(same.sum()), int(held.sum()))
            assert flagged.mean() <= 0.02, float(flagged.mean())
        if held.any():
            assert rel[held].max() <= BOUND, (name, lam, line_search, float(rel[held].max()), int((rel[held] > BOUND).sum()))
        assert np.isfinite(out["theta"]).all()


def test_mixed_is_bit_reproducible_and_records_histories(torch_cuda):
    B = 256
    rig, cons, th0 = _cfg2(B)
    pb = _problem(torch_cuda, rig, cons, B)
    opt = GnOptions.make(min_iterations=6, max_iterations=6, threshold=1.0, regularization=1e-3, do_line_search=2, precision=MMX_PRECISION_MIXED)
    a = _solve(torch_cuda, pb, th0, opt, want_history=True, want_parameter_history=True)
    b = _solve(torch_cuda, pb, th0, opt, want_history=True, want_parameter_history=True)
    for k in ("theta", "error", "iterations", "status", "error_history", "parameter_history"):
        assert np.array_equal(a[k], b[k]), k
    # iterationHistory_["parameters"].col(i) = the parameters after iteration i (solver.cpp:101-106): the last one is theta
    assert np.array_equal(a["parameter_history"][:, -1, :], a["theta"])


def test_mixed_with_an_enabled_subset_and_per_instance_weights(torch_cuda, orc):
    B = 128
    rig, cons, th0 = _cfg2(B, seed=31)
    rng = np.random.default_rng(5)
    cons.pos_weight[:] = rng.uniform(0.2, 2.0, size=cons.pos_weight.shape).astype(np.float32)
    cons.ori_weight[:] = rng.uniform(0.2, 2.0, size=cons.ori_weight.shape).astype(np.float32)
    cons.pos_weight[:, 3] = 0.0  # a constraint with weight 0 keeps zero rows (joint_error_function-inl.h:197-199)
    pb = _problem(torch_cuda, rig, cons, B)
    enabled = np.ones(rig.num_params, np.uint8)
    enabled[rng.choice(rig.num_params, size=40, replace=False)] = 0
    pb.set_enabled(enabled)
    opt = GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.02, precision=MMX_PRECISION_MIXED)
    out = _solve(torch_cuda, pb, th0, opt)
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores(), enabled=enabled)
    assert _rel(out["theta"].astype(np.float64), ref["theta"]).max() <= 2e-6
    assert np.array_equal(out["theta"][:, enabled == 0], th0[:, enabled == 0])  # disabled parameters do not move


def test_mixed_outside_its_scope_is_the_double_instantiation(torch_cuda):
    """The trust region is not built into the mixed instantiation: MMX_PRECISION_MIXED then runs the double kernel on every
    element: MMX_PRECISION_F64's result bit for bit (no MMX_SOLVE_MIXED bit on the status)."""
    B = 64
    rig, cons, th0 = _cfg2(B)
    pb = _problem(torch_cuda, rig, cons, B)
    mk = lambda prec: GnOptions.make(min_iterations=5, max_iterations=5, threshold=1.0, step_rule=MMX_STEP_TRUST_REGION, precision=prec)
    a = _solve(torch_cuda, pb, th0, mk(MMX_PRECISION_MIXED))
    d = _solve(torch_cuda, pb, th0, mk(MMX_PRECISION_F64))
    assert np.all(a["status"] & MMX_SOLVE_MIXED == 0)
    assert np.array_equal(a["theta"], d["theta"]) and np.array_equal(a["status"], d["status"])


@pytest.mark.parametrize("seed", range(40))
def test_mixed_on_random_rigs_follows_the_double_run(torch_cuda, orc, seed):
    """The randomised rigs of tests/test_gpu_fuzz.py (chains, stars, bushy trees; shared parameters, translation / scale dofs,
    transform rows with three entries, non-zero transform offsets, random pre-rotations, random constraint sets, weights and
    enabled masks) through the mixed-precision instantiation: every block count from one to eight, every step rule it carries, at a
    damping where single precision is held to 2e-5 only -- within 2e-6 of the oracle's double run, iteration counts and error
    histories the double run's."""
    from tests.test_gpu_fuzz import random_rig

    torch = torch_cuda
    rng = np.random.default_rng(7000 + seed)
    J = int(rng.integers(2, 110))
    rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
    P = rig.num_params
    Kp, Ko = int(rng.integers(0, 9)), int(rng.integers(0, 6))
    if Kp + Ko == 0:
        Kp = 1
    pp = rng.integers(0, J, size=Kp).astype(np.int32)
    op = rng.integers(0, J, size=Ko).astype(np.int32)
    B = 3
    cons, th0, _ = make_problem(rig, pp, op, B, seed=seed, perturb=0.25, random_offsets=True, weights="random")
    pb = _problem(torch, rig, cons, B)
    en = (rng.uniform(size=P) < 0.8).astype(np.uint8)
    en[:3] = 1
    pb.set_enabled(en)
    rule = seed % 4  # plain, GaussNewtonSolverT's line search, the driver's, the LM schedule
    opt = GnOptions.make(min_iterations=6, max_iterations=6, regularization=0.1, do_line_search=rule if rule < 3 else 0,
                         step_rule=MMX_STEP_LM_SCHEDULE if rule == 3 else 0, precision=MMX_PRECISION_MIXED)  # fmt: skip
    try:
        out = _solve(torch, pb, th0, opt, want_history=True)
    except capi.MmxError as e:  # (more than 128 solved parameters: the wide route's problem -- the double kernel takes MIXED there)
        pytest.skip(str(e))
    ref = orc.solve_batch(rig, cons, th0, opt, enabled=en, dtype="f64")
    st = out["status"]
    if not np.all(st & MMX_SOLVE_MIXED != 0):
        assert np.all(st & MMX_SOLVE_MIXED == 0)  # outside the instantiation's scope (block count): the double kernel, all or nothing
    assert np.all(st & 3 == 0) and np.all(st & MMX_SOLVE_PRECISION_SUSPECT == 0), st
    assert np.array_equal(out["iterations"], ref["iterations"])
    h, href = out["error_history"], ref["error_history"]
    same = np.all(np.abs(h - href) <= 1e-6 * np.abs(href) + 1e-9 * href[:, :1], axis=1)
    assert same.sum() >= B - 1, (seed, h, href)  # (a line-search / gain-ratio decision on its threshold may go the other way on one)
    rel = np.linalg.norm(out["theta"].astype(np.float64) - ref["theta"], axis=1) / np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
    assert rel[same].max() <= 2e-6, (seed, rel)
    assert np.all(out["theta"][:, en == 0] == th0[:, en == 0])  # disabled parameters are never touched


@pytest.mark.parametrize("which", ["chain8", "humanoid72"])
@pytest.mark.parametrize("blocks", ["limits", "model", "both"])
@pytest.mark.parametrize("rule", [0, 2, 3])
def test_mixed_with_parameter_space_rows_follows_the_double_run(torch_cuda, orc, which, blocks, rule):
    """LimitErrorFunctionT<double> (MinMax, Linear incl. piecewise ranges, HalfPlane, MinMaxJoint, LinearJoint) and
    ModelParametersErrorFunctionT<double> (some weights <= 0: rows dropped) inside the mixed-precision instantiation -- every
    production solve carries both (momentum/marker_tracking/marker_tracker.cpp:916-918,956-960): their share of g, of the CG's
    operator and of the errors in double, their J^T J in the single-precision preconditioner.  Plain steps, the driver's line
    search and the LM schedule, within 2e-6 of the oracle's double run."""
    from tests.test_gpu_parameter_rows import _problem as rows_problem

    torch = torch_cuda
    if which == "chain8":
        rig, pp, op, B = make_test_character(8), [7, 3], [6], 8
    else:
        rig = make_humanoid72(unit=UNIT)
        pp = op = humanoid72_landmark_joints(rig)
        B = 6
    rh, pb, full, th0 = rows_problem(torch, orc, rig, pp, op, B, 300, blocks != "model", blocks != "limits")
    if capi.default_route == "prefer_wide":
        pb.set_route("fused")
    opt = GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05, do_line_search=rule if rule < 3 else 0,
                         step_rule=MMX_STEP_LM_SCHEDULE if rule == 3 else 0, precision=MMX_PRECISION_MIXED)  # fmt: skip
    out = _solve(torch, pb, th0, opt, want_history=True)
    ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
    assert np.all(out["status"] & MMX_SOLVE_MIXED != 0) and np.all(out["status"] & (3 | MMX_SOLVE_PRECISION_SUSPECT) == 0), out["status"]
    assert np.array_equal(out["iterations"], ref["iterations"])
    h, href = out["error_history"], ref["error_history"]
    same = np.all(np.abs(h - href) <= 1e-6 * np.abs(href) + 1e-9 * href[:, :1], axis=1)
    assert same.sum() >= B - 1, (h, href)
    rel = _rel(out["theta"].astype(np.float64), ref["theta"])
    if rule >= 2:
        # A backtracking / gain-ratio decision at a CONVERGED iterate compares error differences of 1e-9 that went through getError's
        # float rounding (skeleton_solver_function.cpp:82): two double implementations halve (reject) such a step differently, theta
        # moves by 1e-4 in a direction the error does not see.  The double KERNEL -- an independent implementation of the same
        # instantiation -- is the witness: the mixed route takes ITS decisions on every element (measured: the same theta to nine
        # digits), and the oracle's on most.
        d = _solve(torch, pb, th0, GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05, do_line_search=rule if rule < 3 else 0,
                                                  step_rule=MMX_STEP_LM_SCHEDULE if rule == 3 else 0, precision=MMX_PRECISION_F64))  # fmt: skip
        assert _rel(out["theta"].astype(np.float64), d["theta"].astype(np.float64)).max() <= 2e-6
        assert (rel <= 2e-6).sum() >= B // 2, rel
    else:
        assert rel[same].max() <= 2e-6, rel
