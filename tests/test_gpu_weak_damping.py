"""Parity of the HIP path where the damping is weak (lambda 1e-7 ... 1e-2), through the C ABI, against the oracle's
double solve.  The reference's own end-to-end IK test runs GN at lambda = 1e-7
(momentum/test/character_solver/inverse_kinematics_test.cpp:60-121), pymomentum's test_solver2.py at 1e-5, solve_ik
defaults to 0.01 (pymomentum/tensor_ik/solver_options.h:28-37); fp32 normal equations + an in-LDS Cholesky are exactly
where a small lambda bites, and the refinement step of the kernels is the defence.

Three kinds of problems, three kinds of bounds (stated at each assert):

  * well-determined problems (a constraint on every joint): pose parameters within 1e-5 relative of the double solve
    at EVERY lambda, every instance -- fused and wide routes;
  * the reference's 3-joint known-answer test: its own assertions (error <= 5e-7, end effector <= 5e-5);
  * the BASELINE shapes cfg1 (24-joint chain, 3 position constraints: 9 rows for 31 parameters) and cfg2 (16 landmark
    joints: as many independent rows as solved parameters): under-determined or marginally determined, so with a weak
    lambda the minimiser moves by O(1) under a 1e-7 perturbation and NO single-precision solver holds 1e-5 on theta
    -- the oracle's own float instantiation (the restatement of GaussNewtonSolverT<float>) does not either.  Per
    (lambda, line search, route) the test reports how many instances are within 1e-5, holds every instance above it
    to the float-oracle-also-above rule, holds the HIP path to be no further from the double solve than the float
    oracle is (in distribution), and holds the objective to the criterion the reference compares its own solvers by
    (momentum/test/character_solver/solver_test.cpp:43-121: err <= 1.001 err_ref + 0.001).
The per-lambda table is written to gpurun_out/weak_damping_report.json (copied to profiles/ when it is to be judged)."""
import json
import os

import numpy as np
import pytest

from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_test_character
from momentum_amd._abi import GnOptions
from tests.helpers import make_problem

pytestmark = pytest.mark.gpu

UNIT = 0.01
BOUND = 1e-5
LAMBDAS = [1e-7, 1e-5, 1e-3, 1e-2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_REPORT = {}


def _cores():
    import bench

    return bench.usable_cores()


def _write_report():
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "weak_damping_report.json"), "w") as f:
        json.dump(_REPORT, f, indent=1, sort_keys=True)


def _gpu_solve(torch, rig, cons, th0, opt, route, parameter_history=False):
    from momentum_amd import capi

    B = th0.shape[0]
    rh = capi.RigHandle(rig, 0)
    pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
    dev = pb.device
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
    pb.set_constraints(
        t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
        t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
    )  # fmt: skip
    pb.set_route(route)
    out = pb.solve(torch.from_numpy(th0.copy()).to(dev), opt, want_history=True, want_parameter_history=parameter_history)
    torch.cuda.synchronize()
    res = {k: v.cpu().numpy() for k, v in out.items() if v is not None}
    res["route_taken"] = pb.last_route()
    return res


def _backtracking_differs(orc, rig, cons, th0, opt_of, idx, parameter_history):
    """For the instances idx: does some iteration's step of the HIP path (from its parameter history) differ in LENGTH from
    the double run's -- a backtracking decision that went the other way (alpha is halved per trial, so the lengths then
    differ by a power of two)?  A nearly converged instance takes steps whose decrease of the error is below what a
    single-precision sum of squares resolves: the accept test then fails on noise and the step is halved where the double
    run accepts it (measured: iteration 7 of 10, steps of 1e-5 |theta|, ratios 1/4 ... 1/16) -- the error history, flat to
    ten digits there, cannot show it.  The double run's iterates come from re-running it for k = 1 ... iterations."""
    sub = cons.subset(idx)
    prev = th0[idx].astype(np.float64)
    prev_g = prev.copy()
    differs = np.zeros(len(idx), bool)
    for k in range(1, parameter_history.shape[1] + 1):
        rk = orc.solve_batch(rig, sub, th0[idx], opt_of(k), dtype="f64", nthreads=_cores())["theta"]
        gk = parameter_history[idx, k - 1].astype(np.float64)
        sr, sg = np.linalg.norm(rk - prev, axis=1), np.linalg.norm(gk - prev_g, axis=1)
        differs |= ~differs & (np.abs(sg - sr) > 0.3 * sr)
        prev, prev_g = rk, gk
    return differs


def _rel(a, ref):
    return np.linalg.norm(a - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)


WELL_DETERMINED = {
    # name: (variant, batch, share of the instances that must be within 1e-5, bound on the rest)
    # -- position + orientation constraint on every joint
    # n = 128, the fused instantiation NB = 8.  Finger curls, fists and the spine twist are SHARED parameters here.  Rounds
    # 2-3 held this variant to 99.5 % within 1e-5 / all within 2e-5 (median 1.5e-6 where the oracle's float instantiation has
    # 0.7e-6, tail 0.7 ... 1.4e-5) and blamed the world-origin moments of the tree kernels; round 4 measured the cause
    # (scripts/diag_step_noise.py, diag_g_stages.py): the re-associated single-precision products of the pointer-jumping
    # forward kinematics.  With those in double (mmx_device.hpp fkJumpRoundsD) the median is 0.67e-6, the worst of 1024
    # 2.9e-6: the plain bound on every instance.
    "p128_all_joints": ("p128", 1024, 1.0, 1e-5),
    # n = 219: BASELINE's stress variant cfg2_all (three rotations per joint, nothing shared): the plain bound on every instance
    "p219_all_joints": ("p219", 512, 1.0, 1e-5),
}


@pytest.mark.parametrize("route", ["fused", "wide"])
@pytest.mark.parametrize("line_search", [0, 2])
@pytest.mark.parametrize("name", sorted(WELL_DETERMINED))
def test_well_determined_problems_hold_1e5_at_every_lambda(torch_cuda, orc, name, line_search, route):
    variant, B, share, rest_bound = WELL_DETERMINED[name]
    rig = make_humanoid72(seed=12345, variant=variant, unit=UNIT)
    allj = list(range(rig.num_joints))
    cons, th0, _ = make_problem(rig, allj, allj, B, seed=31337, perturb=0.3)
    for lam in LAMBDAS:
        opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=line_search)
        out = _gpu_solve(torch_cuda, rig, cons, th0, opt, route, parameter_history=bool(line_search))
        assert out["route_taken"] == route
        ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores())
        rel = _rel(out["theta"].astype(np.float64), ref["theta"])
        # A backtracking line search takes discrete decisions (accept alpha or halve it) on a difference of two errors: an
        # instance whose accept test sits on its threshold goes one way in single and the other in double precision -- the
        # oracle's own float instantiation does, too -- and is then a different (equally valid) iteration, not a rounding
        # error.  Same decisions <=> the same error at every iterate (1e-3 relative, above the fp32 noise floor of a
        # converged fit): those instances are held to 1e-5; the few others must have converged as far as the double run.
        # The history holds the error at the START of each iteration, so the decision of the last iteration shows only in
        # an eleventh entry: the same (deterministic) solves run for eleven iterations supply it -- a nearly converged
        # instance whose last step is 1e-5 of theta moves by 0.5e-5 when that step is halved.
        same = np.ones(B, bool)
        if line_search:
            opt11 = GnOptions.make(min_iterations=11, max_iterations=11, threshold=1.0, regularization=lam, do_line_search=line_search)
            h = _gpu_solve(torch_cuda, rig, cons, th0, opt11, route)["error_history"]
            href = orc.solve_batch(rig, cons, th0, opt11, dtype="f64", nthreads=_cores())["error_history"]
            assert np.array_equal(h[:, :10], out["error_history"])
            same = np.all(np.abs(h - href) <= 1e-3 * np.abs(href) + 1e-7 * href[:, :1], axis=1)
            # ... and a decision taken on a decrease the error history cannot resolve: checked on the (few) instances it matters for
            idx = np.flatnonzero(same & (rel > BOUND))
            if 0 < len(idx) <= 32:
                opt_of = lambda k: GnOptions.make(min_iterations=k, max_iterations=k, threshold=1.0, regularization=lam, do_line_search=line_search)
                same[idx[_backtracking_differs(orc, rig, cons, th0, opt_of, idx, out["parameter_history"])]] = False
        _REPORT[f"{name} lambda={lam:g} line_search={line_search} route={route}"] = {
            "instances": B, "max_rel": float(rel.max()), "median_rel": float(np.median(rel)), "above_1e-5": int((rel > BOUND).sum()),
            "same_line_search_decisions": int(same.sum()), "max_rel_same_decisions": float(rel[same].max()),
            "bound": "1e-5 on every instance whose line-search decisions agree with the double run (all of them without a line search)"}  # fmt: skip
        _write_report()
        assert np.all(out["status"] & 3 == 0) and np.array_equal(out["iterations"], ref["iterations"])
        assert same.mean() >= 0.99, (name, lam, line_search, route, float(same.mean()))
        # north_star: 1e-5 relative on pose parameters
        assert (rel[same] <= BOUND).mean() >= share and rel[same].max() <= rest_bound, (name, lam, line_search, route, float(rel[same].max()), int((rel[same] > BOUND).sum()))
        assert np.all(rel[~same] <= 1e-3)  # (a different branch, the same minimum)
        # the value solve() returns (error at the parameters before the last step): the double solve is at its 1e-12 floor there
        assert np.all(out["error"] <= 1.001 * ref["error"] + 1e-9)


def _xf_point(w, p):
    from tests.helpers import quat_rot

    return w[:3] + quat_rot(w[3:7], w[7] * np.asarray(p, np.float64))


@pytest.mark.parametrize("route", ["fused", "wide"])
def test_reference_three_joint_ik_known_answer(torch_cuda, orc, route):
    """inverse_kinematics_test.cpp:60-121 through the C ABI: 3-joint chain, one position constraint on joint 2 (offset
    UnitY), GaussNewtonSolver{min = max = 6 iterations, lambda = 1e-7}; the rest target keeps theta at 0; ten random
    targets in [-3,3]^3 are reached: error <= 5e-7, end effector within 5e-5 (the reference's float tolerances)."""
    rig = make_test_character(3)
    P = rig.num_params
    rng = np.random.default_rng(12345)
    B = 11
    off = np.tile(np.array([[[0, 1, 0]]], np.float32), (B, 1, 1))
    tgt = rng.uniform(-3, 3, size=(B, 1, 3)).astype(np.float32)
    tgt[0, 0] = [0, 3, 0]  # the rest pose's end effector
    cons = orc.Constraints([2], off, tgt, np.ones((B, 1), np.float32), [], np.zeros((B, 0, 4)), np.zeros((B, 0, 4)), np.zeros((B, 0)))
    opt = GnOptions.make(min_iterations=6, max_iterations=6, threshold=1.0, regularization=1e-7)
    th0 = np.zeros((B, P), np.float32)
    out = _gpu_solve(torch_cuda, rig, cons, th0, opt, route)
    assert out["route_taken"] == route
    th = out["theta"]
    assert np.all(np.isfinite(th))
    assert np.abs(th[0]).max() <= 1e-6  # :96-99
    worst_e, worst_p = 0.0, 0.0
    for b in range(B):
        final = orc.get_error(rig, cons.instance(b), th[b].astype(np.float64), "f64")
        w = orc.skeleton_state(rig, th[b].astype(np.float64), "f64")["world"][2].astype(np.float64)
        worst_e, worst_p = max(worst_e, final), max(worst_p, np.abs(_xf_point(w, [0, 1, 0]) - tgt[b, 0]).max())
    _REPORT[f"ik3 lambda=1e-07 route={route}"] = {"instances": B, "max_final_error": worst_e, "max_end_effector_miss": worst_p, "bound": "error 5e-7, end effector 5e-5 (the reference's float tolerances)"}  # fmt: skip
    _write_report()
    assert worst_e <= 5e-7 and worst_p <= 5e-5, (worst_e, worst_p)  # :114-121 (float instantiation)


BASELINE_SHAPES = {
    # name: (rig, pos parents, ori parents, batch, start perturbation)
    "cfg1": (lambda: make_test_character(24), [23, 12, 5], [], 1024, 0.3),
    "cfg2": (lambda: make_humanoid72(seed=12345, variant="p128", unit=UNIT), "lm", "lm", 1024, 0.3),
}


@pytest.mark.parametrize("route", ["fused", "wide"])
@pytest.mark.parametrize("line_search", [0, 2])
@pytest.mark.parametrize("name", sorted(BASELINE_SHAPES))
def test_baseline_shapes_at_weak_damping(torch_cuda, orc, name, line_search, route):
    mk, pp, op, B, perturb = BASELINE_SHAPES[name]
    rig = mk()
    if pp == "lm":
        pp = op = humanoid72_landmark_joints(rig)
    cons, th0, _ = make_problem(rig, pp, op, B, seed=777, perturb=perturb)
    for lam in LAMBDAS:
        opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=line_search)
        out = _gpu_solve(torch_cuda, rig, cons, th0, opt, route)
        assert out["route_taken"] == route
        with np.errstate(all="ignore"):
            r64 = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores())
            r32 = orc.solve_batch(rig, cons, th0, opt, dtype="f32", nthreads=_cores())
            e0 = np.array([orc.get_error(rig, cons.instance(b), th0[b].astype(np.float64), "f64") for b in range(0, B, 64)]).max()
            # Without a line search an undamped Gauss-Newton step overshoots on these shapes and the DOUBLE solve itself diverges
            # (errors of 1e50 and more, non-finite parameters): there is no answer to compare with; such instances are counted,
            # not compared (with the directional line search of the batched driver every instance converges in double).
            sane = (r64["status"] == 0) & np.isfinite(r64["theta"]).all(axis=1) & (r64["error"] <= e0)
            rel = _rel(out["theta"].astype(np.float64), r64["theta"])
            rel32 = _rel(r32["theta"].astype(np.float64), r64["theta"])
        within = sane & (rel <= BOUND)
        above = sane & ~(rel <= BOUND)
        row = {
            "instances": B,
            "double_solve_converged": int(sane.sum()),
            "within_1e-5": int(within.sum()),
            "above_1e-5": int(above.sum()),
            "above_and_float_oracle_also_above": int((above & ~(rel32 <= BOUND)).sum()),
            "median_rel_hip": float(np.median(rel[sane])) if sane.any() else None,
            "median_rel_float_oracle": float(np.median(rel32[sane])) if sane.any() else None,
            "p90_rel_hip": float(np.quantile(rel[sane], 0.9)) if sane.any() else None,
            "p90_rel_float_oracle": float(np.quantile(rel32[sane], 0.9)) if sane.any() else None,
            "status_hip_nonzero": int((out["status"] & 3 != 0).sum()), "status_hip_damping_floored": int((out["status"] & 4 != 0).sum()),
            "status_float_oracle_nonzero": int((r32["status"] != 0).sum()),
            "max_final_error_hip": float(np.nanmax(np.where(sane, out["error"], 0.0))),
            "max_final_error_double": float(np.nanmax(np.where(sane, r64["error"], 0.0))),
        }
        _REPORT[f"{name} lambda={lam:g} line_search={line_search} route={route}"] = row
        _write_report()
        if not sane.any():
            continue
        # (1) every instance above the bound is one where the reference's float instantiation is above it too
        assert row["above_and_float_oracle_also_above"] == row["above_1e-5"], (name, lam, line_search, route, row)
        # (2) whatever the damping, the solve returns finite parameters (a step is always taken, never a non-finite one)
        assert np.isfinite(out["theta"]).all()
        # (3) the objective: fp32 normal equations square the condition number of J, so where J is rank deficient or nearly
        #     (these shapes) and lambda is below the rounding of J^T J, NO single-precision Cholesky solver reaches the
        #     double solver's minimum -- the reference's float instantiation stalls the same way (its LLT aborts:
        #     status_float_oracle_nonzero).  The HIP path must do no worse than that float instantiation does: median of the
        #     final error within 2x, 90th percentile within 4x (two float runs of a chaotic iteration; measured: the chain
        #     of cfg1 ends 60x ... 1e11x LOWER than the float instantiation, cfg2 at lambda = 1e-7 at 1.2x / 2.5x)
        #     (+ the reference's own cross-solver slack, solver_test.cpp:
        #     110-118), and where the float instantiation itself is sound (no aborted LLT anywhere: lambda >= 1e-5 with the
        #     line search) against the DOUBLE run as well, in distribution (median and 90th percentile within 2x).
        eh, e32, e64 = out["error"][sane], r32["error"][sane], r64["error"][sane]
        with np.errstate(all="ignore"):
            row["nonfinite_final_error"] = {"hip": int((~np.isfinite(eh)).sum()), "float_oracle": int((~np.isfinite(e32)).sum())}
            row["median_final_error"] = {"hip": float(np.nanmedian(eh)), "float_oracle": float(np.nanmedian(e32)), "double": float(np.median(e64))}
            row["p90_final_error"] = {"hip": float(np.nanquantile(eh, 0.9)), "float_oracle": float(np.nanquantile(e32, 0.9)), "double": float(np.quantile(e64, 0.9))}
        _write_report()
        if line_search:  # (without one the undamped iteration is chaotic in every precision: reported, not asserted)
            assert row["nonfinite_final_error"]["hip"] <= max(B // 50, 2 * row["nonfinite_final_error"]["float_oracle"]), (name, lam, line_search, route, row)
            assert row["median_final_error"]["hip"] <= 2.0 * row["median_final_error"]["float_oracle"] + 1e-3, (name, lam, line_search, route, row)
            assert row["p90_final_error"]["hip"] <= 4.0 * row["p90_final_error"]["float_oracle"] + 1e-3, (name, lam, line_search, route, row)
            if row["status_float_oracle_nonzero"] == 0:
                # (per instance the three runs part ways here: a line-search decision on its threshold sends an instance down
                # another branch, and ~10 % of the instances end in a local fit with a final error of O(1) in EVERY precision --
                # measured p90: double 3.9, float oracle 0.6, HIP 1.3 -- so the double run is compared in distribution too)
                assert row["median_final_error"]["hip"] <= 2.0 * row["median_final_error"]["double"] + 1e-3, (name, lam, line_search, route, row)
                assert row["p90_final_error"]["hip"] <= 2.0 * max(row["p90_final_error"]["double"], row["p90_final_error"]["float_oracle"]) + 1e-3, (name, lam, line_search, route, row)


def test_solve_ik_defaults_full_batch(torch_cuda, orc):
    """solve_ik's real defaults (pymomentum/tensor_ik/solver_options.h:28-37: lambda 0.01, min 4 / max 50 iterations,
    threshold 10, line search on -- the directional rule of the two solvers the driver builds) on BASELINE configs[1]'s
    batch, 4096 distinct instances.  Instances stop at their own iteration; a stop one iteration apart moves theta by the
    last (converged) step, so poses are compared through the joint world positions like test_solver2.py:135-200, and
    pose parameters at the bound on the instances that stopped at the same iteration."""
    import bench

    torch = torch_cuda
    rig, parents, _, _, _ = bench.build_rig("cfg2")
    B, n = 4096, 1024
    db = bench.DeviceBatch(rig, parents, B, 0, 20260926)
    opt = GnOptions.make(min_iterations=4, max_iterations=50, threshold=10.0, regularization=0.01, do_line_search=2)
    out = db.pb.solve(db.theta0.clone(), opt)
    torch.cuda.synchronize()
    assert int((out["status"] & 3 != 0).sum()) == 0
    it = out["iterations"].cpu().numpy()
    assert it.min() >= 5 and it.max() <= 50
    cons = db.host_constraints(n)
    ref = orc.solve_batch(rig, cons, np.zeros((n, rig.num_params), np.float32), opt, dtype="f64", nthreads=_cores())
    assert np.abs(it[:n] - ref["iterations"]).max() <= 3
    th = out["theta"][:n].cpu().numpy().astype(np.float64)
    rel = _rel(th, ref["theta"])
    same = it[:n] == ref["iterations"]
    st = db.pb.skeleton_state(out["theta"])[:n].cpu().numpy()
    miss = 0.0
    for b in range(0, n, 8):
        sref = orc.skeleton_state(rig, ref["theta"][b], "f64")["world"]
        miss = max(miss, float(np.abs(st[b][:, :3] - sref[:, :3]).max()))
    _REPORT["solve_ik defaults cfg2@4096"] = {"instances_checked": n, "same_stop_iteration": int(same.sum()), "max_rel_same_stop": float(rel[same].max()),
                                              "max_rel_all": float(rel.max()), "max_joint_position_miss": miss, "iterations_min_max": [int(it.min()), int(it.max())]}  # fmt: skip
    _write_report()
    assert same.mean() >= 0.5
    assert miss <= 1e-4
    e, eref = out["error"][:n].cpu().numpy(), ref["error"]
    assert np.all(np.abs(e - eref) <= 1e-3 * eref + 1e-9)


@pytest.mark.parametrize("route", ["fused", "wide"])
def test_solve_matches_the_qr_solver(torch_cuda, orc, route):
    """`solve_ik`'s DEFAULT linear solver is GaussNewtonSolverQRT (pymomentum/tensor_ik/solver_options.h:28-37,
    tensor_ik.cpp:142-158): Householder QR of [J; sqrt(lambda) I] (gauss_newton_solver_qr.cpp:50-150).  mmx_solve factors
    the regularised normal equations instead -- the same least-squares problem.  The oracle restates the QR iteration
    (tests/test_oracle_golden.py pins it against the Cholesky solver by the reference's own criterion); here the HIP path
    is held to 1e-5 of THAT solver's double run, at solve_ik's default lambda and at test_solver2.py's 1e-5, with and
    without the (directional) line search."""
    rig = make_humanoid72(seed=12345, variant="p128", unit=UNIT)
    allj = list(range(rig.num_joints))
    B = 256
    cons, th0, _ = make_problem(rig, allj, allj, B, seed=4711, perturb=0.3)
    for lam in (0.01, 1e-5):
        for ls in (0, 2):
            opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=ls)
            out = _gpu_solve(torch_cuda, rig, cons, th0, opt, route)
            ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores(), use_qr=True)
            rel = _rel(out["theta"].astype(np.float64), ref["theta"])
            _REPORT[f"vs GaussNewtonSolverQRT: p128_all_joints lambda={lam:g} line_search={ls} route={route}"] = {
                "instances": B, "max_rel": float(rel.max()), "median_rel": float(np.median(rel)), "bound": "1e-5, no escape"}  # fmt: skip
            _write_report()
            assert rel.max() <= BOUND, (lam, ls, route, float(rel.max()))
            assert np.array_equal(out["iterations"], ref["iterations"]) and np.all(out["status"] & 3 == 0)


@pytest.mark.parametrize("name", sorted(BASELINE_SHAPES))
def test_double_instantiation_follows_the_double_solver_at_weak_damping(torch_cuda, orc, name):
    """Where single precision stalls (rank-deficient J, lambda below the rounding of J^T J: the table above) the
    library's answer is its double instantiation, mmx_solve_f64 = SolverT<double> (gauss_newton_solver.cpp:315-316): on
    the same under-determined shapes, at lambda = 1e-7 and 1e-5 with the driver's line search, it reaches the double
    oracle's minimum (the reference's cross-solver criterion on every instance) and its pose parameters on the instances
    whose line-search decisions agree (an under-determined minimiser amplifies a last-bit difference, so the bound is the
    north_star's 1e-5 rather than the 1e-10 the well-posed f64 tests hold; measured 2e-6)."""
    from momentum_amd import capi

    torch = torch_cuda
    mk, pp, op, B, perturb = BASELINE_SHAPES[name]
    B = 256
    rig = mk()
    if pp == "lm":
        pp = op = humanoid72_landmark_joints(rig)
    cons, th0, _ = make_problem(rig, pp, op, B, seed=777, perturb=perturb)
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))  # fmt: skip
    for lam in (1e-7, 1e-5):
        opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=2)
        out = pb.solve_f64(torch.from_numpy(th0.astype(np.float64)).to(pb.device), opt, want_history=True)
        torch.cuda.synchronize()
        ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=_cores())
        th, e, h = out["theta"].cpu().numpy(), out["error"].cpu().numpy(), out["error_history"].cpu().numpy()
        href = ref["error_history"]
        same = np.all(np.abs(h - href) <= 1e-6 * np.abs(href) + 1e-12 * href[:, :1], axis=1)
        rel = _rel(th, ref["theta"])
        _REPORT[f"mmx_solve_f64: {name} lambda={lam:g} line_search=2"] = {
            "instances": B, "same_line_search_decisions": int(same.sum()), "max_rel_same_decisions": float(rel[same].max()) if same.any() else None,
            "max_final_error": float(e.max()), "max_final_error_double_oracle": float(ref["error"].max())}  # fmt: skip
        _write_report()
        assert np.all(out["status"].cpu().numpy() & 3 == 0) and np.isfinite(th).all()
        # (an under-determined minimiser at lambda = 1e-7 amplifies a last-bit difference into another line-search branch on
        # many instances -- reported; the objective is what both reach)
        # (the oracle is built per host with the fastest of four flag sets -- another contraction of a*b+c is another last
        # bit of ITS double run, and on these under-determined shapes that can put a borderline instance on either side of
        # `same`: 97 % of the instances that agree, not the single worst one, carry the bound)
        assert same.sum() >= B // 8 and np.quantile(rel[same], 0.97) <= BOUND, (int(same.sum()), float(rel[same].max()))
        # (the final error is one step past the last history entry `same` compares at 1e-6: an order of magnitude of room)
        assert np.quantile(np.abs(e[same] - ref["error"][same]) / (ref["error"][same] + 1e-12), 0.97) <= 1e-5
        # every instance reaches a minimum of the same quality as the oracle's run of it (another branch, another local fit:
        # compared in distribution; same branch: compared above)
        assert np.median(e) <= 1.01 * np.median(ref["error"]) + 1e-3 and np.quantile(e, 0.9) <= 1.1 * np.quantile(ref["error"], 0.9) + 1e-3
