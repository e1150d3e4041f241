"""Oracle checks for the parameter-space error functions, re-expressing the
reference's own tests: LimitError_GradientsAndJacobians
(momentum/test/character_solver/limit_error_function_test.cpp:27-290: MinMax, Linear, piecewise
Linear, HalfPlane fixtures) and ModelParametersError_GradientsAndJacobians
(momentum/test/character_solver/state_error_function_test.py ... state_error_function_test.cpp:88-118)
through TEST_GRADIENT_AND_JACOBIAN = Jacobian vs finite differences, |r|^2 == error,
2 J^T r == gradient (error_function_helpers.cpp:169-281)."""
import numpy as np
import pytest

from momentum_amd import make_test_character
from momentum_amd._abi import ParameterLimit
from oracle import oracle as orc

FLT_MAX = float(np.finfo(np.float32).max)


def _cons(rig, limits=None, mp=None, wl=1.0, wm=1.0):
    P = rig.num_params
    z = np.zeros
    return orc.Constraints(
        z(0, np.int32), z((0, 3)), z((0, 3)), z(0), z(0, np.int32), z((0, 4)), z((0, 4)), z(0),
        limits=limits, limit_function_weight=wl,
        model_target=None if mp is None else mp[0], model_weights=None if mp is None else mp[1], model_function_weight=wm,
    )


def _check(rig, cons, theta, fd_tol=1e-6, enabled=None):
    J, r, err = orc.eval_jacobian(rig, cons, theta, enabled=enabled, dtype="f64")
    assert J.shape == (cons.rows, rig.num_params)
    assert abs(r @ r - err) <= 1e-7 * max(1.0, err)  # sWeight of the model-parameter rows is a float in the reference (:109)
    ge = orc.get_error(rig, cons, theta, "f64") if enabled is None else None
    if ge is not None:
        assert abs(ge - err) <= 2e-6 * max(1.0, err)  # getError is rounded through float
    h = 1e-7
    g = 2 * J.T @ r
    for p in range(rig.num_params):
        if enabled is not None and not enabled[p]:
            assert np.all(J[:, p] == 0)
            continue
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h
        tm[p] -= h
        _, rp, ep = orc.eval_jacobian(rig, cons, tp, enabled=enabled, dtype="f64")
        _, rm, em = orc.eval_jacobian(rig, cons, tm, enabled=enabled, dtype="f64")
        assert np.abs((rp - rm) / (2 * h) - J[:, p]).max() <= fd_tol * max(1.0, np.abs(J).max())
        assert abs((ep - em) / (2 * h) - g[p]) <= 1e-5 * max(1.0, np.abs(g).max())
    J32, r32, e32 = orc.eval_jacobian(rig, cons, theta.astype(np.float32), enabled=enabled, dtype="f32")
    assert np.abs(J32 - J).max() <= 1e-5 * max(1.0, np.abs(J).max())
    assert np.abs(r32 - r).max() <= 1e-5 * max(1.0, np.abs(r).max())
    return J, r, err


def test_limit_minmax():
    rig = make_test_character(5)
    cons = _cons(rig, [ParameterLimit.minmax(0, -0.1, 0.1, 1.0)])
    J, r, err = _check(rig, cons, np.zeros(rig.num_params))
    assert err == 0 and not J.any()
    rng = np.random.default_rng(1)
    for _ in range(10):
        theta = rng.uniform(-1, 1, rig.num_params)
        J, r, err = _check(rig, cons, theta)
        v = theta[0] - np.clip(theta[0], -0.1, 0.1)
        # kLimitWeight = 10 (limit_error_function.h:91): error = 10 * w * val^2, row = sqrt(10 w) * (val, 1)
        assert err == pytest.approx(10.0 * v * v, rel=1e-6)
        if v != 0:
            assert J[0, 0] == pytest.approx(np.sqrt(10.0), rel=1e-6) and r[0] == pytest.approx(np.sqrt(10.0) * v, rel=1e-6)


def test_limit_linear_and_piecewise():
    rig = make_test_character(5)
    cons = _cons(rig, [ParameterLimit.linear(0, 5, 0.25, 0.25, weight=1.5)])
    _check(rig, cons, np.zeros(rig.num_params))
    rng = np.random.default_rng(2)
    for _ in range(10):
        theta = rng.uniform(-1, 1, rig.num_params)
        J, r, err = _check(rig, cons, theta)
        res = theta[5] * 0.25 - 0.25 - theta[0]
        assert err == pytest.approx(10.0 * 1.5 * res * res, rel=1e-6)
    # piecewise: |p5 + 3| on both sides of -3, C0-continuous error (limit_error_function_test.cpp:105-168)
    lims = [
        ParameterLimit.linear(0, 5, -1.0, 3.0, -FLT_MAX, -3.0, weight=0.5),
        ParameterLimit.linear(0, 5, 1.0, -3.0, -3.0, FLT_MAX, weight=0.5),
    ]
    cons = _cons(rig, lims)
    errs = []
    for v in (-3.01, -3.0, -2.99):
        theta = np.zeros(rig.num_params)
        theta[5] = np.float32(v)
        if v != -3.0:
            _check(rig, cons, theta, fd_tol=1e-5)
        errs.append(orc.get_error(rig, cons, theta, "f64"))
    assert abs(errs[0] - errs[1]) < 0.03 and abs(errs[1] - errs[2]) < 0.03


def test_limit_halfplane():
    rig = make_test_character(5)
    n = np.array([1.0, -1.0]) / np.sqrt(2.0)
    cons = _cons(rig, [ParameterLimit.halfplane(0, 2, n[0], n[1], 0.5)])
    rng = np.random.default_rng(3)
    hit = 0
    for _ in range(12):
        theta = rng.uniform(-1, 1, rig.num_params)
        J, r, err = _check(rig, cons, theta)
        res = np.float32(n[0]) * theta[0] + np.float32(n[1]) * theta[2] - 0.5
        if res < 0:
            hit += 1
            assert err == pytest.approx(10.0 * res * res, rel=1e-5)
        else:
            assert err == 0 and not J.any()
    assert hit > 0


def test_model_parameters_error():
    # state_error_function_test.cpp:96-117: weights = ones, w[0]=4, w[1]=5, w[2]=0, targets 0
    rig = make_test_character(5)
    P = rig.num_params
    w = np.ones(P)
    w[:3] = [4.0, 5.0, 0.0]
    cons = _cons(rig, mp=(np.zeros(P), w))
    _check(rig, cons, np.zeros(P))
    rng = np.random.default_rng(4)
    for _ in range(10):
        theta = 0.25 * rng.uniform(-1, 1, P)
        J, r, err = _check(rig, cons, theta)
        assert err == pytest.approx(0.1 * np.sum((w * theta) ** 2), rel=1e-6)  # kMotionWeight = 0.1
        # rows are compacted over the parameters with weight > 0 (getJacobian :113-121)
        used = [i for i in range(P) if w[i] > 0]
        for out, i in enumerate(used):
            assert J[out, i] == pytest.approx(np.sqrt(np.float32(0.1)) * w[i], rel=1e-6)
        assert not J[len(used):].any()


def test_blocks_combine_with_joint_constraints_and_enabled_set():
    from tests.helpers import make_problem

    rig = make_test_character(6)
    P = rig.num_params
    cons, th0, ths = make_problem(rig, [5, 2], [4], 1, seed=21, random_offsets=True)
    c = cons.instance(0)
    rng = np.random.default_rng(5)
    full = orc.Constraints(
        c.pos_parent, c.pos_offset, c.pos_target, c.pos_weight, c.ori_parent, c.ori_offset, c.ori_target, c.ori_weight,
        limits=[ParameterLimit.minmax(3, -0.05, 0.05, 2.0), ParameterLimit.linear(1, 4, 0.5, 0.1), ParameterLimit.halfplane(6, 7, 0.6, 0.8, 0.2)],
        limit_function_weight=0.7,
        model_target=rng.uniform(-0.2, 0.2, P), model_weights=rng.uniform(0.0, 2.0, P), model_function_weight=1.3,
    )
    assert full.rows == 3 * 2 + 9 + 3 + P
    theta = rng.uniform(-0.5, 0.5, P)
    en = np.ones(P, np.uint8)
    en[[1, 7]] = 0
    _check(rig, full, theta)
    _check(rig, full, theta, enabled=en)
    # a solve with the extra blocks decreases the total error monotonically from the first step on
    from momentum_amd._abi import GnOptions

    opt = GnOptions.make(min_iterations=8, max_iterations=8, regularization=0.05)
    out = orc.solve(rig, full, np.zeros(P), opt, dtype="f64")
    hist = out["error_history"]
    assert np.all(np.diff(hist) <= 1e-9) and hist[-1] < hist[0]


def test_limit_minmax_joint_and_linear_joint():
    # limit_error_function_test.cpp:62-81 (MinMaxJoint on joint 2, parameter 5, limits +-0.1) and
    # :170-196 (piecewise LinearJoint between joint 0 tz and joint 1 rz, driven by shared_rz)
    rig = make_test_character(5)
    cons = _cons(rig, [ParameterLimit.minmax_joint(2, 5, -0.1, 0.1, 1.0)])
    _check(rig, cons, np.zeros(rig.num_params))
    rng = np.random.default_rng(11)
    hits = 0
    for _ in range(10):
        theta = rng.uniform(-1, 1, rig.num_params)
        J, r, err = _check(rig, cons, theta)
        hits += int(err > 0)
    assert hits > 0
    lims = [
        ParameterLimit.linear_joint(0, 2, 1, 5, 1.0, -4.0, -FLT_MAX, 0.0, weight=0.75),
        ParameterLimit.linear_joint(0, 2, 1, 5, -1.0, -4.0, 0.0, 2.0, weight=0.75),
        ParameterLimit.linear_joint(0, 2, 1, 5, 1.0, 0.0, 2.0, FLT_MAX, weight=0.75),
    ]
    cons = _cons(rig, lims)
    rz = rig.param_names.index("shared_rz")
    for test_pos in (-4.0, 0.0, 4.0):
        errs = []
        for d in (-0.001, 0.001):
            theta = np.zeros(rig.num_params)
            theta[rz] = np.float32(test_pos + d)
            _check(rig, cons, theta, fd_tol=1e-5)
            errs.append(orc.get_error(rig, cons, theta, "f64"))
        assert abs(errs[0] - errs[1]) < 0.03  # C0 continuity across the pieces (:216-221)
    # a joint limit writes every column of the transform row, enabled or not
    # (jacobian_jointParams_to_modelParams, error_function_utils.h:77-91), and is skipped only when
    # no column of the row is enabled
    theta = rng.uniform(-1, 1, rig.num_params)
    cons = _cons(rig, [ParameterLimit.minmax_joint(1, 5, -0.01, 0.01, 1.0)])
    J, r, err = orc.eval_jacobian(rig, cons, theta, dtype="f64")
    cols = np.flatnonzero(J[0])
    assert len(cols) >= 1
    en = np.ones(rig.num_params, np.uint8)
    en[cols[0]] = 0
    J2, _, _ = orc.eval_jacobian(rig, cons, theta, enabled=en, dtype="f64")
    if len(cols) > 1:
        assert np.array_equal(J2, J)
    else:
        assert not J2.any()
