"""Oracle checks for the further JointErrorFunctionT specialisations,
re-expressing the reference's own tests: PlaneErrorL2 / HalfPlaneErrorL2_GradientsAndJacobians
(momentum/test/character_solver/plane_error_function_test.cpp:27-108), AimDist / AimDir
(aim_error_function_test.cpp), FixedAxisDiff / Cos / Angle (fixed_axis_error_function_test.cpp) and
NormalError (normal_error_function_test.cpp): createTestCharacter(), two constraints with weight
4.5 on joints 2 and 1, theta = 0 and ten theta ~ U[-1,1]^P, through TEST_GRADIENT_AND_JACOBIAN =
Jacobian vs finite differences, |r|^2 == error, 2 J^T r == gradient
(error_function_helpers.cpp:169-281)."""
import numpy as np
import pytest

from momentum_amd import make_test_character
from momentum_amd import _abi
from momentum_amd._abi import JointBlock
from oracle import oracle as orc

TYPES = {
    "plane": _abi.MMX_JC_PLANE,
    "half_plane": _abi.MMX_JC_HALF_PLANE,
    "aim_dist": _abi.MMX_JC_AIM_DIST,
    "aim_dir": _abi.MMX_JC_AIM_DIR,
    "fixed_axis_diff": _abi.MMX_JC_FIXED_AXIS_DIFF,
    "fixed_axis_cos": _abi.MMX_JC_FIXED_AXIS_COS,
    "fixed_axis_angle": _abi.MMX_JC_FIXED_AXIS_ANGLE,
    "normal": _abi.MMX_JC_NORMAL,
}


def make_block(type_, parents, rng, weight=4.5, batch=None, function_weight=1.0, loss=(2.0, 1.0)):
    """Random payload shaped like the reference tests' fixtures (uniform(0,1) points, uniform(0.1,1) directions)."""
    K = len(parents)
    shp = (K,) if batch is None else (batch, K)
    u = lambda lo, hi, *d: rng.uniform(lo, hi, shp + d).astype(np.float32)
    kw = dict(weight=np.full(shp, weight, np.float32), function_weight=function_weight, loss=loss)
    if type_ in (_abi.MMX_JC_PLANE, _abi.MMX_JC_HALF_PLANE):
        return JointBlock(type_, parents, global_=u(0.1, 1, 3), local_point=u(0, 1, 3), plane_d=u(0, 1), **kw)
    if type_ in (_abi.MMX_JC_AIM_DIST, _abi.MMX_JC_AIM_DIR):
        return JointBlock(type_, parents, global_=u(0, 1, 3) + 2.0, local_point=u(0, 1, 3), local_dir=u(0.1, 1, 3), **kw)
    if type_ == _abi.MMX_JC_NORMAL:
        return JointBlock(type_, parents, global_=u(0, 1, 3), local_point=u(0, 1, 3), local_dir=u(0.1, 1, 3), **kw)
    return JointBlock(type_, parents, global_=u(0.1, 1, 3), local_dir=u(0.1, 1, 3), **kw)


def cons_with(blocks, Kp=0):
    z = np.zeros
    return orc.Constraints(z(0, np.int32), z((0, 3)), z((0, 3)), z(0), z(0, np.int32), z((0, 4)), z((0, 4)), z(0), joint_blocks=blocks)


def check(rig, cons, theta, fd_tol=2e-6, enabled=None):
    J, r, err = orc.eval_jacobian(rig, cons, theta, enabled=enabled, dtype="f64")
    assert J.shape == (cons.rows, rig.num_params)
    assert abs(r @ r - err) <= 1e-9 * max(1.0, err)  # L2: |r|^2 == error (error_function_helpers.cpp:214-217)
    if enabled is None:
        assert abs(orc.get_error(rig, cons, theta, "f64") - err) <= 2e-6 * max(1.0, err)  # getError rounds through float
    h = 1e-6
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
        assert np.abs((rp - rm) / (2 * h) - J[:, p]).max() <= fd_tol * max(1.0, np.abs(J).max()), p
        assert abs((ep - em) / (2 * h) - g[p]) <= 1e-5 * max(1.0, np.abs(g).max())
    J32, r32, _ = orc.eval_jacobian(rig, cons, theta.astype(np.float32), enabled=enabled, dtype="f32")
    assert np.abs(J32 - J).max() <= 2e-4 * max(1.0, np.abs(J).max())  # the f32 instantiation (acos / 1/sin amplify rounding)
    assert np.abs(r32 - r).max() <= 2e-4 * max(1.0, np.abs(r).max())
    return J, r, err


@pytest.mark.parametrize("name", list(TYPES))
def test_gradients_and_jacobians(name):
    rig = make_test_character(5)
    rng = np.random.default_rng(12345)
    blk = make_block(TYPES[name], [2, 1], rng)
    cons = cons_with([blk])
    assert cons.rows == (6 if name in ("aim_dist", "aim_dir", "fixed_axis_diff") else 2)
    check(rig, cons, np.zeros(rig.num_params))
    for _ in range(10):
        check(rig, cons, rng.uniform(-1, 1, rig.num_params))


def test_half_plane_is_one_sided():
    """PlaneErrorFunctionT(above=true): zero residual and zero Jacobian on the positive side (:63-70)."""
    rig = make_test_character(5)
    rng = np.random.default_rng(3)
    theta = rng.uniform(-0.5, 0.5, rig.num_params)
    pt = np.zeros((1, 3), np.float32)
    n = np.array([[0.0, 1.0, 0.0]], np.float32)
    w = np.ones(1, np.float32)
    st = orc.skeleton_state(rig, theta)["world"]
    y = st[3, 1]
    for d, inside in ((y - 0.5, True), (y + 0.5, False)):  # plane below / above the joint
        blk = JointBlock(_abi.MMX_JC_HALF_PLANE, [3], w, n, local_point=pt, plane_d=np.array([d], np.float32))
        J, r, err = orc.eval_jacobian(rig, cons_with([blk]), theta)
        if inside:
            assert err == 0 and not J.any() and not r.any()
        else:
            assert r[0] == pytest.approx(-0.5, abs=1e-6) and np.abs(J).max() > 0
        full = JointBlock(_abi.MMX_JC_PLANE, [3], w, n, local_point=pt, plane_d=np.array([d], np.float32))
        _, rf, _ = orc.eval_jacobian(rig, cons_with([full]), theta)
        assert rf[0] == pytest.approx(0.5 if inside else -0.5, abs=1e-6)


def test_mixed_blocks_rows_weights_and_disabled_parameters():
    """Several blocks + position/orientation constraints: block order fixes the row layout; a block
    with weight_ <= 0 and a constraint with weight 0 keep zero rows; disabled parameters zero columns."""
    rig = make_test_character(8)
    rng = np.random.default_rng(7)
    P = rig.num_params
    blocks = [
        make_block(_abi.MMX_JC_PLANE, [7, 3, 5], rng, weight=2.0),
        make_block(_abi.MMX_JC_AIM_DIR, [6], rng, weight=1.5, function_weight=0.7),
        make_block(_abi.MMX_JC_FIXED_AXIS_COS, [2, 4], rng, function_weight=0.0),
        make_block(_abi.MMX_JC_NORMAL, [7, 1], rng, loss=(0.0, 0.8)),  # Cauchy
    ]
    blocks[0].weight[1] = 0.0
    Kp, Ko = 2, 1
    cons = orc.Constraints(
        np.array([7, 4], np.int32), rng.uniform(-1, 1, (Kp, 3)), rng.uniform(-2, 2, (Kp, 3)), np.ones(Kp),
        np.array([5], np.int32), rng.normal(size=(Ko, 4)), rng.normal(size=(Ko, 4)), np.ones(Ko), joint_blocks=blocks,
    )  # fmt: skip
    assert cons.rows == 3 * Kp + 9 * Ko + 3 + 3 + 2 + 2
    theta = rng.uniform(-0.6, 0.6, P)
    enabled = np.ones(P, np.uint8)
    enabled[[4, 9]] = 0
    J, r, err = orc.eval_jacobian(rig, cons, theta, enabled=enabled)
    base = 3 * Kp + 9 * Ko
    assert not J[base + 1].any() and r[base + 1] == 0  # weight 0 constraint
    assert not J[base + 6 : base + 8].any() and not r[base + 6 : base + 8].any()  # disabled block
    assert not J[:, 4].any() and not J[:, 9].any()
    # finite differences on the whole stack (the Cauchy block breaks |r|^2 == error: check the gradient instead)
    h = 1e-6
    for p in range(P):
        if not enabled[p]:
            continue
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h
        tm[p] -= h
        ep = orc.eval_jacobian(rig, cons, tp, enabled=enabled)[2]
        em = orc.eval_jacobian(rig, cons, tm, enabled=enabled)[2]
        g = 2 * J[:, p] @ r
        assert abs((ep - em) / (2 * h) - g) <= 2e-5 * max(1.0, abs(g))


def test_solve_reaches_the_planes():
    """GN on plane + fixed-axis constraints converges and the f32 / f64 instantiations agree."""
    rig = make_test_character(6)
    rng = np.random.default_rng(11)
    blocks = [make_block(_abi.MMX_JC_PLANE, [5, 3], rng, weight=1.0), make_block(_abi.MMX_JC_FIXED_AXIS_DIFF, [4], rng, weight=1.0)]
    cons = cons_with(blocks)
    opt = _abi.GnOptions.make(min_iterations=12, max_iterations=12, regularization=0.05)
    th0 = np.zeros(rig.num_params)
    r64 = orc.solve(rig, cons, th0, opt, dtype="f64")
    r32 = orc.solve(rig, cons, th0.astype(np.float32), opt, dtype="f32")
    e0 = orc.get_error(rig, cons, th0)
    assert r64["error"] < 1e-2 * e0
    assert np.abs(r64["theta"] - r32["theta"]).max() < 1e-4
