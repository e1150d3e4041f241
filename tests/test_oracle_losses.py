"""GeneralizedLoss in the oracle (momentum/math/generalized_loss.cpp), pinned like the reference's
own test (momentum/test/math/generalized_loss_test.cpp): through the position / orientation error
functions the Jacobian must match finite differences of the robustified residual, |r|^2 = sum w loss'(s) s,
error = sum w loss(s), and 2 J^T r = the gradient of the error."""
import numpy as np
import pytest

from momentum_amd import make_test_character
from momentum_amd._abi import MMX_LOSS_WELSCH
from oracle import oracle as orc
from tests.helpers import make_problem


def _value(alpha, c, s):
    q = s / (c * c)
    if alpha == 2.0:
        return q
    if alpha == 1.0:
        return np.sqrt(q + 1) - 1
    if alpha == 0.0:
        return np.log(0.5 * q + 1)
    if alpha == MMX_LOSS_WELSCH:
        return 1 - np.exp(-0.5 * q)
    return (np.power(q / abs(alpha - 2) + 1, 0.5 * alpha) - 1) * abs(alpha - 2) / alpha  # generalized_loss_test.cpp:31-36


@pytest.mark.parametrize("alpha", [2.0, 1.0, 0.0, MMX_LOSS_WELSCH, -2.0, 0.5, 10.0])
@pytest.mark.parametrize("c", [1.0, 0.3])
def test_robust_loss_through_error_functions(alpha, c):
    rig = make_test_character(6)
    cons, th0, ths = make_problem(rig, [5, 2], [4, 1], 1, seed=31, perturb=0.5, random_offsets=True, weights="random")
    ci = cons.instance(0)
    ci.pos_loss, ci.ori_loss = (alpha, c), (alpha, 2 * c)
    ci.pos_function_weight, ci.ori_function_weight = 0.7, 1.3
    rng = np.random.default_rng(6)
    theta = rng.uniform(-0.5, 0.5, rig.num_params)
    J, r, err = orc.eval_jacobian(rig, ci, theta, dtype="f64")
    # the error is sum_c w_c * loss(|f_c|^2): recompute from the L2 residuals of the same problem
    l2 = cons.instance(0)
    l2.pos_function_weight, l2.ori_function_weight = 1.0, 1.0
    l2.pos_weight, l2.ori_weight = np.ones_like(l2.pos_weight), np.ones_like(l2.ori_weight)
    _, f, _ = orc.eval_jacobian(rig, l2, theta, dtype="f64")
    spos = (f[: 3 * ci.Kp].reshape(ci.Kp, 3) ** 2).sum(1)
    sori = (f[3 * ci.Kp :].reshape(ci.Ko, 9) ** 2).sum(1)
    expect = 0.7 * np.sum(ci.pos_weight * _value(alpha, c, spos)) + 1.3 * np.sum(ci.ori_weight * _value(alpha, 2 * c, sori))
    assert err == pytest.approx(expect, rel=1e-6)
    assert orc.get_error(rig, ci, theta, "f64") == pytest.approx(expect, rel=2e-6)
    # 2 J^T r = gradient of the error (error_function_helpers.cpp:220,262) -- this is what pins loss'(s)
    g = 2 * J.T @ r
    h = 1e-6
    for p in range(rig.num_params):
        tp, tm = theta.copy(), theta.copy()
        tp[p] += h
        tm[p] -= h
        ep = orc.eval_jacobian(rig, ci, tp, dtype="f64")[2]
        em = orc.eval_jacobian(rig, ci, tm, dtype="f64")[2]
        assert abs((ep - em) / (2 * h) - g[p]) <= 2e-5 * max(1.0, np.abs(g).max())
    J32, r32, e32 = orc.eval_jacobian(rig, ci, theta.astype(np.float32), dtype="f32")
    assert np.abs(J32 - J).max() <= 3e-5 * max(1.0, np.abs(J).max())
    assert abs(e32 - err) <= 3e-5 * max(1.0, err)
