"""The oracle's LM schedule (stepRule 1: the lambda form of TrustRegionQRT's radius rule,
momentum/character_solver/trust_region_qr.cpp:244-268) reports the damping and the gain ratio of every iteration; they must
obey the rule they were decided by, in both precisions, and leave the solve itself untouched."""
import numpy as np

from momentum_amd import humanoid72_landmark_joints, make_humanoid72
from momentum_amd._abi import MMX_STEP_LM_SCHEDULE, GnOptions
from tests.helpers import make_problem


def test_step_history_obeys_the_rule(orc):
    rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
    lm = humanoid72_landmark_joints(rig)
    B = 24
    cons, th0, _ = make_problem(rig, lm, lm, B, seed=5, perturb=0.3)
    opt = GnOptions.make(min_iterations=8, max_iterations=8, threshold=1.0, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE)
    for dtype, T in (("f64", np.float64), ("f32", np.float32)):
        plain = orc.solve_batch(rig, cons, th0, opt, dtype=dtype, nthreads=2)
        out = orc.solve_batch(rig, cons, th0, opt, dtype=dtype, nthreads=2, step_history=True)
        assert np.array_equal(plain["theta"], out["theta"]) and np.array_equal(plain["error_history"], out["error_history"])
        lam, rho, h = out["lambda_history"], out["gain_ratio_history"], out["error_history"]
        assert lam.shape == (B, 8) and np.allclose(lam[:, 0], T(0.05))
        nxt = np.where(~(rho >= 0.25), T(4.0) * lam.astype(T), np.where(rho > 0.75, T(0.5) * lam.astype(T), lam.astype(T))).astype(np.float64)
        assert np.array_equal(nxt[:, :-1], lam[:, 1:])
        assert np.array_equal(~(rho[:, :-1] > 0), h[:, 1:] == h[:, :-1])  # a rejected step leaves the error where it was
        assert (rho > 0.75).any() and np.isfinite(rho).all()


def test_branch_analysis_counts_a_flip():
    import bench

    K = 4
    lam = np.tile(0.05 * 0.5 ** np.arange(K), (3, 1))
    rho_ref = np.full((3, K), 0.9)
    ref = {"lambda_history": lam.copy(), "gain_ratio_history": rho_ref.copy(), "error_history": np.ones((3, K))}
    g = np.stack([lam, rho_ref], axis=-1).copy()
    g[1, 2, 1] = 0.7499  # the GPU's ratio fell on the other side of 0.75 at iteration 2 ...
    ref["gain_ratio_history"][1, 2] = 0.7503  # ... of a double ratio 3e-4 from the threshold
    g[1, 3, 0] = lam[1, 2]  # (no scaling: the next lambda is the old one)
    rel = np.array([1e-7, 3e-3, 2e-7])
    res = bench.lm_branch_analysis(g, np.ones((3, K)), ref, rel)
    assert res["same_decisions"] == 2 and res["lm_branch_flips"] == 1 and res["flip_iteration"] == [2]
    assert res["num_above_bound"] == 1 and res["num_above_bound_with_same_decisions"] == 0 and res["pass"]
    assert res["flips_double_rho_within_1e-3_of_threshold"] == 1 and abs(res["flip_max_distance_of_double_rho_to_threshold"] - 3e-4) < 1e-9
