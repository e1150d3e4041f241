"""Schedule of the fused solve's lookahead (mmx_fused.hip, phase H: kLook), block level: during panel k's
elimination the waves without a panel row take the contributions of the block columns j < k to block column k + 1, so
that the left-looking update in front of panel k + 1 is left with column k's alone.  Checked here: the schedule computes
the Cholesky factor, and a wave without a panel row exists at every step it is used at (NB <= 8)."""
import numpy as np
import pytest


def _blocked_cholesky_with_lookahead(H, NB):
    L = np.tril(H).copy()
    blk = lambda I, Jc: (slice(16 * I, 16 * I + 16), slice(16 * Jc, 16 * Jc + 16))
    for k in range(NB):
        if k > 0:  # (u) what is left for block column k
            j_first = k - 1 if k >= 2 else 0
            for I in range(k, NB):
                for j in range(j_first, k):
                    L[blk(I, k)] -= L[blk(I, j)] @ L[blk(k, j)].T
        if 1 <= k and k + 1 < NB:  # idle waves, under the chain: column k + 1 takes the finished columns j < k
            for I in range(k + 1, NB):
                for j in range(k):
                    L[blk(I, k + 1)] -= L[blk(I, j)] @ L[blk(k + 1, j)].T
        D = np.linalg.cholesky(np.tril(L[blk(k, k)]) + np.tril(L[blk(k, k)], -1).T)
        L[blk(k, k)] = D
        for I in range(k + 1, NB):
            L[blk(I, k)] = np.linalg.solve(D, L[blk(I, k)].T).T
    return np.tril(L)


@pytest.mark.parametrize("NB", [1, 2, 3, 6, 8])
def test_lookahead_schedule_factors(NB):
    rng = np.random.default_rng(NB)
    n = 16 * NB
    A = rng.normal(size=(n + 4, n))
    H = A.T @ A + 0.05 * np.eye(n)
    assert np.allclose(_blocked_cholesky_with_lookahead(H, NB), np.linalg.cholesky(H), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("NB", range(1, 9))
def test_a_wave_without_panel_rows_exists(NB):
    NP = 16 * NB
    for k in range(1, NB - 1):
        first_idle = (NP - 16 * k + 47) // 48  # waves 0 .. first_idle - 1 hold the 16 identity + the panel rows
        assert 1 <= first_idle <= 3
        works = [w == 0 or 16 * k + 48 * w < NP for w in range(4)]
        assert works == [w < first_idle for w in range(4)]
        tiles = sorted(I for w in range(first_idle, 4) for I in range(k + 1 + (w - first_idle), NB, 4 - first_idle))
        assert tiles == list(range(k + 1, NB))
