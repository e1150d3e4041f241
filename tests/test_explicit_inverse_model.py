"""Index algebra of the explicit-inverse experiment of the fused solve (tests/explicit_inverse_np.py is the lane-level
model the kernel code follows): in-place tile inversion and the two mat-vec sweeps against numpy."""
import numpy as np
import pytest

from tests import explicit_inverse_np as m


def _factor(NB, seed):
    rng = np.random.default_rng(seed)
    n = 16 * NB
    A = rng.normal(size=(n + 8, n))
    H = A.T @ A + 0.05 * np.eye(n)
    return np.linalg.cholesky(H), H, rng


@pytest.mark.parametrize("NB", [1, 2, 3, 5, 6, 7, 8])
def test_tiles_become_the_inverse_and_the_sweeps_solve(NB):
    L, H, rng = _factor(NB, 40 + NB)
    lds, inv_diag = m.store_factor(L, NB)
    m.invert_in_place(lds, inv_diag, NB)
    X = np.linalg.inv(L)
    for I in range(NB):
        for Jc in range(I):
            blk = X[16 * I : 16 * I + 16, 16 * Jc : 16 * Jc + 16]
            got = np.array([[lds[256 * m.tile_index(I, Jc) + m.tile_addr(r, c)] for c in range(16)] for r in range(16)])
            assert np.allclose(got, blk, rtol=1e-9, atol=1e-12), (I, Jc)
    b = rng.normal(size=16 * NB)
    x = m.solve_with_inverse(lds, inv_diag, NB, b.copy())
    assert np.allclose(x, np.linalg.solve(H, b), rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("NB", range(1, 9))
def test_every_block_has_exactly_one_wave(NB):
    rows = sorted(I for w in range(4) for I in m.row_blocks_of_wave(NB, w))
    assert rows == list(range(NB))
    cols = sorted(Jc for w in range(4) for Jc in m.col_blocks_of_wave(NB, w))
    assert cols == list(range(NB))
    for j in range(NB - 1):
        tiles = sorted(i for w in range(4) for i in m.column_tiles_of_wave(NB, j, w))
        assert tiles == list(range(j + 1, NB)), (j, tiles)
        assert max(len(m.column_tiles_of_wave(NB, j, w)) for w in range(4)) <= 2
