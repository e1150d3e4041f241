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


def test_panel_tile_times_inverse_transpose_of_the_packed_diagonal_block():
    """MMX_EXP_MFMAPANEL: X = T L_kk^-T for a panel tile, operands as the kernel reads them (A: a row chunk of T,
    B: the packed diagonal tile right of its diagonal / invDiag), result stored in place."""
    rng = np.random.default_rng(7)
    A = rng.normal(size=(40, 32))
    L = np.linalg.cholesky(A.T @ A + 0.1 * np.eye(32))
    lds, inv_diag = m.store_factor(L, 2)
    # tile (1, 0) still holds H's block: put a random one there and solve it against L_00
    T = rng.normal(size=(16, 16))
    base = 256 * m.tile_index(1, 0)
    for r in range(16):
        for c in range(16):
            lds[base + m.tile_addr(r, c)] = T[r, c]
    dk = 256 * m.tile_index(0, 0)
    a = np.zeros((4, 64))
    b = np.zeros((4, 64))
    for l in range(64):
        q, g = l & 15, l >> 4
        a[:, l] = m.lds_row4(lds, base, q, g)
        for e in range(4):
            kk = 4 * g + e
            b[e, l] = lds[dk + m.tile_addr(kk, q)] if q > kk else (inv_diag[kk] if q == kk else 0.0)
    acc = np.zeros((64, 4))
    for e in range(4):
        m.mfma(a[e], b[e], acc)
    for l in range(64):
        q, g = l & 15, l >> 4
        for r in range(4):
            lds[base + m.tile_addr(4 * g + r, q)] = acc[l, r]
    got = np.array([[lds[base + m.tile_addr(r, c)] for c in range(16)] for r in range(16)])
    assert np.allclose(got, T @ np.linalg.inv(L[:16, :16]).T, rtol=1e-10, atol=1e-12)
