"""Lane-level numpy model of the fused solve's explicit-inverse experiment (momentum_amd/csrc/mmx_fused.hip,
MMX_EXP_INVERSE): the tiled factor in its LDS storage format is turned into L^-1 in place by block columns with
16x16x4 matrix-core products, and (L L^T) x = b becomes two triangular mat-vecs over the four waves.

The model follows the kernel statement by statement -- same tile storage (swizzled 16x16 tiles, packed diagonal
tiles: L_kk below, the strict upper triangle of L_kk^-T above, 1 / l_ii in invDiag), same operand lanes, same work
split over the waves -- so that the index algebra can be checked on the CPU (tests/test_explicit_inverse_model.py)
before the kernel runs anywhere.

Matrix-core convention (v_mfma_f32_16x16x4_f32, the one every product in mmx_fused.hip relies on): lane l supplies
A[l & 15][l >> 4] and B[l >> 4][l & 15]; afterwards lane l holds D[4 (l >> 4) + r][l & 15] in accumulator slot r.
"""
import numpy as np


def tile_addr(row, col):
    return row * 16 + ((((col >> 2) ^ (row >> 2)) & 3) << 2) + (col & 3)


def tile_index(i, j):
    return i * (i + 1) // 2 + j


def lds_row4(lds, base, row, chunk):
    a = base + row * 16 + (((chunk ^ (row >> 2)) & 3) << 2)
    return lds[a : a + 4].copy()


def mfma(a, b, acc):
    """a, b: [64] per-lane operands; acc: [64, 4] accumulators (updated in place)."""
    A = np.zeros((16, 4), acc.dtype)
    B = np.zeros((4, 16), acc.dtype)
    for l in range(64):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    D = A @ B
    for l in range(64):
        for r in range(4):
            acc[l, r] += D[4 * (l >> 4) + r, l & 15]


def store_factor(Lmat, NB, dtype=np.float64):
    """The factor as phase H leaves it: (lds tiles, invDiag)."""
    T = NB * (NB + 1) // 2
    lds = np.zeros(256 * T, dtype)
    inv_diag = np.zeros(16 * NB, dtype)
    for I in range(NB):
        for Jc in range(I + 1):
            blk = Lmat[16 * I : 16 * I + 16, 16 * Jc : 16 * Jc + 16]
            base = 256 * tile_index(I, Jc)
            if I != Jc:
                for r in range(16):
                    for c in range(16):
                        lds[base + tile_addr(r, c)] = blk[r, c]
            else:
                inv = np.linalg.inv(blk)  # L_kk^-1 (lower); the tile's upper triangle holds L_kk^-T
                for r in range(16):
                    for c in range(16):
                        lds[base + tile_addr(r, c)] = blk[r, c] if c <= r else inv[c, r]
                    inv_diag[16 * I + r] = inv[r, r]
    return lds, inv_diag


def column_tiles_of_wave(NB, j, w):
    """Tiles (i, j), i > j, of block column j that wave w inverts: the four lowest tiles (longest sums) one per wave,
    the rest dealt to the waves with the shortest first tile (NB <= 8: at most two per wave)."""
    out = []
    for i in (NB - 1 - w, NB - 8 + w):
        if j + 1 <= i <= NB - 1 and i not in out:
            out.append(i)
    return out


def invert_in_place(lds, inv_diag, NB):
    """Off-diagonal tiles T(i, j) := (L^-1)_ij; the diagonal tiles stay packed."""
    dt = lds.dtype
    for j in range(NB - 2, -1, -1):
        results = []  # (i, per-lane float4) held in registers across the barrier
        for w in range(4):
            for i in column_tiles_of_wave(NB, j, w):
                acc = [np.zeros((64, 4), dt), np.zeros((64, 4), dt)]
                for k in range(j + 1, i + 1):
                    a = np.zeros((4, 64), dt)
                    bq = np.zeros((4, 64), dt)
                    tkj = 256 * tile_index(k, j)
                    for l in range(64):
                        q, g = l & 15, l >> 4
                        for s in range(4):
                            kk = 4 * s + g  # (rows 4 s + g of a tile column: 64 lanes, 64 banks)
                            a[s, l] = lds[tkj + tile_addr(kk, q)]  # A[i'][kk] = L_kj[kk][i']
                            if k < i:
                                bq[s, l] = lds[256 * tile_index(i, k) + tile_addr(q, kk)]  # B[kk][j'] = X_ik[j'][kk]
                            else:
                                di = 256 * tile_index(i, i)
                                bq[s, l] = lds[di + tile_addr(kk, q)] if kk < q else (inv_diag[16 * i + q] if kk == q else 0.0)
                    for s in range(4):
                        mfma(a[s], bq[s], acc[k & 1])
                S = acc[0] + acc[1]  # S^T[4g + r][j'] in slot r
                res = np.zeros((64, 4), dt)
                dj = 256 * tile_index(j, j)
                a = np.zeros((4, 64), dt)
                for l in range(64):
                    q, g = l & 15, l >> 4
                    row = lds_row4(lds, dj, q, g)
                    for s in range(4):
                        kk = 4 * g + s
                        a[s, l] = -(row[s] if q < kk else (inv_diag[16 * j + kk] if q == kk else 0.0))  # -Minv_j[kk][i']
                for s in range(4):
                    mfma(a[s], S[:, s].copy(), res)
                results.append((i, res))
        # __syncthreads(); every wave stores its tiles; __syncthreads()
        for i, res in results:
            tij = 256 * tile_index(i, j)
            for l in range(64):
                q, g = l & 15, l >> 4
                a0 = tij + q * 16 + (((g ^ (q >> 2)) & 3) << 2)
                lds[a0 : a0 + 4] = res[l]


def row_blocks_of_wave(NB, w):
    out = []
    for I in (NB - 1 - w, NB - 8 + w):
        if 0 <= I <= NB - 1 and I not in out:
            out.append(I)
    return out


def col_blocks_of_wave(NB, w):
    return [NB - 1 - I for I in row_blocks_of_wave(NB, w)]


def quad_sum(v):
    """v: [64] -> every lane of a quad gets the quad's sum."""
    out = np.zeros_like(v)
    for l in range(64):
        q = l & ~3
        out[l] = (v[l] + v[l ^ 1]) + (v[l ^ 2] + v[(l ^ 2) ^ 1])
    return out


def solve_with_inverse(lds, inv_diag, NB, x):
    """x := (L L^T)^-1 x with the inverted tiles; tmp holds y = L^-1 b between the two sweeps."""
    dt = lds.dtype
    tmp = np.zeros(16 * NB, dt)
    for w in range(4):  # forward: y_I = sum_{J <= I} X_IJ b_J
        for I in row_blocks_of_wave(NB, w):
            acc = np.zeros(64, dt)
            for l in range(64):
                i, g = l >> 2, l & 3
                for Jc in range(I):
                    acc[l] += lds_row4(lds, 256 * tile_index(I, Jc), i, g) @ x[16 * Jc + 4 * g : 16 * Jc + 4 * g + 4]
                dI = 256 * tile_index(I, I)
                for t in range(4):
                    c = 4 * t + g
                    m = lds[dI + tile_addr(c, i)] if c < i else (inv_diag[16 * I + i] if c == i else 0.0)
                    acc[l] += m * x[16 * I + c]
            acc = quad_sum(acc)
            for l in range(0, 64, 4):
                tmp[16 * I + (l >> 2)] = acc[l]
    # __syncthreads()
    for w in range(4):  # backward: x_J = sum_{I >= J} X_IJ^T y_I
        for Jc in col_blocks_of_wave(NB, w):
            acc = np.zeros(64, dt)
            for l in range(64):
                i, g = l >> 2, l & 3
                for I in range(Jc + 1, NB):
                    tij = 256 * tile_index(I, Jc)
                    for t in range(4):
                        c = 4 * t + g
                        acc[l] += lds[tij + tile_addr(c, i)] * tmp[16 * I + c]
                row = lds_row4(lds, 256 * tile_index(Jc, Jc), i, g)
                for e in range(4):
                    c = 4 * g + e
                    m = row[e] if c > i else (inv_diag[16 * Jc + i] if c == i else 0.0)
                    acc[l] += m * tmp[16 * Jc + c]
            acc = quad_sum(acc)
            for l in range(0, 64, 4):
                x[16 * Jc + (l >> 2)] = acc[l]
    # __syncthreads()
    return x
