"""numpy prototype of the tree-structured normal-equation assembly used by the fused HIP kernel
(momentum_amd/csrc): J^T J from per-joint subtree moments, J^T y by an adjoint (leaves-to-root)
pass and J d by a tangent (root-to-leaves) pass -- without ever forming J.  Test infrastructure:
it documents the math and is validated against the oracle's explicit Jacobian
(tests/test_tree_algebra.py); the product implements the same formulas in HIP.

Notation: unit u = one constraint vector (a point for position
constraints, a direction for each column of an orientation constraint) with world vector p_u,
joint j_u, scale sigma_u, residual f_u.  For joint-parameter row (a,d) the derivative of p_u is
g = alpha + B p_u with
   points:      translation  alpha = tau_d (column d of parent.toLinear()), B = 0
                rotation     alpha = -omega x t_a,  B = [omega]x           (omega = rotationAxis col)
                scale        alpha = -ln2 t_a,      B = ln2 I
   directions:  rotation     alpha = 0,             B = [omega]x ; other dofs do not act.
"""
import numpy as np

LN2 = np.log(2.0)


def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.asarray(w).dtype)


def axial(G):
    """<G, [w]x>_F = w . axial(G)"""
    return np.array([G[2, 1] - G[1, 2], G[0, 2] - G[2, 0], G[1, 0] - G[0, 1]])


class Tree:
    def __init__(self, rig, state, enabled=None, dtype=np.float64, center=None):
        """state = oracle.skeleton_state(...) dict (world, trans_axis, rot_axis)."""
        self.rig = rig
        self.J = rig.num_joints
        self.P = rig.num_params
        self.parent = rig.parent
        self.dtype = dtype
        self.center = np.zeros(3) if center is None else np.asarray(center, dtype=np.float64)
        # all formulas depend on p - t_a only, so positions may be shifted by a common point
        self.t = (state["world"][:, :3].astype(np.float64) - self.center).astype(dtype)
        self.tau = state["trans_axis"].astype(dtype)  # [J,3,3], column d
        self.om = state["rot_axis"].astype(dtype)
        self.ln2 = dtype(LN2)
        self.enabled = np.ones(self.P, bool) if enabled is None else np.asarray(enabled, bool)
        self.A = rig.dense_transform().astype(dtype)
        self.A[:, ~self.enabled] = 0.0
        self.active = np.abs(self.A).sum(axis=1) > 0
        anc = np.zeros((self.J, self.J), bool)  # anc[a, j]: a ancestor-or-self of j
        for j in range(self.J):
            a = j
            while a >= 0:
                anc[a, j] = True
                a = self.parent[a]
        self.anc = anc

    # alpha, B of joint-parameter row (a, d) for points / directions
    def alphaB(self, a, d, point=True):
        if d < 3:
            z3, z33 = np.zeros(3, self.dtype), np.zeros((3, 3), self.dtype)
            return (self.tau[a][:, d], z33) if point else (z3, z33)
        if d < 6:
            w = self.om[a][:, d - 3]
            return (-np.cross(w, self.t[a]) if point else np.zeros(3, self.dtype)), skew(w)
        if point:
            return -self.ln2 * self.t[a], self.ln2 * np.eye(3, dtype=self.dtype)
        return np.zeros(3, self.dtype), np.zeros((3, 3), self.dtype)


def unit_arrays(cons_units):
    """cons_units: list of dict(joint, point(bool), p(3), sigma, f(3))"""
    return cons_units


def subtree_sums(tree, units, y=None):
    """Per joint: sums over the units in the joint's SUBTREE.
    second-order (for H): m0, m1, M2 (points), M2d (directions) weighted by sigma^2
    first-order (for J^T y): F, N, D (points), Nd (directions) with y_u a 3-vector per unit
    (y_u = sigma_u * f_u * sigma_u = sigma_u * r_u gives J^T r)."""
    J, T = tree.J, tree.dtype
    z = lambda *shape: np.zeros(shape, T)
    S = dict(m0=z(J), m1=z(J, 3), M2=z(J, 3, 3), M2d=z(J, 3, 3), F=z(J, 3), N=z(J, 3), D=z(J), Nd=z(J, 3))
    for i, u in enumerate(units):
        j, s2 = u["joint"], T(u["sigma"]) ** 2
        p = (u["p"] - tree.center).astype(T) if u["point"] else u["p"].astype(T)
        if y is not None:
            y = [np.asarray(v, T) for v in y] if i == 0 else y
        yy = None if y is None else y[i]
        if u["point"]:
            S["m0"][j] += s2
            S["m1"][j] += s2 * p
            S["M2"][j] += s2 * np.outer(p, p)
            if yy is not None:
                S["F"][j] += yy
                S["N"][j] += np.cross(p, yy)
                S["D"][j] += p @ yy
        else:
            S["M2d"][j] += s2 * np.outer(p, p)
            if yy is not None:
                S["Nd"][j] += np.cross(p, yy)
    for j in range(J - 1, 0, -1):  # children before parents (parent < child)
        pa = tree.parent[j]
        if pa >= 0:
            for k in S:
                S[k][pa] += S[k][j]
    return S


def jt_times(tree, S):
    """J^T y in model-parameter space from the first-order subtree sums."""
    gj = np.zeros(7 * tree.J, tree.dtype)
    LN2 = tree.ln2
    for a in range(tree.J):
        F, N, D, Nd, ta = S["F"][a], S["N"][a], S["D"][a], S["Nd"][a], tree.t[a]
        for d in range(3):
            gj[7 * a + d] = tree.tau[a][:, d] @ F
            w = tree.om[a][:, d]
            gj[7 * a + 3 + d] = w @ (N + Nd - np.cross(ta, F))
        gj[7 * a + 6] = LN2 * (D - ta @ F)
    gj[~tree.active] = 0.0
    return tree.A.T @ gj


def j_times(tree, units, delta):
    """J delta (3 rows per unit) by a tangent pass: prefix sums over ancestors."""
    T = tree.dtype
    LN2 = tree.ln2
    jd = tree.A @ np.asarray(delta, T)
    J = tree.J
    C = np.zeros((J, 3), T)
    W = np.zeros((J, 3), T)
    Sd = np.zeros(J, T)
    for a in range(J):
        Tv = tree.tau[a] @ jd[7 * a : 7 * a + 3]
        Om = tree.om[a] @ jd[7 * a + 3 : 7 * a + 6]
        sd = jd[7 * a + 6]
        own = Tv - np.cross(Om, tree.t[a]) - LN2 * sd * tree.t[a]
        pa = tree.parent[a]
        C[a] = own + (C[pa] if pa >= 0 else 0)
        W[a] = Om + (W[pa] if pa >= 0 else 0)
        Sd[a] = sd + (Sd[pa] if pa >= 0 else 0)
    out = np.zeros(3 * len(units), T)
    for i, u in enumerate(units):
        j = u["joint"]
        p = (u["p"] - tree.center).astype(T) if u["point"] else u["p"].astype(T)
        if u["point"]:
            v = C[j] + np.cross(W[j], p) + LN2 * Sd[j] * p
        else:
            v = np.cross(W[j], p)
        out[3 * i : 3 * i + 3] = T(u["sigma"]) * v
    return out


def joint_dof_tables(tree, S):
    """Per joint-parameter row: (Gamma0, axsum, tr) from its subtree moments (used when the row is
    the deeper one of a pair) and (alpha, type, omega) (used when it is the ancestor one)."""
    R = 7 * tree.J
    G0 = np.zeros((R, 3), tree.dtype)
    AX = np.zeros((R, 3), tree.dtype)
    TR = np.zeros(R, tree.dtype)
    for a in range(tree.J):
        m0, m1, M2, M2d = S["m0"][a], S["m1"][a], S["M2"][a], S["M2d"][a]
        for d in range(7):
            al, B = tree.alphaB(a, d, True)
            _, Bd = tree.alphaB(a, d, False)
            G0[7 * a + d] = al * m0 + B @ m1
            G1 = np.outer(al, m1) + B @ M2
            AX[7 * a + d] = axial(G1) + axial(Bd @ M2d)
            TR[7 * a + d] = np.trace(G1)
    return G0, AX, TR


def hj_entry(tree, tabs, r, rp):
    """H_joint[r, rp] with r = (a,d), rp = (a',d')."""
    G0, AX, TR = tabs
    a, d = divmod(r, 7)
    ap, dp = divmod(rp, 7)
    if tree.anc[ap, a]:
        deep, an_a, an_d = r, ap, dp
    elif tree.anc[a, ap]:
        deep, an_a, an_d = rp, a, d
    else:
        return tree.dtype(0.0)
    al, _ = tree.alphaB(an_a, an_d, True)
    h = G0[deep] @ al
    if 3 <= an_d < 6:
        h += tree.om[an_a][:, an_d - 3] @ AX[deep]
    elif an_d == 6:
        h += tree.ln2 * TR[deep]
    return h


def jtj(tree, S):
    tabs = joint_dof_tables(tree, S)
    P = tree.P
    src = [[(r, tree.A[r, p]) for r in np.flatnonzero(tree.A[:, p])] for p in range(P)]
    H = np.zeros((P, P), tree.dtype)
    for p in range(P):
        for q in range(p + 1):
            h = tree.dtype(0.0)
            for r, w in src[p]:
                for rp, wp in src[q]:
                    h += w * wp * hj_entry(tree, tabs, r, rp)
            H[p, q] = H[q, p] = h
    return H


# ---------------------------------------------------------------------------------------------
# Local-lever forms (round 4): the same two passes with every moment taken about the joint it
# belongs to instead of the world origin.  In exact arithmetic they equal jt_times / j_times; in
# single precision the origin forms subtract two terms of size |t_a| |F| to get one of size
# |lever| |F| (a finger joint a metre from the origin with a centimetre lever loses ~6.6 bits),
# the local forms only ever multiply by physical levers (bone vectors, p_u - t_j).
# ---------------------------------------------------------------------------------------------
def jt_times_local(tree, units, y):
    """J^T y: own sums about the unit's own joint, then leaves-to-root with the child's sums shifted to
    the parent: N_a += N_c + (t_c - t_a) x F_c, D_a += D_c + (t_c - t_a) . F_c, F_a += F_c."""
    T, J = tree.dtype, tree.J
    F, N, D = np.zeros((J, 3), T), np.zeros((J, 3), T), np.zeros(J, T)
    for i, u in enumerate(units):
        j, yy = u["joint"], np.asarray(y[i], T)
        if u["point"]:
            lever = (u["p"] - tree.center).astype(T) - tree.t[j]
            F[j] += yy
            N[j] += np.cross(lever, yy)
            D[j] += lever @ yy
        else:
            N[j] += np.cross(u["p"].astype(T), yy)
    for j in range(J - 1, 0, -1):  # children before parents (parent < child)
        pa = tree.parent[j]
        if pa >= 0:
            lever = tree.t[j] - tree.t[pa]
            F[pa] += F[j]
            N[pa] += N[j] + np.cross(lever, F[j])
            D[pa] += D[j] + lever @ F[j]
    gj = np.zeros(7 * J, T)
    for a in range(J):
        for d in range(3):
            gj[7 * a + d] = tree.tau[a][:, d] @ F[a]
            gj[7 * a + 3 + d] = tree.om[a][:, d] @ N[a]
        gj[7 * a + 6] = tree.ln2 * D[a]
    gj[~tree.active] = 0.0
    return tree.A.T @ gj


def j_times_local(tree, units, delta):
    """J delta: V_a = velocity of the ancestors' motion AT t_a, root to leaves:
    V_a = T_a + V_p + W_p x (t_a - t_p) + ln2 S_p (t_a - t_p); a unit adds W_a x (p - t_a) + ln2 S_a (p - t_a)."""
    T, J = tree.dtype, tree.J
    jd = tree.A @ np.asarray(delta, T)
    V, W, Sd = np.zeros((J, 3), T), np.zeros((J, 3), T), np.zeros(J, T)
    for a in range(J):
        Tv = tree.tau[a] @ jd[7 * a : 7 * a + 3]
        Om = tree.om[a] @ jd[7 * a + 3 : 7 * a + 6]
        pa = tree.parent[a]
        if pa >= 0:
            lever = tree.t[a] - tree.t[pa]
            V[a] = Tv + (V[pa] + np.cross(W[pa], lever) + tree.ln2 * Sd[pa] * lever)
            W[a] = Om + W[pa]
            Sd[a] = jd[7 * a + 6] + Sd[pa]
        else:
            V[a], W[a], Sd[a] = Tv, Om, jd[7 * a + 6]
    out = np.zeros(3 * len(units), T)
    for i, u in enumerate(units):
        j = u["joint"]
        if u["point"]:
            lever = (u["p"] - tree.center).astype(T) - tree.t[j]
            v = V[j] + np.cross(W[j], lever) + tree.ln2 * Sd[j] * lever
        else:
            v = np.cross(W[j], u["p"].astype(T))
        out[3 * i : 3 * i + 3] = T(u["sigma"]) * v
    return out
