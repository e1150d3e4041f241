"""Shared test helpers: seeded synthetic IK problems (targets = FK(theta*) through the ORACLE)."""
from __future__ import annotations

import numpy as np

from momentum_amd.rigs import Rig
from oracle import oracle as orc


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array(
        [
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by + ay * bw + az * bx - ax * bz,
            aw * bz + az * bw + ax * by - ay * bx,
            aw * bw - ax * bx - ay * by - az * bz,
        ]
    )


def quat_rot(q, v):
    qv = np.asarray(q[:3], dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    uv = 2.0 * np.cross(qv, v)
    return v + q[3] * uv + np.cross(qv, uv)


def rand_quat(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_problem(
    rig: Rig,
    pos_parent,
    ori_parent,
    batch: int,
    seed: int = 12345,
    perturb: float = 0.3,
    random_offsets: bool = False,
    theta0_scale: float = 0.0,
    weights: str = "ones",
    offset_scale: float = 1.0,
):
    """Synthetic batch as in SURVEY.md section 8d: instance i is seeded with seed+i; targets are
    FK(theta*) with theta* = theta0 + U[-perturb, perturb]^P.  Returns (Constraints [B,...],
    theta0 [B,P] float32, theta_star [B,P])."""
    pos_parent = np.asarray(pos_parent, dtype=np.int32).reshape(-1)
    ori_parent = np.asarray(ori_parent, dtype=np.int32).reshape(-1)
    Kp, Ko, P = len(pos_parent), len(ori_parent), rig.num_params
    po = np.zeros((batch, Kp, 3), np.float32)
    pt = np.zeros((batch, Kp, 3), np.float32)
    pw = np.ones((batch, Kp), np.float32)
    oo = np.zeros((batch, Ko, 4), np.float32)
    oo[..., 3] = 1.0
    ot = np.zeros((batch, Ko, 4), np.float32)
    ow = np.ones((batch, Ko), np.float32)
    th0 = np.zeros((batch, P), np.float32)
    ths = np.zeros((batch, P), np.float32)
    for b in range(batch):
        rng = np.random.default_rng(seed + b)
        th0[b] = (theta0_scale * rng.uniform(-1, 1, size=P)).astype(np.float32)
        ths[b] = (th0[b] + rng.uniform(-perturb, perturb, size=P)).astype(np.float32)
        if random_offsets:
            po[b] = offset_scale * rng.uniform(-1, 1, size=(Kp, 3))
            oo[b] = rand_quat(rng, Ko)
        if weights == "random":
            pw[b] = rng.uniform(0.2, 2.0, size=Kp)
            ow[b] = rng.uniform(0.2, 2.0, size=Ko)
        st = orc.skeleton_state(rig, ths[b].astype(np.float64), "f64")["world"]
        for c in range(Kp):
            w = st[pos_parent[c]]
            pt[b, c] = w[:3] + quat_rot(w[3:7], w[7] * po[b, c].astype(np.float64))
        for c in range(Ko):
            w = st[ori_parent[c]]
            ot[b, c] = quat_mul(w[3:7], oo[b, c].astype(np.float64))
    cons = orc.Constraints(pos_parent, po, pt, pw, ori_parent, oo, ot, ow)
    return cons, th0, ths
