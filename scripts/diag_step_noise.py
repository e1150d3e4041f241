"""Where does the HIP path's distance to the double solve come from?  On the well-determined 128-parameter problem
(a position + orientation constraint on every joint, shared short-lever parameters), next to the oracle's float
instantiation, everything measured against the oracle's double run of the same inputs:

  1. H = J^T J and g = J^T r as the fused kernel builds them (parity hook), as errors of the STEP they produce;
  2. one Gauss-Newton iteration from a mid-solve pose, with no / one / up to three refinement rounds;
  3. the distance after k = 1..10 iterations.
usage: python scripts/diag_step_noise.py [variant=p128] [lambda=1e-3] [B=256] [route=fused]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from momentum_amd import capi, make_humanoid72
from momentum_amd._abi import GnOptions
from oracle import oracle as orc
from tests.helpers import make_problem

variant = sys.argv[1] if len(sys.argv) > 1 else "p128"
lam = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
route = sys.argv[4] if len(sys.argv) > 4 else "fused"
cores = bench.usable_cores()
rig = make_humanoid72(seed=12345, variant=variant, unit=0.01)
allj = list(range(rig.num_joints))
cons, th0, ths = make_problem(rig, allj, allj, B, seed=31337, perturb=0.3)
rh = capi.RigHandle(rig, 0)
pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
dev = pb.device
t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
pb.set_constraints(
    t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
    t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
)  # fmt: skip


def rel(a, ref):
    return np.linalg.norm(a - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)


def q(v):
    return f"median {np.median(v):.2e} p90 {np.quantile(v, .9):.2e} max {v.max():.2e}"


def opt(k):
    return GnOptions.make(min_iterations=k, max_iterations=k, threshold=1.0, regularization=lam)


# ---- a mid-solve pose: the double run's iterate after four iterations
mid = orc.solve_batch(rig, cons, th0, opt(4), dtype="f64", nthreads=cores)["theta"].astype(np.float32)
nb = min(B, 32)
print(f"== {variant} lambda {lam:g} B {B} route {route}")
lst, H, g = pb.fused_normal_equations(torch.from_numpy(mid).to(dev))
H, g = H.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64)
eH, eG, eH32, eG32, eD0, eD032 = [], [], [], [], [], []
for b in range(nb):
    c = cons.instance(b)
    J, r, _ = orc.eval_jacobian(rig, c, mid[b].astype(np.float64), dtype="f64")
    J, Hx = J[:, lst], None
    Hx = J.T @ J
    gx = J.T @ r
    Hd = Hx + lam * np.eye(len(lst))
    dx = np.linalg.solve(Hd, gx)
    J32, r32, _ = orc.eval_jacobian(rig, c, mid[b], dtype="f32")
    J32 = J32[:, lst].astype(np.float32)
    H32 = (J32.T @ J32).astype(np.float64)
    g32 = (J32.T @ r32.astype(np.float32)).astype(np.float64)
    Hg = np.tril(H[b]) + np.tril(H[b], -1).T
    sc = np.sqrt(np.outer(np.diag(Hx), np.diag(Hx))) + 1e-30
    eH.append(np.abs(Hg - Hx).max() / np.abs(Hx).max())
    eH32.append(np.abs(H32 - Hx).max() / np.abs(Hx).max())
    nd = np.linalg.norm(dx)
    eG.append(np.linalg.norm(np.linalg.solve(Hd, g[b] - gx)) / nd)
    eG32.append(np.linalg.norm(np.linalg.solve(Hd, g32 - gx)) / nd)
    # the step the kernel's H and g give when solved exactly: what no refinement can do better than without a better residual
    eD0.append(np.linalg.norm(np.linalg.solve(Hg + lam * np.eye(len(lst)), g[b]) - dx) / nd)
    eD032.append(np.linalg.norm(np.linalg.solve(H32 + lam * np.eye(len(lst)), g32) - dx) / nd)
print("1. normal equations at the mid pose (errors relative to the exact step; float oracle = J32^T J32 in f32)")
print("   max |dH| / max |H|          hip:", q(np.array(eH)), "| float oracle:", q(np.array(eH32)))
print("   |H^-1 dg| / |step|          hip:", q(np.array(eG)), "| float oracle:", q(np.array(eG32)))
print("   exact solve of (H~, g~)     hip:", q(np.array(eD0)), "| float oracle:", q(np.array(eD032)))

# ---- one iteration from the mid pose
ref1 = orc.solve_batch(rig, cons, mid, opt(1), dtype="f64", nthreads=cores)["theta"]
f321 = orc.solve_batch(rig, cons, mid, opt(1), dtype="f32", nthreads=cores)["theta"].astype(np.float64)
step = np.linalg.norm(ref1 - mid, axis=1)
print("2. one iteration from the mid pose: |theta_1 - theta_1(double)| / |step|")
print("   float oracle              :", q(np.linalg.norm(f321 - ref1, axis=1) / step))
for steps in (-1, 1, 0):
    pb.set_route(route, steps)
    out = pb.solve(torch.from_numpy(mid.copy()).to(dev), opt(1))
    torch.cuda.synchronize()
    th = out["theta"].cpu().numpy().astype(np.float64)
    label = {-1: "no refinement", 1: "one round", 0: "default (<= 3 rounds)"}[steps]
    print(f"   hip, {label:22s}:", q(np.linalg.norm(th - ref1, axis=1) / step))
pb.set_route(route, 0)

# ---- after k iterations from theta0
print("3. relative distance of theta to the double run after k iterations (hip | float oracle)")
o10 = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam)
out = pb.solve(torch.from_numpy(th0.copy()).to(dev), o10, want_parameter_history=True)
torch.cuda.synchronize()
ph = out["parameter_history"].cpu().numpy().astype(np.float64)
for k in (1, 2, 4, 6, 8, 10):
    rk = orc.solve_batch(rig, cons, th0, opt(k), dtype="f64", nthreads=cores)["theta"]
    fk = orc.solve_batch(rig, cons, th0, opt(k), dtype="f32", nthreads=cores)["theta"].astype(np.float64)
    hip = ph[:, k - 1] if k < 10 else out["theta"].cpu().numpy().astype(np.float64)
    print(f"   k = {k:2d}: hip {q(rel(hip, rk))} | float oracle {q(rel(fk, rk))}")
