"""Stage by stage: which part of the HIP path's g = J^T r differs from the double value, at a mid-solve pose of the
well-determined 128-parameter problem?  Everything as the error of the STEP it causes, |H^-1 dg| / |step|, by parameter group.
  a  J-assembly kernel's J, r (same FK / unit helpers as the solve kernels), product in double
  a' ... product in single precision (numpy)
  b  the fused kernel's g (phases C-F: moments about the world origin)
  c  the wide route's tree normal equations g
  o  the oracle's float instantiation: J32, r32, product in double / in single
usage: python scripts/diag_g_stages.py [variant=p128] [lambda=1e-3] [B=64] [k_mid=4]"""
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
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
kmid = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cores = bench.usable_cores()
rig = make_humanoid72(seed=12345, variant=variant, unit=0.01)
allj = list(range(rig.num_joints))
cons, th0, ths = make_problem(rig, allj, allj, B, seed=31337, perturb=0.3)
pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
dev = pb.device
t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
pb.set_constraints(
    t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
    t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
)  # fmt: skip
opt = GnOptions.make(min_iterations=kmid, max_iterations=kmid, threshold=1.0, regularization=lam)
mid = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=cores)["theta"].astype(np.float32)
thd = torch.from_numpy(mid).to(dev)
lst, Hf, gf = pb.fused_normal_equations(thd)
gf = gf.cpu().numpy().astype(np.float64)
Ht, gt = pb.tree_normal_equations(thd)
gt = gt.cpu().numpy().astype(np.float64)
Jg, rg, _ = pb.eval_jacobian(thd)
Jg, rg = Jg.cpu().numpy(), rg.cpu().numpy()  # [B][P][M] column-major M x P per instance
enabled = np.arange(rig.num_params)  # everything enabled: the enabled list is 0..P-1
# parameter groups by the deepest joint a parameter drives
depth = rig.depth()
A = rig.dense_transform()
pdepth = np.array([depth[np.flatnonzero(np.abs(A[:, p]) > 0) // 7].max() if np.abs(A[:, p]).any() else 0 for p in range(rig.num_params)])
groups = {"depth 0-3": (0, 3), "depth 4-7": (4, 7), "depth 8-9": (8, 9), "depth 10+": (10, 99)}
res = {}


def add(tag, v):
    res.setdefault(tag, []).append(v)


for b in range(B):
    c = cons.instance(b)
    J, r, _ = orc.eval_jacobian(rig, c, mid[b].astype(np.float64), dtype="f64")
    Jl = J[:, lst]
    Hd = Jl.T @ Jl + lam * np.eye(len(lst))
    gx = Jl.T @ r
    dx = np.linalg.solve(Hd, gx)
    nd = np.linalg.norm(dx)
    Jb = Jg[b].T[:, lst]  # [M][n]
    rb = rg[b]
    J32, r32, _ = orc.eval_jacobian(rig, c, mid[b], dtype="f32")
    J32 = J32[:, lst].astype(np.float32)
    r32 = r32.astype(np.float32)
    cand = {
        "a  hip J,r (J-assembly kernel), double product": Jb.astype(np.float64).T @ rb.astype(np.float64),
        "a' hip J,r, single product": (Jb.T @ rb).astype(np.float64),
        "b  fused kernel g": gf[b],
        "o  float oracle J,r, double product": J32.astype(np.float64).T @ r32.astype(np.float64),
        "o' float oracle J,r, single product": (J32.T @ r32).astype(np.float64),
    }
    # the wide hook returns the enabled-list order; everything is enabled here, so map through lst
    cand["c  tree normal equations g (wide route)"] = gt[b][lst] if gt.shape[1] == rig.num_params else gt[b]
    for tag, g in cand.items():
        dd = np.linalg.solve(Hd, g - gx)
        add(tag, np.linalg.norm(dd) / nd)
        for gn, (lo, hi) in groups.items():
            m = (pdepth[lst] >= lo) & (pdepth[lst] <= hi)
            add(tag + " | " + gn, np.linalg.norm(dd[m]) / nd)
    # residual rows themselves
    add("r: |r_hip - r64| / |r64|", np.linalg.norm(rb - r) / np.linalg.norm(r))
    add("r: |r_f32oracle - r64| / |r64|", np.linalg.norm(r32 - r) / np.linalg.norm(r))
    add("|step| / |theta|", nd / np.linalg.norm(mid[b]))
print(f"== {variant} lambda {lam:g} B {B}: mid pose = double iterate after {kmid} iterations; medians over the batch")
for tag, v in res.items():
    v = np.array(v)
    print(f"  {tag:75s} median {np.median(v):.2e}  p90 {np.quantile(v, .9):.2e}")
