"""p128_all_joints, lambda 1e-3, directional line search: the instances furthest from the double run -- per iteration the
step length of the HIP path against the double run's (a backtracking decision that differs shows as a ratio of 2^k)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from momentum_amd import capi, make_humanoid72
from momentum_amd._abi import GnOptions
from oracle import oracle as orc
from tests.helpers import make_problem

rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
allj = list(range(rig.num_joints))
B = 1024
cons, th0, _ = make_problem(rig, allj, allj, B, seed=31337, perturb=0.3)
pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
dev = pb.device
t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(dev)
pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                   t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))
opt = lambda k: GnOptions.make(min_iterations=k, max_iterations=k, threshold=1.0, regularization=1e-3, do_line_search=2)
pb.set_route(sys.argv[1] if len(sys.argv) > 1 else "fused")
out = pb.solve(torch.from_numpy(th0.copy()).to(dev), opt(10), want_history=True, want_parameter_history=True)
torch.cuda.synchronize()
ph = out["parameter_history"].cpu().numpy().astype(np.float64)
ref = orc.solve_batch(rig, cons, th0, opt(10), dtype="f64", nthreads=bench.usable_cores())
rel = np.linalg.norm(ph[:, 9] - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
worst = np.argsort(rel)[-4:][::-1]
print("worst instances", worst, rel[worst])
sub = cons.subset(worst)
prev = th0[worst].astype(np.float64)
prevg = th0[worst].astype(np.float64)
for k in range(1, 11):
    rk = orc.solve_batch(rig, sub, th0[worst], opt(k), dtype="f64", nthreads=4)["theta"]
    sr = np.linalg.norm(rk - prev, axis=1)
    sg = np.linalg.norm(ph[worst, k - 1] - prevg, axis=1)
    d = np.linalg.norm(ph[worst, k - 1] - rk, axis=1) / np.linalg.norm(rk, axis=1)
    print(f"k={k:2d} step double {np.array2string(sr, precision=3)} hip/double {np.array2string(sg / sr, precision=4)} rel dist {np.array2string(d, precision=2)}")
    prev, prevg = rk, ph[worst, k - 1]
