"""cfg2 at lambda = 1e-7 with the directional line search: where does the HIP path leave the double solve -- error
histories of a few instances, with and without refinement steps, against the oracle in double and float."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from momentum_amd import capi, humanoid72_landmark_joints, make_humanoid72
from momentum_amd._abi import GnOptions
from tests.helpers import make_problem
from oracle import oracle as orc
np.set_printoptions(linewidth=250, precision=3)
rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
lm = humanoid72_landmark_joints(rig)
B = 64
cons, th0, _ = make_problem(rig, lm, lm, B, seed=777, perturb=0.3)
for lam, ls in ((1e-7, 2), (1e-5, 2), (1e-5, 0)):
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=ls)
    r64 = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=16)
    r32 = orc.solve_batch(rig, cons, th0, opt, dtype="f32", nthreads=16)
    print(f"== lambda {lam} line search {ls}: double final err median {np.median(r64['error']):.2e} max {r64['error'].max():.2e}; float oracle median {np.median(r32['error']):.2e} max {np.nanmax(r32['error']):.2e} status {np.bincount(r32['status'], minlength=3)}")
    for route in ("fused", "wide", "explicit_jacobian"):
        for steps in (0, -1, 1):
            pb = capi.Problem(capi.RigHandle(rig, 0), B, lm, lm)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
            pb.set_constraints(t(cons.pos_offset), t(cons.pos_target), t(cons.pos_weight), t(cons.ori_offset), t(cons.ori_target), t(cons.ori_weight))
            pb.set_route(route, steps)
            out = pb.solve(t(th0), opt, want_history=True)
            torch.cuda.synchronize()
            e = out["error"].cpu().numpy(); st = out["status"].cpu().numpy(); h = out["error_history"].cpu().numpy()
            with np.errstate(all="ignore"):
                print(f"  {route:18s} refine {steps:2d}: final err median {np.nanmedian(e):.2e} max {np.nanmax(e):.2e} nonfinite hist {int((~np.isfinite(h)).any(axis=1).sum())} status counts {np.bincount(st, minlength=3)} increasing-error instances {int((np.diff(h, axis=1) > 1e-6 * np.abs(h[:, :-1])).any(axis=1).sum())}")
            if route == "fused" and steps == 0:
                print("   history of instance 0..2 (hip):\n", h[:3], "\n   (double):\n", r64["error_history"][:3])
