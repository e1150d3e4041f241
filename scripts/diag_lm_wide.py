"""cfg3's LM schedule on the wide route: the instance(s) that leave the bound although their error history follows the
double run -- histories of the wide route, the fused route and the double oracle side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from momentum_amd._abi import GnOptions, MMX_STEP_LM_SCHEDULE
from oracle import oracle as o

rig, parents, _, _, _ = bench.build_rig("cfg3")
B, n = 8192, 1024
db = bench.DeviceBatch(rig, parents, B, 0, 99)
cons = db.host_constraints(n)
res = {}
for its in (10, 11):
    opt = GnOptions.make(min_iterations=its, max_iterations=its, threshold=1.0, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE)
    ref = o.solve_batch(rig, cons, np.zeros((n, rig.num_params), np.float32), opt, dtype="f64", nthreads=bench.usable_cores())
    for route in ("fused", "wide"):
        db.pb.set_route(route)
        out = db.pb.solve(db.theta0.clone(), opt, want_history=True)
        torch.cuda.synchronize()
        res[(route, its)] = (out["theta"][:n].cpu().numpy().astype(np.float64), out["error_history"][:n].cpu().numpy(), out["status"][:n].cpu().numpy())
    res[("ref", its)] = (ref["theta"], ref["error_history"], ref["status"])
rel = lambda a, r: np.linalg.norm(a - r, axis=1) / np.linalg.norm(r, axis=1)
for route in ("fused", "wide"):
    r10 = rel(res[(route, 10)][0], res[("ref", 10)][0])
    h, href = res[(route, 11)][1], res[("ref", 11)][1]
    same = np.all(np.abs(h - href) <= 1e-3 * np.abs(href) + 1e-7 * href[:, :1], axis=1)
    bad = np.flatnonzero(same & (r10 > 1e-5))
    print(route, "same path", same.mean(), "max rel on same path", r10[same].max(), "offenders", bad, "status bits", np.unique(res[(route, 10)][2]))
    for b in bad[:3]:
        print("  instance", b, "rel", r10[b])
        print("   hip ", np.array2string(h[b], precision=6))
        print("   ref ", np.array2string(href[b], precision=6))
        other = "fused" if route == "wide" else "wide"
        print("   ", other, np.array2string(res[(other, 11)][1][b], precision=6), "rel", rel(res[(other, 10)][0], res[("ref", 10)][0])[b])
