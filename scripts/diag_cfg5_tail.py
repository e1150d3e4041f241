"""cfg5, 4096 instances of one seed: the instances above the parity bound with the double oracle's error history on them."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from momentum_amd._abi import GnOptions
from oracle import oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20240611
rig, parents, _, rule, _ = bench.build_rig("cfg5")
db = bench.DeviceBatch(rig, parents, 4096, 0, seed)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=rule)
th = db.theta0.clone()
db.pb.solve(th, opt)
torch.cuda.synchronize()
chk = bench.parity_check(db, th, opt, 4096)
print(json.dumps({k: v for k, v in chk.items() if not k.endswith("rule") and k != "reference"}, indent=1))
idx = chk.get("above_bound_instances", [])
if idx:
    cons = db.host_constraints(4096).subset(np.asarray(idx))
    th0 = db.theta0[:4096].cpu().numpy()[idx]
    ref = orc.solve_batch(db.rig, cons, th0, opt, dtype="f64")
    for k, i in enumerate(idx):
        print(i, " ".join(f"{e:.3e}" for e in ref["error_history"][k]), "final", f"{ref['error'][k]:.3e}")
    # all instances: distribution of the final error of the double run
    refall = orc.solve_batch(db.rig, db.host_constraints(4096), db.theta0[:4096].cpu().numpy(), opt, dtype="f64", nthreads=bench.usable_cores())
    e = refall["error"]
    print("final error quantiles 50/90/99/99.9/max:", [float(f"{x:.3e}") for x in np.quantile(e, [0.5, 0.9, 0.99, 0.999, 1.0])])
    print("instances with final error > 5e-3:", np.nonzero(e > 5e-3)[0].tolist(), e[e > 5e-3].tolist())
