"""cfg3 (LM schedule), both routes: quantiles of the precision estimate, its ratio to the measured error and the number of marked elements
(-> profiles/rNN_precision_estimate.txt, first lines)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from momentum_amd._abi import MMX_STEP_LM_SCHEDULE, GnOptions
rig, parents, _, _, _ = bench.build_rig("cfg3")
db = bench.DeviceBatch(rig, parents, 8192, 0, 424242)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE)
for route in ("fused","wide"):
    db.pb.set_route(route)
    out = db.pb.solve(db.theta0.clone(), opt)
    d = db.pb.solve_diagnostics().cpu().numpy()
    st = out["status"].cpu().numpy()
    print("cfg3 LM", route, "est quantiles [50 99 100]", np.quantile(d[:,0],[.5,.99,1.0]), "ratio median", np.median(d[:,1]), "marked", int((st&8!=0).sum()), "of", len(st))
