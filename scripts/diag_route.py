"""Worst instance of a probe configuration through the fused solve and through the wide path against the double oracle
and the oracle's float instantiation."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from momentum_amd._abi import GnOptions
from oracle import oracle as orc

cfg = sys.argv[1]
rig, parents, _, rule, _ = bench.build_rig(cfg)
B = 256
db = bench.DeviceBatch(rig, parents, B, 0, 20240611)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
cons = db.host_constraints(B)
th0 = db.theta0.cpu().numpy()
ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=bench.usable_cores())
r32 = orc.solve_batch(rig, cons, th0, opt, dtype="f32", nthreads=bench.usable_cores())
den = np.linalg.norm(ref["theta"], axis=1)
rel32 = np.linalg.norm(r32["theta"] - ref["theta"], axis=1) / den
np.set_printoptions(linewidth=220, precision=4)
for route in ("fused", "wide"):
    os.environ.pop("MMX_FORCE_WIDE", None); os.environ.pop("MMX_PREFER_FUSED", None)
    os.environ["MMX_FORCE_WIDE" if route == "wide" else "MMX_PREFER_FUSED"] = "1"
    out = db.pb.solve(db.theta0.clone(), opt, want_history=True)
    th = out["theta"].cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / den
    w = np.argsort(-rel)[:3]
    print(route, "worst", w, rel[w], "float oracle there", rel32[w])
    print("   history gpu", out["error_history"][w[0]].cpu().numpy())
    print("   history f64", ref["error_history"][w[0]])
print("float oracle: max", rel32.max(), "above 1e-5:", int((rel32 > 1e-5).sum()))
