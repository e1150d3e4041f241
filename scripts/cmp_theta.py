"""Bit-compare the solve of one configuration between two builds of the library: python scripts/cmp_theta.py save|check cfg B file
[route] (run once per library, MMX_LIB selects it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
import bench
from momentum_amd._abi import GnOptions
mode, cfg, B, path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
route = sys.argv[5] if len(sys.argv) > 5 else None
rig, parents, _, rule, _ = bench.build_rig(cfg)
db = bench.DeviceBatch(rig, parents, B, 0, 20240611)
if route:
    db.pb.set_route(route)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=rule)
th = db.theta0.clone()
out = db.pb.solve(th, opt, want_history=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    db.pb.solve(db.theta0.clone(), opt)
torch.cuda.synchronize()
print(f"{cfg} B = {B}: {3 * B / (time.perf_counter() - t0):.4g} solves/s, route {db.pb.last_route()}")
a = th.cpu().numpy()
if mode == "save":
    np.save(path, a)
else:
    ref = np.load(path)
    print("bit-identical:", np.array_equal(a.view(np.uint32), ref.view(np.uint32)), "max abs diff", np.abs(a - ref).max())
