"""Where the wide route overtakes the one-launch solve: the 72-joint humanoid with 219 parameters, random subsets of m
enabled parameters (the solve list is what the constraints reach of them), both routes pinned, B = 4096, 10 iterations.
Run on the GPU box: python scripts/route_crossover.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from momentum_amd._abi import GnOptions

rig, parents, _, rule, _ = bench.build_rig("cfg2_all")
B = 4096
db = bench.DeviceBatch(rig, parents, B, 0, 4242)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
rng = np.random.default_rng(1)
print("enabled  solved  blocks   fused solves/s   wide solves/s   tiles")
for m in (100, 120, 135, 150, 165, 180, 200, 219):
    en = np.zeros(rig.num_params, np.uint8)
    en[rng.choice(rig.num_params, size=m, replace=False)] = 1
    en[:7] = 1
    db.pb.set_enabled(en)
    n = bench.solved_parameters(db.pb)
    row = []
    for route in ("fused", "wide"):
        try:
            db.pb.set_route(route)
            for _ in range(2):
                db.pb.solve(db.theta0.clone(), opt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                db.pb.solve(db.theta0.clone(), opt)
            torch.cuda.synchronize()
            row.append(B * 5 / (time.perf_counter() - t0))
        except Exception as e:
            row.append(float("nan"))
    db.pb.set_route("wide")
    ts = db.pb.tile_structure()
    print(f"{int(en.sum()):7d} {n:7d} {(n + 15) // 16:7d}   {row[0]:14.4g}  {row[1]:14.4g}   {ts['tiles']}/{ts['dense_tiles']}  products {ts['products']}/{ts['dense_products']}")
