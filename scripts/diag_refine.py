"""What the refinement buys on the BASELINE workloads: parity of the final pose parameters and of the error history, and
the time per solve, with up to three / one / no refinement steps per iteration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from momentum_amd._abi import GnOptions
from oracle import oracle as orc
for config, B, n, its in (("cfg2", 4096, 1024, 10), ("cfg5", 2048, 256, 10), ("cfg2_all", 2048, 256, 10)):
    rig, parents, _, rule, _ = bench.build_rig(config)
    db = bench.DeviceBatch(rig, parents, B, 0, 12345)
    opt = GnOptions.make(min_iterations=its, max_iterations=its, threshold=1.0, regularization=0.05)
    cons = db.host_constraints(n)
    ref = orc.solve_batch(rig, cons, np.zeros((n, rig.num_params), np.float32), opt, dtype="f64", nthreads=bench.usable_cores())
    for steps in (0, 1, -1):
        db.pb.set_route("auto", steps)
        out = db.pb.solve(db.theta0.clone(), opt, want_history=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            db.pb.solve(db.theta0.clone(), opt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        th = out["theta"][:n].cpu().numpy().astype(np.float64)
        rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        h = out["error_history"][:n].cpu().numpy(); href = ref["error_history"]
        hd = np.abs(h - href) / np.maximum(np.abs(href), 1e-7 * href[:, :1])
        print(f"{config} B={B} route {db.pb.last_route()} refine {steps:2d}: {B / dt:.4g} solves/s | theta rel max {rel.max():.2e} median {np.median(rel):.2e} above 1e-5: {(rel > 1e-5).sum()} | error history rel dev: max per iterate {np.array2string(hd.max(axis=0), precision=1)}")
