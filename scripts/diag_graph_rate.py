"""Eager against graph-replayed solves at small batches (launch-bound on the wide route: some eighty short launches per solve).
    python scripts/diag_graph_rate.py  ->  rows: config, batch, route, eager solves/s, graph-replay solves/s"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from momentum_amd._abi import MMX_PRECISION_AUTO, GnOptions  # noqa: E402

for cfg, route, kw in (("cfg5", "auto", {}), ("cfg2", "wide", {}), ("cfg2", "fused", {}), ("cfg2", "fused", dict(precision=MMX_PRECISION_AUTO))):
    rig, parents, _, _, _ = bench.build_rig(cfg)
    for B in (16, 64, 256, 1024):
        db = bench.DeviceBatch(rig, parents, B, 0, 424242)
        pb = db.pb
        pb.set_route(route)
        opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, **kw)
        dev = pb.device
        out = dict(error=torch.empty((B,), dtype=torch.float64, device=dev), iterations=torch.empty((B,), dtype=torch.int32, device=dev), status=torch.empty((B,), dtype=torch.int32, device=dev))
        theta = db.theta0.clone()

        def step():
            theta.copy_(db.theta0)
            pb.solve(theta, opt, outputs=out)

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        eager = B * reps / (time.perf_counter() - t0)
        ref = theta.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                step()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        graph = B * reps / (time.perf_counter() - t0)
        print(f"{cfg} B={B:5d} route={pb.last_route():6s} {'auto' if kw else 'f32 '} eager {eager:10.4g} solves/s   graph replay {graph:10.4g} solves/s   x{graph / eager:.2f}   same result {bool(torch.equal(theta, ref))}", flush=True)
