"""Experiment: does the wide route gain from two half-batches on two streams (kernels of different stages co-resident on a CU)?
One handle with B instances against two handles with B / 2 each, solved concurrently on two torch streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from momentum_amd._abi import GnOptions
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
rig, parents, _, rule, _ = bench.build_rig(cfg)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=rule)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
one = bench.DeviceBatch(rig, parents, B, 0, 1)
t1 = timed(lambda: one.pb.solve(one.theta0.clone(), opt))
print(f"{cfg}: one handle, B = {B}: {B / t1:.4g} solves/s ({1e3 * t1:.2f} ms)")
del one
for parts in (2, 4):
    hs = [bench.DeviceBatch(rig, parents, B // parts, 0, 10 + i) for i in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    ths = [h.theta0.clone() for h in hs]
    def run():
        for h, s, th in zip(hs, streams, ths):
            with torch.cuda.stream(s):
                th.copy_(h.theta0)
                h.pb.solve(th, opt)
    t2 = timed(run)
    print(f"{cfg}: {parts} handles x {B // parts} on {parts} streams: {B / t2:.4g} solves/s ({1e3 * t2:.2f} ms)")
    del hs
