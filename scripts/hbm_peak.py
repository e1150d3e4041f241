#!/usr/bin/env python3
"""Achievable HBM bandwidth of this box with plain torch kernels: write-only fill, read-only sum, copy."""
import torch

def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

for mb in (412, 1650, 6600):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    y = torch.empty(n, dtype=torch.float32, device="cuda")
    tf = timeit(lambda: x.fill_(1.0))
    tz = timeit(lambda: x.zero_())
    tc = timeit(lambda: y.copy_(x))
    ts = timeit(lambda: x.sum())
    print(f"{mb} MB: fill {4*n/tf/1e9:.0f} GB/s ({tf*1e6:.0f} us), zero_ {4*n/tz/1e9:.0f} GB/s, copy {8*n/tc/1e9:.0f} GB/s (r+w), sum {4*n/ts/1e9:.0f} GB/s")
    del x, y
