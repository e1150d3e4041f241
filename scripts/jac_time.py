#!/usr/bin/env python3
"""Times the J-assembly kernel and its FK-only variant (mmx_eval_skeleton_state) at several batch sizes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from momentum_amd import humanoid72_landmark_joints, make_humanoid72  # noqa: E402

rig = make_humanoid72(seed=12345, variant="p128", unit=bench.UNIT)
parents = humanoid72_landmark_joints(rig)
for B in [int(x) for x in (sys.argv[1:] or ["4096", "16384", "65536"])]:
    rh, pb, theta0, theta_star = bench.make_device_problem(rig, parents, B, 0, 12345)
    jac = torch.empty((B, pb.P, pb.M), dtype=torch.float32, device=pb.device)
    res = torch.empty((B, pb.M), dtype=torch.float32, device=pb.device)
    err = torch.empty((B,), dtype=torch.float64, device=pb.device)
    nbytes = B * bench.algorithmic_bytes_per_instance(pb.M, pb.P, len(parents), len(parents))

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    tj = timeit(lambda: pb.eval_jacobian(theta_star, jac, res, err))
    tf = timeit(lambda: pb.skeleton_state(theta_star))
    print(f"B={B}: J-assembly {tj:.1f} us = {nbytes / tj / 1e3:.0f} GB/s ; FK-only {tf:.1f} us")
    del jac, res, err, pb, rh
