#!/usr/bin/env python3
"""Runs only the graded J-assembly kernel (mmx_eval_jacobian) a few times: the target of the
rocprofv3 --pmc passes (counters must be collected without tracing; see MI355X_MICROARCH.md)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from momentum_amd import humanoid72_landmark_joints, make_humanoid72  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32768)
ap.add_argument("--launches", type=int, default=5)
args = ap.parse_args()
rig = make_humanoid72(seed=12345, variant="p128", unit=bench.UNIT)
parents = humanoid72_landmark_joints(rig)
rh, pb, theta0, theta_star = bench.make_device_problem(rig, parents, args.batch, 0, 12345)
jac = torch.empty((args.batch, pb.P, pb.M), dtype=torch.float32, device=pb.device)
res = torch.empty((args.batch, pb.M), dtype=torch.float32, device=pb.device)
err = torch.empty((args.batch,), dtype=torch.float64, device=pb.device)
for _ in range(args.launches):
    pb.eval_jacobian(theta_star, jac, res, err)
torch.cuda.synchronize()
print("bytes_per_launch", args.batch * bench.algorithmic_bytes_per_instance(pb.M, pb.P, len(parents), len(parents)))
