#!/usr/bin/env python3
"""Solves the first N instances of a bench.py workload with the CPU oracle (f64) and compares them
with the GPU solve of the same synthetic batch: relative pose-parameter difference and final errors."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from momentum_amd import humanoid72_landmark_joints, make_humanoid72, make_rig300  # noqa: E402
from momentum_amd._abi import GnOptions  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="cfg5")
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--check", type=int, default=32)
args = ap.parse_args()
variant, which, defB, step_rule, desc = bench.CONFIGS[args.config]
if variant == "rig300":
    rig = make_rig300(seed=12345, unit=bench.UNIT)
    prng = np.random.default_rng(77)
    pos = prng.choice(rig.num_joints, size=150, replace=False).astype(np.int32)
    ori = prng.choice(rig.num_joints, size=50, replace=False).astype(np.int32)
else:
    rig = make_humanoid72(seed=12345, variant=variant, unit=bench.UNIT)
    pos = ori = humanoid72_landmark_joints(rig)
B = args.batch
rh, pb, theta0, theta_star = bench.make_device_problem(rig, (pos, ori), B, 0, 12345)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=step_rule)
out = pb.solve(theta0.clone(), opt)
th = out["theta"].cpu().numpy()
err = out["error"].cpu().numpy()
print("GPU final error: sum %.4f  median %.3e  max %.3e  (#instances with error > 100 x median: %d)" % (err.sum(), np.median(err), err.max(), int((err > 100 * np.median(err)).sum())))
N = args.check
st = pb.skeleton_state(theta_star)
Kp, Ko = len(pos), len(ori)
pt = st[:N, torch.as_tensor(pos.astype(np.int64), device=pb.device), 0:3].cpu().numpy()
ot = st[:N, torch.as_tensor(ori.astype(np.int64), device=pb.device), 3:7].cpu().numpy()
oo = np.zeros((N, Ko, 4), np.float32)
oo[..., 3] = 1
cons = orc.Constraints(pos, np.zeros((N, Kp, 3), np.float32), pt, np.ones((N, Kp), np.float32), ori, oo, ot, np.ones((N, Ko), np.float32))
ref = orc.solve_batch(rig, cons, np.zeros((N, rig.num_params), np.float32), opt, dtype="f64", nthreads=16)
rel = np.linalg.norm(th[:N] - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
print("first %d instances vs oracle f64: max rel theta diff %.3e, median %.3e ; max |err - err_ref| / err_ref %.3e" % (N, rel.max(), np.median(rel), np.max(np.abs(err[:N] - ref["error"]) / np.maximum(ref["error"], 1e-12))))
worst = np.argsort(-rel)[:4]
for i in worst:
    print("  instance %d: rel %.3e, gpu err %.4e, oracle err %.4e" % (i, rel[i], err[i], ref["error"][i]))
