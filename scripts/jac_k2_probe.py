#!/usr/bin/env python3
"""Times the J-assembly at B = 4096 (cfg2): default two-kernel form; with MMX_JAC_SKIP_K1 the column kernel alone
on the hand-over data of an earlier launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
rig, parents, _, _, _ = bench.build_rig("cfg2")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
db = bench.DeviceBatch(rig, parents, B, 0, 1)
pb = db.pb
jac = torch.empty((B, pb.P, pb.M), dtype=torch.float32, device=pb.device)
res = torch.empty((B, pb.M), dtype=torch.float32, device=pb.device)
err = torch.empty((B,), dtype=torch.float64, device=pb.device)
for _ in range(3):
    pb.eval_jacobian(db.theta_star, jac, res, err)
torch.cuda.synchronize()
ms = [pb.eval_jacobian_kernel_ms(db.theta_star, jac, res, err) for _ in range(20)]
print("B=%d  %s  %.1f us (min %.1f)" % (B, "K2 only" if os.environ.get("MMX_JAC_SKIP_K1") else ("one kernel" if os.environ.get("MMX_JAC_ONE_KERNEL") else "K1+K2"), 1e3 * np.mean(ms), 1e3 * np.min(ms)))
