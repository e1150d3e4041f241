"""Child process of bench.py --measure-traffic: launches mmx_eval_jacobian a few times on one BASELINE shape so that a
rocprofv3 --pmc pass around it sees the graded kernel (fkJacobianKernel<true>).  python scripts/pmc_traffic.py cfg2 4096"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402

config, B = sys.argv[1], int(sys.argv[2])
rig, parents, _, _, _ = bench.build_rig(config)
db = bench.DeviceBatch(rig, parents, B, 0, 12345)
pb = db.pb
jac = torch.empty((B, pb.P, pb.M), dtype=torch.float32, device=pb.device)
res = torch.empty((B, pb.M), dtype=torch.float32, device=pb.device)
err = torch.empty((B,), dtype=torch.float64, device=pb.device)
for _ in range(4):
    pb.eval_jacobian(db.theta_star, jac, res, err)
torch.cuda.synchronize()
