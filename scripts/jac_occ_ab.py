#!/usr/bin/env python3
"""A/B of the J-assembly kernel's occupancy at the BASELINE batch sizes: kernel duration (events on the dispatch packet)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
for config, B in (("cfg2", 4096), ("cfg2", 3072), ("cfg2", 5120), ("cfg2", 32768), ("cfg5", 8192)):
    rig, parents, _, _, _ = bench.build_rig(config)
    db = bench.DeviceBatch(rig, parents, B, 0, 12345)
    pb = db.pb
    jac = torch.empty((B, pb.P, pb.M), dtype=torch.float32, device=pb.device)
    res = torch.empty((B, pb.M), dtype=torch.float32, device=pb.device)
    err = torch.empty((B,), dtype=torch.float64, device=pb.device)
    for _ in range(3):
        pb.eval_jacobian(db.theta_star, jac, res, err)
    torch.cuda.synchronize()
    ms = [pb.eval_jacobian_kernel_ms(db.theta_star, jac, res, err) for _ in range(20)]
    nbytes = B * bench.algorithmic_bytes_per_instance(pb.M, pb.P, len(parents[0]), len(parents[1]))
    print(f"{os.environ.get('MMX_LIB','main').split('_')[-1]:12s} {config} B={B}: {np.mean(ms)*1e3:.1f} us (min {np.min(ms)*1e3:.1f})  {nbytes/np.mean(ms)/1e6:.0f} GB/s = {nbytes/np.mean(ms)/1e6/8000:.3f}")
    del db, jac
    torch.cuda.empty_cache()
