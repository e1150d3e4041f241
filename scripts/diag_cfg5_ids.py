"""cfg5 (wide route), three seeds: which instance ids leave the 1e-5 bound (the ids tests/test_gpu_baseline_parity.py pins)."""
import sys; sys.path.insert(0,'.')
import torch, numpy as np, bench
from momentum_amd._abi import GnOptions
for seed in (20240611, 424242, 7):
    rig, parents, _, rule, _ = bench.build_rig("cfg5")
    db = bench.DeviceBatch(rig, parents, 4096, 0, seed)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
    out = db.pb.solve(db.theta0.clone(), opt); torch.cuda.synchronize()
    chk = bench.parity_check(db, out["theta"], opt, 4096)
    print(seed, chk["num_above_bound"], chk.get("above_bound_instances"), chk.get("above_bound_rel"), chk.get("above_bound_float_oracle_rel"))
