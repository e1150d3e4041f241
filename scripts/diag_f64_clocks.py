"""Per-phase cycles of workgroup 0 of solveF64Kernel (library variant f64clk: MMX_LIB=.../libmmx_hip_f64clk.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from momentum_amd._abi import GnOptions
rig, parents, _, rule, _ = bench.build_rig("cfg2")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
db = bench.DeviceBatch(rig, parents, B, 0, 12345)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
th = db.theta0.double().clone()
out = db.pb.solve_f64(th, opt, want_history=True)
torch.cuda.synchronize()
h = out["error_history"][0].cpu().numpy()
names = ["FK", "units + error", "J chunks: rest", "parameter rows", "factor", "solve", "update / bookkeeping", "loop top", "J chunks: assembly (n > 96)", "units: assembly + g, H accumulation"]
tot = h[:10].sum()
print(f"B = {B}: cycles of workgroup 0 over 10 iterations: total {tot:.0f}")
for n_, v in zip(names, h[:10]):
    print(f"  {n_:24s} {v:12.0f}  {100 * v / tot:5.1f} %")
