"""cfg5 (wide route) with MMX_PHASE_CLOCKS=1: cycles of block 0 per stage over the ten iterations (stderr of the library)."""
import os, sys
os.environ["MMX_PHASE_CLOCKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from momentum_amd._abi import GnOptions
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
rig, parents, _, rule, _ = bench.build_rig(cfg)
db = bench.DeviceBatch(rig, parents, B, 0, 1)
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=rule)
for _ in range(2):
    db.pb.solve(db.theta0.clone(), opt)
torch.cuda.synchronize()
