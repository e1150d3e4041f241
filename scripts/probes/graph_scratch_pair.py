"""Which pairs of solves can share one captured HIP graph?  (round 6: MMX_PRECISION_AUTO with elements in BOTH its second and third
pass aborts on replay; each pass alone replays fine.)  usage: python scripts/probes/graph_scratch_pair.py <first> <second>
with first / second in f32 | mixed | f64 -> prints OK or dies."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from momentum_amd import capi, humanoid72_landmark_joints, make_humanoid72  # noqa: E402
from momentum_amd._abi import MMX_PRECISION_AUTO, MMX_PRECISION_F64, MMX_PRECISION_MIXED, GnOptions  # noqa: E402
from tests.helpers import make_problem  # noqa: E402

B = 128
rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
lm = humanoid72_landmark_joints(rig)
cons, th0, _ = make_problem(rig, lm, lm, B, seed=4242, perturb=0.3)
pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                   t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))  # fmt: skip
prec = {"f32": 0, "mixed": MMX_PRECISION_MIXED, "f64": MMX_PRECISION_F64}
if sys.argv[1] == "auto":  # auto <lambda> <precision_bound>: one MMX_PRECISION_AUTO solve
    opts = [GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=float(sys.argv[2]), precision=MMX_PRECISION_AUTO, precision_bound=float(sys.argv[3]))]
else:
    opts = [GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, precision=prec[a]) for a in sys.argv[1:3]]
dev = pb.device
outs = [dict(error=torch.empty((B,), dtype=torch.float64, device=dev), iterations=torch.empty((B,), dtype=torch.int32, device=dev),
             status=torch.empty((B,), dtype=torch.int32, device=dev)) for _ in opts]  # fmt: skip
if os.environ.get("PROBE_HISTORY"):
    for o in outs:
        o["error_history"] = torch.empty((B, 10), dtype=torch.float64, device=dev)
thetas = [torch.from_numpy(th0.copy()).to(dev) for _ in opts]
src = torch.from_numpy(th0.copy()).to(dev)
for o, out, th in zip(opts, outs, thetas):  # warm-up
    pb.solve(th, o, outputs=out)
torch.cuda.synchronize()
st = outs[0]["status"].cpu().numpy()
print("eager: mixed", int((st & 32 != 0).sum()), "f64", int((st & 16 != 0).sum()), "suspect", int((st & 8 != 0).sum()), flush=True)
ref = [th.clone() for th in thetas]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for o, out, th in zip(opts, outs, thetas):
            th.copy_(src)
            pb.solve(th, o, outputs=out)
torch.cuda.current_stream().wait_stream(side)
for _ in range(3):
    g.replay()
    torch.cuda.synchronize()
# (the warm-up solved from th0 in place; the graph from the same th0)
print(sys.argv[1:3], "OK", [bool(torch.equal(a, b)) for a, b in zip(thetas, ref)], flush=True)
