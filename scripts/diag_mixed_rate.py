"""Rates of the precision routes on BASELINE's shapes (one-launch route): single / mixed / auto / double at
cfg2 @ 4096 and 32768 (lambda 0.05, 0.01, 1e-3), the driver's line search, cfg3 @ 65536 (LM schedule).
    python scripts/diag_mixed_rate.py -> gpurun_out/mixed_rate.json   (MMX_PHASE_CLOCKS=1: the mixed kernel's phase clocks on stderr)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from momentum_amd import capi, humanoid72_landmark_joints, make_humanoid72  # noqa: E402
from momentum_amd._abi import MMX_PRECISION_AUTO, MMX_PRECISION_F64, MMX_PRECISION_MIXED, MMX_STEP_LM_SCHEDULE, GnOptions  # noqa: E402
from tests.helpers import make_problem  # noqa: E402

rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
lm = humanoid72_landmark_joints(rig)
D = 512  # distinct instances, tiled to the batch
cons, th0, _ = make_problem(rig, lm, lm, D, seed=12345, perturb=0.3)
out = {}
clocks = bool(os.environ.get("MMX_PHASE_CLOCKS"))
for B in ((4096,) if clocks else (4096, 32768, 65536)):
    rep = lambda a: np.ascontiguousarray(np.tile(a, (B // D,) + (1,) * (a.ndim - 1)))
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a: torch.from_numpy(rep(np.asarray(a, np.float32))).to(pb.device)
    pb.set_constraints(t(cons.pos_offset), t(cons.pos_target), t(cons.pos_weight), t(cons.ori_offset), t(cons.ori_target), t(cons.ori_weight))
    if os.environ.get("MMX_MIXED_TOL") or os.environ.get("MMX_MIXED_MAXCG"):
        pb.set_mixed(float(os.environ.get("MMX_MIXED_TOL", "0")), int(os.environ.get("MMX_MIXED_MAXCG", "0")))
    th0d = t(th0)
    cases = [("gn lambda=0.05", dict(regularization=0.05)), ("gn lambda=0.01", dict(regularization=0.01)), ("gn lambda=1e-3", dict(regularization=1e-3)),
             ("driver line search lambda=0.01", dict(regularization=0.01, do_line_search=2)), ("lm schedule", dict(regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE))]  # fmt: skip
    if clocks:
        cases = cases[:1]
    for name, kw in cases:
        row = {}
        for pname, prec in (("f32", 0), ("mixed", MMX_PRECISION_MIXED), ("auto", MMX_PRECISION_AUTO), ("f64", MMX_PRECISION_F64)):
            if clocks and pname != "mixed":
                continue
            if pname == "f64" and B > 4096:
                continue
            opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, precision=prec, **kw)
            th = th0d.clone()
            pb.solve(th, opt)
            torch.cuda.synchronize()
            reps = 1 if clocks else (5 if B <= 4096 else 2)
            t0 = time.perf_counter()
            for _ in range(reps):
                th.copy_(th0d)
                o = pb.solve(th, opt)
            torch.cuda.synchronize()
            row[pname] = B * reps / (time.perf_counter() - t0)
            if pname == "mixed":
                row["mixed_cg_per_iteration"] = float(pb.solve_diagnostics()[:, 2].mean().item())
            if pname == "auto":
                st = o["status"].cpu().numpy()
                row["auto_mixed"], row["auto_f64"] = int((st & 32 != 0).sum()), int((st & 16 != 0).sum())
        out[f"B={B} {name}"] = row
        print(B, name, {k: (f"{v:.3g}" if isinstance(v, float) else v) for k, v in row.items()}, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
if not clocks:
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mixed_rate.json"), "w"), indent=1)
