"""Run-to-run determinism of mmx_solve: the same batch solved `reps` times must give bit-identical theta and error histories.
python scripts/diag_determinism.py [reps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from momentum_amd import make_humanoid72, capi, humanoid72_landmark_joints
from momentum_amd._abi import GnOptions
from tests.helpers import make_problem

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rig = make_humanoid72(seed=12345, variant="p128", unit=0.01)
allj = list(range(rig.num_joints)); lm = humanoid72_landmark_joints(rig); B = 1024
for name, pp in (("all_joints(NB=8)", allj), ("landmarks(NB=6)", lm)):
    cons, th0, _ = make_problem(rig, pp, pp, B, seed=31337, perturb=0.3)
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)), t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))
    pb.set_route("fused")
    for rule in (0, 1):
        for ls in ((0, 1, 2) if rule == 0 else (0,)):
            for lam in (1e-7, 1e-3, 0.05):
                opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=ls, step_rule=rule)
                ref = None; bad = 0; first = None
                for r in range(reps):
                    out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
                    torch.cuda.synchronize()
                    h, th = out["error_history"].cpu().numpy(), out["theta"].cpu().numpy()
                    if ref is None:
                        ref = (h, th)
                    else:
                        ne = lambda x, y: ~((x == y) | (np.isnan(x) & np.isnan(y)))  # (a diverged run's NaNs are reproducible too)
                        d = ne(h, ref[0]).any(axis=1) | ne(th, ref[1]).any(axis=1)
                        if d.any() and first is None:
                            i = int(np.flatnonzero(d)[0]); first = (r, i, int(np.argmax(ne(h[i], ref[0][i]))) if ne(h[i], ref[0][i]).any() else -1)
                        bad += int(d.sum())
                print(f"{name:18s} rule {rule} ls {ls} lambda {lam:g}: {bad} differing instance-solves of {(reps - 1) * B}" + (f"  first: rep {first[0]} instance {first[1]} iteration {first[2]}" if first else ""), flush=True)
