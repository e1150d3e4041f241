"""Round 6: the same table with MMX_PRECISION_MIXED beside it (gpurun_out/mixed_table.json -> profiles/r06_weak_damping.json).
The precision policy's table (profiles/r05_weak_damping.json, part "auto"): BASELINE configs[0] / configs[1] shapes x
lambda {1e-2, 1e-3, 1e-5} x {no line search, the batched driver's}: what single precision holds, what it marks
(MMX_SOLVE_PRECISION_SUSPECT), what MMX_PRECISION_AUTO returns and at which rate.

    python scripts/diag_auto_table.py [B]   -> gpurun_out/auto_table.json

Per row: instances whose double run is sane (finite, converging) and STABLE (the oracle's double run from theta0 + 1e-12 ends
within 1e-7 of its run from theta0: without a line search the undamped iteration is chaotic in double on some of these
marginally determined instances -- such instances are counted, not compared), and among them: above 1e-5 in single
precision, marked, above and not marked (must be 0), above under AUTO on the instances whose line-search decisions are the
double run's (must be 0), escalated, and the three rates (single, AUTO, double) on the same batch."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from momentum_amd import capi, humanoid72_landmark_joints, make_humanoid72, make_test_character  # noqa: E402
from momentum_amd._abi import MMX_PRECISION_AUTO, MMX_PRECISION_F64, MMX_PRECISION_MIXED, GnOptions  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import make_problem  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rel = lambda a, ref: np.linalg.norm(a - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
cores = bench.usable_cores()
h72 = make_humanoid72(seed=12345, variant="p128", unit=0.01)
lm = humanoid72_landmark_joints(h72)
shapes = {"cfg1": (make_test_character(24), [23, 12, 5], []), "cfg2": (h72, lm, lm)}
table = {}
for name, (rig, pp, op) in shapes.items():
    cons, th0, _ = make_problem(rig, pp, op, B, seed=777, perturb=0.3)
    e0 = np.array([orc.get_error(rig, cons.instance(b), th0[b].astype(np.float64), "f64") for b in range(0, B, 64)]).max()
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))  # fmt: skip
    th0d = torch.from_numpy(th0.copy()).to(pb.device)
    if os.environ.get("MMX_MIXED_TOL") or os.environ.get("MMX_MIXED_MAXCG"):
        pb.set_mixed(float(os.environ.get("MMX_MIXED_TOL", "0")), int(os.environ.get("MMX_MIXED_MAXCG", "0")))

    def run(opt, reps=5):
        out = pb.solve(th0d.clone(), opt, want_history=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            pb.solve(th0d.clone(), opt)
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in out.items() if v is not None}, B * reps / (time.perf_counter() - t0)

    for lam in (5e-2, 1e-2, 1e-3, 1e-5):
        for ls in (0, 2):
            mk = lambda prec: GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=ls, precision=prec)
            f32, r32 = run(mk(0))
            auto, rauto = run(mk(MMX_PRECISION_AUTO))
            mix, rmix = run(mk(MMX_PRECISION_MIXED))
            cg_per_it = float(np.mean(pb.solve_diagnostics().cpu().numpy()[:, 2]))
            _, r64 = run(mk(MMX_PRECISION_F64))
            with np.errstate(all="ignore"):
                ref = orc.solve_batch(rig, cons, th0, mk(0), dtype="f64", nthreads=cores)
                pert = orc.solve_batch(rig, cons, th0.astype(np.float64) + 1e-12, mk(0), dtype="f64", nthreads=cores)
                sane = (ref["status"] == 0) & np.isfinite(ref["theta"]).all(axis=1) & (ref["error"] <= e0)
                stable = rel(pert["theta"], ref["theta"]) <= 1e-7
            ok = sane & stable
            r_f32, r_auto = rel(f32["theta"].astype(np.float64), ref["theta"]), rel(auto["theta"].astype(np.float64), ref["theta"])
            r_mix = rel(mix["theta"].astype(np.float64), ref["theta"])
            suspect, esc = f32["status"] & 8 != 0, auto["status"] & (16 | 32) != 0
            href = ref["error_history"]
            same = lambda h, tol: np.all(np.abs(h - href) <= tol * np.abs(href) + 1e-7 * href[:, :1], axis=1) if ls else np.ones(B, bool)
            same32, sameA = same(f32["error_history"], 1e-3), same(auto["error_history"], np.where(esc[:, None], 1e-6, 1e-3))
            sameM = same(mix["error_history"], 1e-6)
            row = {
                "instances": B, "double_run_sane": int(sane.sum()), "sane_and_stable": int(ok.sum()),
                "f32_above_1e-5": int((ok & same32 & ~(r_f32 <= 1e-5)).sum()), "f32_marked_suspect": int(suspect.sum()),
                "f32_above_and_not_marked": int((ok & same32 & ~(r_f32 <= 1e-5) & ~suspect).sum()),
                "f32_other_line_search_decision": int((ok & ~same32).sum()),
                "mixed_above_1e-5": int((ok & sameM & ~(r_mix <= 1e-5)).sum()), "mixed_max_rel": float(r_mix[ok & sameM].max()) if (ok & sameM).any() else None,
                "mixed_median_rel": float(np.median(r_mix[ok & sameM])) if (ok & sameM).any() else None,
                "mixed_other_line_search_decision": int((ok & ~sameM).sum()), "mixed_cg_unconverged": int((mix["status"] & 8 != 0).sum()), "mixed_operator_applications_per_iteration": cg_per_it,
                "mixed_status_bits": sorted({int(x) for x in np.unique(mix["status"])}),
                "auto_escalated": int(esc.sum()), "auto_escalated_f64": int((auto["status"] & 16 != 0).sum()), "auto_above_1e-5": int((ok & sameA & ~(r_auto <= 1e-5)).sum()),
                "auto_other_line_search_decision": int((ok & ~sameA).sum()), "auto_max_rel": float(r_auto[ok & sameA].max()) if (ok & sameA).any() else None,
                "smallest_pivot_ratio_median": float(np.median(pb.solve_diagnostics().cpu().numpy()[:, 1])),
                "solves_per_s": {"f32": r32, "mixed": rmix, "auto": rauto, "f64": r64},
            }
            table[f"{name} lambda={lam:g} line_search={ls}"] = row
            print(name, lam, ls, {k: v for k, v in row.items() if k != "solves_per_s"}, {k: f"{v:.3g}" for k, v in row["solves_per_s"].items()}, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(table, open(os.path.join(ROOT, "gpurun_out", "mixed_table%s.json" % os.environ.get("MMX_TABLE_TAG", "")), "w"), indent=1, sort_keys=True)
