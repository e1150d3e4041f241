"""Calibration data of the precision estimate (mmx_problem_solve_diagnostics) and of the LM schedule's branch analysis.

    python scripts/diag_precision.py estimate   -> gpurun_out/precision_estimate.npz
    python scripts/diag_precision.py lm [B] [n] -> gpurun_out/lm_steps.npz

estimate: BASELINE configs[0] / configs[1] shapes and the well-determined all-joints variant, lambda in {5e-2, 1e-2, 1e-3,
1e-5}, without and with the driver's line search: per instance the single-precision solve's diagnostics, its distance from
the oracle's double run, and whether that double run converged.  lm: the cfg3 batch bench.py times (seed 424242): the GPU's
and the double oracle's (lambda, gain ratio) per iteration.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from momentum_amd import capi, humanoid72_landmark_joints, make_humanoid72, make_test_character  # noqa: E402
from momentum_amd._abi import GnOptions  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import make_problem  # noqa: E402

UNIT = 0.01
OUT = os.path.join(ROOT, "gpurun_out")


def rel(a, ref):
    return np.linalg.norm(a - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)


def problem_on_gpu(rig, cons, B):
    pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                       t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)))  # fmt: skip
    return pb


def estimate():
    cores = bench.usable_cores()
    h72 = make_humanoid72(seed=12345, variant="p128", unit=UNIT)
    lm = humanoid72_landmark_joints(h72)
    allj = list(range(h72.num_joints))
    shapes = {
        "cfg1": (make_test_character(24), [23, 12, 5], [], 777),
        "cfg2": (h72, lm, lm, 777),
        "p128_all": (h72, allj, allj, 31337),
    }
    B = 1024
    res = {}
    for name, (rig, pp, op, seed) in shapes.items():
        cons, th0, _ = make_problem(rig, pp, op, B, seed=seed, perturb=0.3)
        e0 = np.array([orc.get_error(rig, cons.instance(b), th0[b].astype(np.float64), "f64") for b in range(0, B, 64)]).max()
        for route in ("fused", "wide"):
            pb = problem_on_gpu(rig, cons, B)
            pb.set_route(route)
            for lam in (5e-2, 1e-2, 1e-3, 1e-5):
                for ls in (0, 2):
                    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=lam, do_line_search=ls)
                    try:
                        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt)
                        diag = pb.solve_diagnostics().cpu().numpy()
                    except capi.MmxError as e:
                        print(name, route, lam, ls, "skipped:", e)
                        continue
                    torch.cuda.synchronize()
                    with np.errstate(all="ignore"):
                        r64 = orc.solve_batch(rig, cons, th0, opt, dtype="f64", nthreads=cores)
                    sane = (r64["status"] == 0) & np.isfinite(r64["theta"]).all(axis=1) & (r64["error"] <= e0)
                    r = rel(out["theta"].cpu().numpy().astype(np.float64), r64["theta"])
                    key = f"{name}|{route}|{lam:g}|{ls}"
                    res[key + "|rel"], res[key + "|diag"], res[key + "|sane"] = r, diag, sane
                    res[key + "|status"] = out["status"].cpu().numpy()
                    ab = sane & ~(r <= 1e-5)
                    est = diag[:, 0]
                    print(f"{key:32s} sane {int(sane.sum()):4d} above {int(ab.sum()):4d} | est: min over above {est[ab].min() if ab.any() else float('nan'):9.3g}"
                          f"  quantiles of within [50 90 99 100] {np.quantile(est[sane & ~ab], [0.5, 0.9, 0.99, 1.0]) if (sane & ~ab).any() else None}"
                          f" | minpiv(above) max {diag[ab, 1].max() if ab.any() else float('nan'):.3g} minpiv(within) min {diag[sane & ~ab, 1].min() if (sane & ~ab).any() else float('nan'):.3g}"
                          f" floored {int((res[key + '|status'] & 4 != 0).sum())}", flush=True)  # fmt: skip
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "precision_estimate.npz"), **res)


def lm(B=65536, n=16384):
    from momentum_amd._abi import MMX_STEP_LM_SCHEDULE

    rig, parents, _, _, _ = bench.build_rig("cfg3")
    db = bench.DeviceBatch(rig, parents, B, 0, 424242)
    opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05, step_rule=MMX_STEP_LM_SCHEDULE)
    out = db.pb.solve(db.theta0.clone(), opt, want_history=True, want_step_history=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = orc.solve_batch(rig, db.host_constraints(n), db.theta0[:n].cpu().numpy(), opt, dtype="f64", nthreads=bench.usable_cores(), step_history=True)
    r32 = orc.solve_batch(rig, db.host_constraints(n), db.theta0[:n].cpu().numpy(), opt, dtype="f32", nthreads=bench.usable_cores(), step_history=True)
    print(f"oracle: {n} instances x 2 precisions in {time.perf_counter() - t0:.1f} s")
    sh = out["step_history"][:n].cpu().numpy()
    th = out["theta"][:n].cpu().numpy().astype(np.float64)
    r = rel(th, ref["theta"])
    res = bench.lm_branch_analysis(sh, out["error_history"][:n].cpu().numpy(), ref, r)
    print({k: v for k, v in res.items() if not isinstance(v, (list, np.ndarray))})
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "lm_steps.npz"), gpu_steps=sh, gpu_err=out["error_history"][:n].cpu().numpy(), rel=r,
                        ref_lambda=ref["lambda_history"], ref_rho=ref["gain_ratio_history"], ref_err=ref["error_history"],
                        f32_lambda=r32["lambda_history"], f32_rho=r32["gain_ratio_history"], rel32=rel(r32["theta"].astype(np.float64), ref["theta"]))  # fmt: skip


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "estimate"
    if what == "estimate":
        estimate()
    else:
        lm(*[int(a) for a in sys.argv[2:4]])
