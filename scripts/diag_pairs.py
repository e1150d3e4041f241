"""tiledFactorPairs against the single-column factor on random wide rigs: prints n, NB and the difference of one iteration."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momentum_amd import capi
from momentum_amd._abi import GnOptions
from tests.helpers import make_problem
from tests.test_gpu_fuzz import random_rig

for seed in [int(x) for x in os.environ.get("DIAG_SEEDS", "14").split(",")]:
    rng = np.random.default_rng(9000 + seed)
    J = int(rng.integers(100, 170))
    rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
    P = rig.num_params
    Kp, Ko = int(rng.integers(20, 70)), int(rng.integers(4, 24))
    pp = rng.integers(0, J, size=Kp).astype(np.int32)
    op = rng.integers(0, J, size=Ko).astype(np.int32)
    B = 3
    cons, th0, ths = make_problem(rig, pp, op, B, seed=seed, perturb=0.2, random_offsets=True, weights="random")
    pb = capi.Problem(capi.RigHandle(rig, 0), B, pp, op)
    t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
    pb.set_constraints(t(cons.pos_offset, (B, Kp, 3)), t(cons.pos_target, (B, Kp, 3)), t(cons.pos_weight, (B, Kp)),
                       t(cons.ori_offset, (B, Ko, 4)), t(cons.ori_target, (B, Ko, 4)), t(cons.ori_weight, (B, Ko)))
    opt = GnOptions.make(min_iterations=1, max_iterations=1, regularization=0.5)
    from oracle import oracle as orc
    ref = orc.solve_batch(rig, cons, th0, opt, dtype="f64")
    den = np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
    outs = []
    for name, env in (("tree", {}), ("tree_norefine", {"MMX_NO_REFINE": "1"}), ("dense_norefine", {"MMX_NO_REFINE": "1", "MMX_TREE_REFINE": "0"}), ("rowmajor_norefine", {"MMX_NO_REFINE": "1", "MMX_TREE_ROWMAJOR": "1"}), ("densene_norefine", {"MMX_NO_REFINE": "1", "MMX_TREE_REFINE": "0", "MMX_TREE_NE": "0"})):
        for k in ("MMX_CHOL_PAIRS", "MMX_TREE_REFINE", "MMX_TREE_NE", "MMX_NO_REFINE", "MMX_TREE_ROWMAJOR"):
            os.environ.pop(k, None)
        os.environ.update(env)
        if os.environ.get("DIAG_SET_ENABLED"):
            pb.set_enabled(np.ones(P, np.uint8))
        o = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=bool(os.environ.get("DIAG_HISTORY")))
        th = o["theta"].cpu().numpy()
        outs.append(th)
        print("   ", name, "rel vs oracle", np.linalg.norm(th - ref["theta"], axis=1) / den, "status", o["status"].cpu().numpy())
    import ctypes as C
    buf, nn = np.zeros(P, np.int32), C.c_int32(0)
    capi._check(capi.lib().mmx_debug_fused_normal_equations(pb._h, None, None, None, capi.as_ptr(buf, C.c_int32), C.byref(nn), None)) if False else None
    d = np.abs(outs[0] - outs[-1]).max(axis=1)
    print("seed", seed, "J", J, "P", P, "Kp", Kp, "Ko", Ko, "U", Kp + 3 * Ko, "shape", ["chain", "star", "bushy"][seed % 3], "max |theta_pairs - theta_single| per instance", d)
