"""The reference's 3-joint IK known answer through every route: statuses, error histories, parameters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from momentum_amd import capi, make_test_character
from momentum_amd._abi import GnOptions
from oracle import oracle as orc
np.set_printoptions(linewidth=250, precision=4)
rig = make_test_character(3)
P = rig.num_params
rng = np.random.default_rng(12345)
B = 4
off = np.tile(np.array([[[0, 1, 0]]], np.float32), (B, 1, 1))
tgt = rng.uniform(-3, 3, size=(B, 1, 3)).astype(np.float32)
tgt[0, 0] = [0, 3, 0]
cons = orc.Constraints([2], off, tgt, np.ones((B, 1), np.float32), [], np.zeros((B, 0, 4)), np.zeros((B, 0, 4)), np.zeros((B, 0)))
for lam in (1e-7, 1e-3):
    opt = GnOptions.make(min_iterations=6, max_iterations=6, threshold=1.0, regularization=lam)
    ref = orc.solve_batch(rig, cons, np.zeros((B, P), np.float32), opt, dtype="f64")
    print("lambda", lam, "oracle f64 history\n", ref["error_history"])
    for route in ("fused", "wide", "explicit_jacobian"):
        pb = capi.Problem(capi.RigHandle(rig, 0), B, [2], [])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
        pb.set_constraints(t(cons.pos_offset), t(cons.pos_target), t(cons.pos_weight), t(np.zeros((B, 0, 4))), t(np.zeros((B, 0, 4))), t(np.zeros((B, 0))))
        pb.set_route(route)
        th = torch.zeros((B, P), device="cuda")
        out = pb.solve(th, opt, want_history=True, want_parameter_history=True)
        torch.cuda.synchronize()
        print(route, "status", out["status"].cpu().numpy(), "iterations", out["iterations"].cpu().numpy())
        print(" history\n", out["error_history"].cpu().numpy())
        print(" theta after it 0 (instance 1):", out["parameter_history"].cpu().numpy()[1, 0])
        print(" oracle theta final (instance 1):", ref["theta"][1])
        lst, jtj, jtr = pb.fused_normal_equations(torch.zeros((B, P), device="cuda")) if route == "fused" else (None, None, None)
        if lst is not None:
            print(" solve list", lst, "diag H (instance 1)", np.diag(jtj[1].cpu().numpy()), "g", jtr[1].cpu().numpy())
