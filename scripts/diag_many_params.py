"""812 solved parameters on the explicit-Jacobian route: first step against numpy's solve of the same normal equations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from momentum_amd import capi
from momentum_amd._abi import GnOptions
from test_gpu_many_parameters import _many_parameter_rig, _upload
from tests.helpers import make_problem
dofs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rig = _many_parameter_rig(dofs)
joints = np.arange(rig.num_joints, dtype=np.int32)
B = 3
cons, th0, _ = make_problem(rig, joints, joints, B, seed=77, perturb=0.1)
rh = capi.RigHandle(rig, 0)
pb = capi.Problem(rh, B, cons.pos_parent, cons.ori_parent)
_upload(torch, pb, cons, B)
t0 = torch.from_numpy(th0.copy()).cuda()
jtj, jtr, err = pb.normal_equations(t0)
H = jtj.double().cpu().numpy(); g = jtr.double().cpu().numpy()
for it in (1, 5):
    opt = GnOptions.make(min_iterations=it, max_iterations=it, threshold=1.0, regularization=0.05)
    out = pb.solve(t0.clone(), opt, want_history=True)
    print("iterations", it, "route", pb.last_route(), "n", pb.n, "status", out["status"].cpu().numpy(), "history", out["error_history"].cpu().numpy()[:, :it], "final", out["error"].cpu().numpy())
    if it == 1:
        th = out["theta"].cpu().numpy().astype(np.float64)
        for b in range(B):
            d = np.linalg.solve(H[b] + 0.05 * np.eye(H.shape[1]), g[b])
            want = th0[b].astype(np.float64) - d
            print(b, "step rel diff", np.linalg.norm(th[b] - want) / np.linalg.norm(d), "nan", np.isnan(th[b]).sum(), "first bad", np.nonzero(np.abs(th[b] - want) > 1e-3 * np.abs(d).max())[0][:10])
