"""Diagnostic: per error-function type, the J / r error of the HIP path vs the f64 oracle and the
relative theta error of a 10-iteration GN solve (and of the oracle's own f32 instantiation)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momentum_amd import _abi, capi, humanoid72_landmark_joints, make_humanoid72
from momentum_amd._abi import GnOptions
from oracle import oracle as orc
from tests.test_oracle_joint_blocks import TYPES, make_block
from tests.test_gpu_joint_blocks import _problem

rig = make_humanoid72(unit=0.01)
lm = humanoid72_landmark_joints(rig)
B = 4
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
for name, ty in TYPES.items():
    for loss in ((2.0, 1.0), (0.0, 0.5)):
        rng = np.random.default_rng(2468)
        blk = make_block(ty, rng.choice(rig.num_joints, size=3), rng, weight=1.0, batch=B, loss=loss)
        rh, pb, full, th0 = _problem(torch, orc, rig, lm, lm, B, 2468, [blk])
        theta = rng.uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)
        jac, res, err = pb.eval_jacobian(torch.from_numpy(theta).cuda())
        jac, res = jac.cpu().numpy(), res.cpu().numpy()
        ej = er = 0.0
        for b in range(B):
            J, r, e = orc.eval_jacobian(rig, full.instance(b), theta[b].astype(np.float64))
            rows = slice(full.rows - blk.rows, full.rows)
            ej = max(ej, np.abs(jac[b].T[rows] - J[rows]).max() / max(1.0, np.abs(J[rows]).max()))
            er = max(er, np.abs(res[b][rows] - r[rows]).max() / max(1.0, np.abs(r[rows]).max()))
        out = pb.solve(torch.from_numpy(th0.copy()).cuda(), opt)
        ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
        r32 = orc.solve_batch(rig, full, th0, opt, dtype="f32")
        n = np.linalg.norm(ref["theta"], axis=1)
        rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"], axis=1) / n
        rel32 = np.linalg.norm(r32["theta"] - ref["theta"], axis=1) / n
        print(f"{name:18s} loss {loss}: J {ej:.1e} r {er:.1e} | gpu rel {rel.max():.1e} oracle-f32 rel {rel32.max():.1e} | final err {ref['error'].max():.3g}")
