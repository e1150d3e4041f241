"""Which further joint error function separates mmx_solve_f64's first error value from the oracle's (one block at a time)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from momentum_amd import _abi, capi, make_humanoid72, humanoid72_landmark_joints
from momentum_amd._abi import GnOptions
from oracle import oracle as orc
from tests.helpers import make_problem
from tests.test_gpu_joint_blocks import _device_block
from tests.test_oracle_joint_blocks import make_block

rig = make_humanoid72(unit=0.01)
lm = humanoid72_landmark_joints(rig)
B = 3
cons, th0, _ = make_problem(rig, lm, lm, B, seed=31, perturb=0.3, weights="random")
names = {_abi.MMX_JC_PLANE: "plane", _abi.MMX_JC_HALF_PLANE: "half_plane", _abi.MMX_JC_AIM_DIST: "aim_dist", _abi.MMX_JC_AIM_DIR: "aim_dir",
         _abi.MMX_JC_FIXED_AXIS_DIFF: "axis_diff", _abi.MMX_JC_FIXED_AXIS_COS: "axis_cos", _abi.MMX_JC_FIXED_AXIS_ANGLE: "axis_angle", _abi.MMX_JC_NORMAL: "normal"}
opt = GnOptions.make(min_iterations=2, max_iterations=2, threshold=1.0, regularization=0.05)
for ty, nm in names.items():
    for variant in ("plain", "fw1.7", "loss"):
        rng = np.random.default_rng(5)
        kw = {}
        if variant == "fw1.7":
            kw["function_weight"] = 1.7
        if variant == "loss":
            kw["loss"] = (1.0, 0.3)
        blk = make_block(ty, rng.choice(np.arange(1, rig.num_joints), size=3, replace=False), rng, weight=1.3, batch=B, **kw)
        full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset, cons.ori_target,
                               cons.ori_weight, joint_blocks=[blk])
        pb = capi.Problem(capi.RigHandle(rig, 0), B, cons.pos_parent, cons.ori_parent)
        t = lambda a, shp: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(shp)).to(pb.device)
        pb.set_constraints(t(cons.pos_offset, (B, cons.Kp, 3)), t(cons.pos_target, (B, cons.Kp, 3)), t(cons.pos_weight, (B, cons.Kp)),
                           t(cons.ori_offset, (B, cons.Ko, 4)), t(cons.ori_target, (B, cons.Ko, 4)), t(cons.ori_weight, (B, cons.Ko)),
                           joint_blocks=[_device_block(torch, blk, pb.device)])
        out = pb.solve_f64(torch.from_numpy(th0.astype(np.float64)).to(pb.device), opt, want_history=True)
        ref = orc.solve_batch(rig, full, th0, opt, dtype="f64")
        h, href = out["error_history"].cpu().numpy(), ref["error_history"]
        rel = np.linalg.norm(out["theta"].cpu().numpy() - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
        print(f"{nm:11s} {variant:6s} |h0-href0|/href0 {np.abs(h[:,0]-href[:,0]).max()/np.abs(href[:,0]).max():.2e}  theta rel {rel.max():.2e}")
