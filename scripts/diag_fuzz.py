"""Diagnosis of one seed of tests/test_gpu_fuzz.py::test_random_rig_with_joint_blocks_and_ellipsoids: the fused general
rows against the explicit-Jacobian kernels, the oracle in double and the oracle in float."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momentum_amd import capi  # noqa: E402
from momentum_amd._abi import EllipsoidLimit, GnOptions  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import make_problem  # noqa: E402
from tests.test_gpu_fuzz import random_rig  # noqa: E402
from tests.test_oracle_joint_blocks import TYPES, make_block  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
J = int(rng.integers(2, 40))
rig = random_rig(rng, J, ["chain", "star", "bushy"][seed % 3])
P = rig.num_params
Kp = int(rng.integers(0, 5))
pp = rng.integers(0, J, size=Kp).astype(np.int32)
op = np.zeros(0, np.int32)
B = 3
cons, th0, _ = make_problem(rig, pp, op, B, seed=seed, perturb=0.25, random_offsets=True, weights="random")
types = list(TYPES.values())
blocks = [make_block(types[int(k)], rng.integers(0, J, size=int(rng.integers(1, 4))), rng, weight=1.0, batch=B,
                     function_weight=float(rng.uniform(0.3, 1.2)), loss=(2.0, 1.0) if rng.uniform() < 0.7 else (0.0, 0.8))
          for k in rng.choice(len(types), size=int(rng.integers(1, 4)), replace=False)]  # fmt: skip
ells = [EllipsoidLimit.make(int(rng.integers(0, J)), rng.uniform(-0.2, 0.2, 3), int(rng.integers(0, J)), rng.uniform(-0.2, 0.2, 3),
                            rng.uniform(-180, 180, 3), rng.uniform(0.1, 0.6, 3), float(rng.uniform(0.5, 2.0)))
        for _ in range(int(rng.integers(0, 3)))]  # fmt: skip
wl = 50.0
full = orc.Constraints(cons.pos_parent, cons.pos_offset, cons.pos_target, cons.pos_weight, cons.ori_parent, cons.ori_offset,
                       cons.ori_target, cons.ori_weight, joint_blocks=blocks, ellipsoid_limits=ells, limit_function_weight=wl)  # fmt: skip
pb = capi.Problem(capi.RigHandle(rig, 0), B, pp, op)
f = lambda a, shp: np.ascontiguousarray(a, np.float32).reshape(shp)
pb.set_constraints(f(cons.pos_offset, (B, Kp, 3)), f(cons.pos_target, (B, Kp, 3)), f(cons.pos_weight, (B, Kp)), f(cons.ori_offset, (B, 0, 4)),
                   f(cons.ori_target, (B, 0, 4)), f(cons.ori_weight, (B, 0)), joint_blocks=blocks, ellipsoid_limits=ells,
                   limit_function_weight=wl)  # fmt: skip
en = (rng.uniform(size=P) < 0.85).astype(np.uint8)
en[:3] = 1
pb.set_enabled(en)
print("J", J, "P", P, "Kp", Kp, "blocks", [(k.type, k.count, k.loss) for k in blocks], "ells", len(ells), "M", pb.M)
for ls in (0, 1):
    opt = GnOptions.make(min_iterations=4, max_iterations=4, regularization=0.5, do_line_search=ls)
    ref = orc.solve_batch(rig, full, th0, opt, enabled=en, dtype="f64")
    r32 = orc.solve_batch(rig, full, th0, opt, enabled=en, dtype="f32")
    den = np.maximum(np.linalg.norm(ref["theta"], axis=1), 1e-3)
    print("line_search", ls, "oracle f32 vs f64:", np.linalg.norm(r32["theta"] - ref["theta"], axis=1) / den)
    for gen in ("1", "0"):
        os.environ["MMX_FUSED_GENERAL"] = gen
        out = pb.solve(torch.from_numpy(th0.copy()).to(pb.device), opt, want_history=True)
        th = out["theta"].cpu().numpy()
        h = out["error_history"].cpu().numpy()
        print("  fused_general", gen, "rel", np.linalg.norm(th - ref["theta"], axis=1) / den, "hist rel diff", np.abs(h - ref["error_history"]).max(axis=1) / np.abs(ref["error_history"]).max(axis=1))
    print("  ref history[0]", ref["error_history"][0])
