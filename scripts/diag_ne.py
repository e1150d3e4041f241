"""Tree-moment normal equations against the dense product on a random wide rig: where do they differ?"""
import ctypes as C
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from momentum_amd import capi
from tests.helpers import make_problem
from tests.test_gpu_fuzz import random_rig

seed = int(sys.argv[1])
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
buf, nn = np.zeros(P, np.int32), C.c_int32(0)
capi._check(capi.lib().mmx_debug_fused_normal_equations(pb._h, None, None, None, capi.as_ptr(buf, C.c_int32), C.byref(nn), None))
lst = buf[: nn.value]
en = np.zeros(P, np.uint8); en[lst] = 1
pb.set_enabled(en)
td = torch.from_numpy(th0.copy()).to(pb.device)
os.environ.pop("MMX_TREE_NE", None)
Hd, gd, _ = pb.normal_equations(td)
os.environ["MMX_TREE_NE"] = "force"
Ht, gt, _ = pb.normal_equations(td)
Hd, gd, Ht, gt = Hd.cpu().numpy(), gd.cpu().numpy(), Ht.cpu().numpy(), gt.cpu().numpy()
n = len(lst)
print("J", J, "P", P, "n", n, "U", Kp + 3 * Ko)
for b in range(1):
    dH = np.abs(np.tril(Ht[b]) - np.tril(Hd[b])); dg = np.abs(gt[b] - gd[b])
    print("max |dH|", dH.max(), "scale", np.abs(Hd[b]).max(), "max |dg|", dg.max(), "scale", np.abs(gd[b]).max())
    bad = np.argwhere(dH > 1e-3 * np.abs(Hd[b]).max())
    print("bad H entries", len(bad), bad[:12].tolist())
    badg = np.nonzero(dg > 1e-3 * np.abs(gd[b]).max())[0]
    print("bad g entries", badg[:20].tolist(), "params", lst[badg][:20].tolist())
    if len(bad):
        r, c = bad[0]
        print("example", r, c, Ht[b][r, c], Hd[b][r, c], "params", lst[r], lst[c])
# structure of the offending columns
rows_of = lambda p: [(r // 7, r % 7) for r in range(7 * J) for k in range(rig.pt_outer[r], rig.pt_outer[r + 1]) if rig.pt_inner[k] == p]
for p in set(lst[badg][:4].tolist()) if len(badg) else []:
    print("param", p, "drives (joint, dof):", rows_of(p))
