"""FK rounding noise of the HIP path against the oracle's float instantiation, both measured against the oracle's double
FK: world positions (absolute, metres) and world rotations (quaternion difference) over random poses of the 72-joint rig."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from momentum_amd import capi, make_humanoid72
from oracle import oracle as orc

for variant in ("p128", "p219"):
    rig = make_humanoid72(seed=12345, variant=variant, unit=0.01)
    B = 256
    rng = np.random.default_rng(1)
    th = rng.uniform(-0.3, 0.3, size=(B, rig.num_params)).astype(np.float32)
    pb = capi.Problem(capi.RigHandle(rig, 0), B, [0], [])
    st = pb.skeleton_state(torch.from_numpy(th).cuda()).cpu().numpy().astype(np.float64)
    ep, eq, ep32, eq32 = [], [], [], []
    for b in range(B):
        r64 = orc.skeleton_state(rig, th[b].astype(np.float64), "f64")["world"]
        r32 = orc.skeleton_state(rig, th[b], "f32")["world"].astype(np.float64)
        def qd(a, c):
            s = np.sign((a * c).sum(-1, keepdims=True))
            return np.abs(a - s * c).max(-1)
        ep.append(np.abs(st[b][:, :3] - r64[:, :3]).max(-1)); eq.append(qd(st[b][:, 3:7], r64[:, 3:7]))
        ep32.append(np.abs(r32[:, :3] - r64[:, :3]).max(-1)); eq32.append(qd(r32[:, 3:7], r64[:, 3:7]))
    ep, eq, ep32, eq32 = map(np.array, (ep, eq, ep32, eq32))
    print(f"{variant}: position error  hip: median {np.median(ep):.2e} p99 {np.quantile(ep, .99):.2e} max {ep.max():.2e} | float oracle: median {np.median(ep32):.2e} p99 {np.quantile(ep32, .99):.2e} max {ep32.max():.2e}")
    print(f"{variant}: rotation error  hip: median {np.median(eq):.2e} p99 {np.quantile(eq, .99):.2e} max {eq.max():.2e} | float oracle: median {np.median(eq32):.2e} p99 {np.quantile(eq32, .99):.2e} max {eq32.max():.2e}")
    # by depth
    depth = np.zeros(rig.num_joints, int)
    for j in range(rig.num_joints):
        p = rig.parent[j]
        depth[j] = 0 if p < 0 else depth[p] + 1
    for d in range(0, depth.max() + 1, 3):
        m = depth == d
        print(f"   depth {d:2d}: pos hip {np.median(ep[:, m]):.2e} f32 {np.median(ep32[:, m]):.2e}  rot hip {np.median(eq[:, m]):.2e} f32 {np.median(eq32[:, m]):.2e}")
