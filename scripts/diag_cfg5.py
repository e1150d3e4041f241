"""cfg5 parity at the bench's batch: which instances leave the bound, do they when solved alone, statuses."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from momentum_amd._abi import GnOptions
from oracle import oracle as orc

rig, parents, _, _, _ = bench.build_rig("cfg5")
opt = GnOptions.make(min_iterations=10, max_iterations=10, threshold=1.0, regularization=0.05)
for B in (8192, 1024):
    db = bench.DeviceBatch(rig, parents, B, 0, 12345)
    out = db.pb.solve(db.theta0.clone(), opt)
    torch.cuda.synchronize()
    n = 1024
    cons = db.host_constraints(n)
    ref = orc.solve_batch(rig, cons, np.zeros((n, rig.num_params), np.float32), opt, dtype="f64", nthreads=bench.usable_cores())
    th = out["theta"][:n].cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(th - ref["theta"], axis=1) / np.linalg.norm(ref["theta"], axis=1)
    st = out["status"].cpu().numpy()
    bad = np.nonzero(rel > 1e-5)[0]
    print(f"B={B} route {db.pb.last_route()}: max rel {rel.max():.3e} above {len(bad)} idx {bad[:10]} status!=0 {(st != 0).sum()} status of bad {st[bad][:10]} err gpu {out['error'][:n].cpu().numpy()[bad][:5]} err ref {ref['error'][bad][:5]}")
    if len(bad):
        idx = torch.as_tensor(bad[:8], device=db.pb.device)
        from momentum_amd import capi
        pb2 = capi.Problem(db.rh, len(idx), parents[0], parents[1])
        c = lambda t: t[idx].contiguous()
        pb2.set_constraints(c(db.pos_offset), c(db.pos_target), c(db.pos_weight), c(db.ori_offset), c(db.ori_target), c(db.ori_weight))
        o2 = pb2.solve(torch.zeros((len(idx), rig.num_params), device=db.pb.device), opt)
        th2 = o2["theta"].cpu().numpy().astype(np.float64)
        rel2 = np.linalg.norm(th2 - ref["theta"][bad[:8]], axis=1) / np.linalg.norm(ref["theta"][bad[:8]], axis=1)
        print("   the same instances solved alone:", rel2, "status", o2["status"].cpu().numpy())
    del db
