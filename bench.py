#!/usr/bin/env python3
"""bench.py -- character IK solves/sec on MI355X (BASELINE.json metric) + J-assembly HBM roofline.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--config cfg2|cfg2_all|cfg3]

One "step" = one pass of the hot path over one batch: a full batched solve (10 Gauss-Newton
iterations: FK -> Jacobian/residual -> JtJ/Jtr -> Cholesky (+1 refinement) -> theta update) of
B independent 72-joint humanoid instances per GPU, inputs resident in HBM.  For N > 1 the driver
launches one rank per GPU (torch.distributed.run); instances are sharded (weak scaling: B per GPU
fixed) and the only collective is one all-reduce of the per-batch residual norms per solve.

Prints ONE JSON line on rank 0 (see the task contract): metric/value/unit/... plus
  "roofline":     achieved algorithmic HBM GB/s of the J-assembly kernel (mmx_eval_jacobian),
                  timed live with HIP events on the launch stream
  "cpu_baseline": the CPU oracle (restatement of momentum's algorithm, kind "port") timed on
                  this box's host cores on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X datasheet (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
UNIT = 0.01  # rig offsets "U[2,30] cm" expressed in metres

CONFIGS = {
    # name: (rig variant, constraint joints, default batch per GPU, step rule, description)
    "cfg2": ("p128", "landmarks", 4096, 0, "BASELINE configs[1]: B x 72-joint humanoid (P=128), position+orientation on 16 landmark joints (M=192), GN lambda=0.05, 10 iterations"),
    "cfg3": ("p128", "landmarks", 65536, 1, "BASELINE configs[2]: 65536 x 72-joint humanoid (P=128, M=192), LM gain-ratio damping schedule (lambda0=0.05), 10 iterations"),
    "cfg5": ("rig300", "cfg5", 8192, 0, "BASELINE configs[4]: 8192 x 300-joint hand+body rig (P=300), 150 position + 50 orientation constraints (M=900), GN lambda=0.05, 10 iterations"),
    "cfg2_all": ("p219", "all", 4096, 0, "BASELINE configs[1] stress variant: P=219, position+orientation on all 72 joints (M=864)"),
}


def algorithmic_bytes_per_instance(M: int, P: int, Kp: int, Ko: int) -> int:
    """write J (M*P) and r (M), read theta (P), constraint payload and
    parent indices; fp32; rig constants are batch-shared and excluded."""
    return 4 * (M * P + M) + 4 * P + 4 * (7 * Kp + 9 * Ko) + 4 * (Kp + Ko)


def make_device_problem(rig, parents, B, device_index, seed):
    """Synthetic batch generated ON the GPU: theta* = U[-0.3,0.3]^P, targets = FK(theta*) through the
    product's own FK kernel, theta0 = 0."""
    from momentum_amd import capi

    pos_parents, ori_parents = parents if isinstance(parents, tuple) else (parents, parents)
    rh = capi.RigHandle(rig, device_index)
    pb = capi.Problem(rh, B, pos_parents, ori_parents)
    dev = pb.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    P, Kp, Ko = rig.num_params, len(pos_parents), len(ori_parents)
    theta_star = (torch.rand((B, P), generator=g, device=dev, dtype=torch.float32) * 2 - 1) * 0.3
    st = pb.skeleton_state(theta_star)  # [B,J,8]
    pidx = torch.as_tensor(np.asarray(pos_parents, dtype=np.int64), device=dev)
    oidx = torch.as_tensor(np.asarray(ori_parents, dtype=np.int64), device=dev)
    pos_offset = torch.zeros((B, Kp, 3), device=dev)
    pos_target = st[:, pidx, 0:3].contiguous()
    ori_offset = torch.zeros((B, Ko, 4), device=dev)
    ori_offset[..., 3] = 1.0
    ori_target = st[:, oidx, 3:7].contiguous()
    pb.set_constraints(pos_offset, pos_target, torch.ones((B, Kp), device=dev), ori_offset, ori_target, torch.ones((B, Ko), device=dev), 1.0, 1.0)
    theta0 = torch.zeros((B, P), device=dev, dtype=torch.float32)
    return rh, pb, theta0, theta_star


def usable_cores() -> int:
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a container can see 256 logical CPUs and be allowed a dozen)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(rig, parents, sample, seed, options):
    """The CPU oracle timed on the host cores (bounded sample of the same workload)."""
    from oracle import oracle as orc
    from tests.helpers import make_problem

    cores = usable_cores()
    pos_parents, ori_parents = parents if isinstance(parents, tuple) else (parents, parents)
    cons, th0, _ = make_problem(rig, pos_parents, ori_parents, sample, seed=seed, perturb=0.3)
    orc.solve_batch(rig, cons, th0[: min(sample, 2 * cores)], options, dtype="f32", nthreads=cores)  # warm
    t0 = time.perf_counter()
    orc.solve_batch(rig, cons, th0, options, dtype="f32", nthreads=cores)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    n1 = max(1, min(sample, 64))
    one = orc.Constraints(
        cons.pos_parent, cons.pos_offset[:n1], cons.pos_target[:n1], cons.pos_weight[:n1],
        cons.ori_parent, cons.ori_offset[:n1], cons.ori_target[:n1], cons.ori_weight[:n1],
    )  # fmt: skip
    orc.solve_batch(rig, one, th0[:n1], options, dtype="f32", nthreads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": sample / dt,
        "unit": "solves/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{sample} instances of the same workload, fp32, one solver per task over {cores} std::threads (= usable cores: affinity mask and cgroup quota; os.cpu_count() = {os.cpu_count()}; mirrors tensor_ik.cpp:127); single-thread: {n1 / dt1:.1f} solves/s",
        "single_thread_value": n1 / dt1,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the config's)")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8192)
    ap.add_argument("--jac-launches", type=int, default=20)
    ap.add_argument("--line-search", type=int, default=0, choices=[0, 1, 2], help="MMX_LINE_SEARCH_*: 0 none (the BASELINE metric), 1 GaussNewtonSolverT's rule, 2 the rule of the batched driver's solvers (SubsetGN / GN-QR)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL; gloo only for the plumbing test)")
    args = ap.parse_args()

    from momentum_amd import distributed as D

    rank, world, local_rank = D.env_rank()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    # one rank per GPU; --backend gloo with fewer GPUs than ranks is only for the plumbing test
    # (tests/test_bench_two_ranks.py), where the ranks share a device
    local_rank = local_rank % torch.cuda.device_count() if args.backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    dist = D.init(args.backend)  # RCCL behind the "nccl" backend on ROCm; None when world == 1

    from momentum_amd import humanoid72_landmark_joints, make_humanoid72
    from momentum_amd._abi import GnOptions

    variant, which, defB, step_rule, desc = CONFIGS[args.config]
    B = args.batch if args.batch > 0 else defB
    if variant == "rig300":
        from momentum_amd import make_rig300

        rig = make_rig300(seed=12345, unit=UNIT)
    else:
        rig = make_humanoid72(seed=12345, variant=variant, unit=UNIT)
    if which == "landmarks":
        pos_parents = ori_parents = humanoid72_landmark_joints(rig)
    elif which == "cfg5":
        prng = np.random.default_rng(77)
        pos_parents = prng.choice(rig.num_joints, size=150, replace=False).astype(np.int32)
        ori_parents = prng.choice(rig.num_joints, size=50, replace=False).astype(np.int32)
    else:
        pos_parents = ori_parents = np.arange(rig.num_joints, dtype=np.int32)
    parents = (pos_parents, ori_parents)
    seed = 12345 + 1000003 * rank  # every rank solves different instances (its shard of the batch)
    rh, pb, theta0, theta_star = make_device_problem(rig, parents, B, local_rank, seed)
    opt = GnOptions.make(min_iterations=args.iterations, max_iterations=args.iterations, threshold=1.0, regularization=0.05, step_rule=step_rule, do_line_search=args.line_search)
    dev = pb.device
    theta = theta0.clone()
    outputs = dict(
        error=torch.empty((B,), dtype=torch.float64, device=dev),
        iterations=torch.empty((B,), dtype=torch.int32, device=dev),
        status=torch.empty((B,), dtype=torch.int32, device=dev),
    )
    norms = torch.zeros(3, dtype=torch.float64, device=dev)

    def step():
        theta.copy_(theta0)
        pb.solve(theta, opt, outputs=outputs)
        # the path's only exchange: per-batch residual norms (sum error, sum iterations, #failed)
        norms[0] = outputs["error"].sum()
        norms[1] = outputs["iterations"].sum()
        norms[2] = (outputs["status"] != 0).sum()
        D.reduce_norms(dist, norms)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = D.reduce_max(dist, elapsed, dev)
    total_err, total_it, failed = [float(x) for x in norms.tolist()]

    # ---- roofline of the J-assembly kernel (mmx_eval_jacobian): HIP events on the launch stream
    M, P = pb.M, pb.P
    Kp_, Ko_ = len(parents[0]), len(parents[1])
    jac = torch.empty((B, P, M), dtype=torch.float32, device=dev)
    res = torch.empty((B, M), dtype=torch.float32, device=dev)
    err = torch.empty((B,), dtype=torch.float64, device=dev)
    for _ in range(3):
        pb.eval_jacobian(theta_star, jac, res, err)
    torch.cuda.synchronize()
    # HIP events attached to the kernel's own dispatch packet on the launch stream
    # (mmx_eval_jacobian_timed -> hipExtLaunchKernelGGL): the kernel's duration as a rocprofv3 kernel
    # trace reports it, without launch latency or the gap between back-to-back dispatches
    jac_ms = float(np.mean([pb.eval_jacobian_kernel_ms(theta_star, jac, res, err) for _ in range(args.jac_launches)]))
    # the same launches bracketed by ordinary recorded events (includes the dispatch latency): reported as context
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.jac_launches)]
    for a, b in evs:
        a.record()
        pb.eval_jacobian(theta_star, jac, res, err)
        b.record()
    torch.cuda.synchronize()
    jac_ms_recorded = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    bytes_per_launch = B * algorithmic_bytes_per_instance(M, P, Kp_, Ko_)
    achieved = bytes_per_launch / (jac_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_jacobian.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            if pm.get("batch") == B and pm.get("config") == args.config:
                traffic = pm.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    del jac
    # context for the roofline: (i) the same kernel at the weak-scaling shard size of BASELINE
    # configs[3] (32768 instances per GPU), where launch ramp-up and the FK prologue are amortised,
    # (ii) what a plain write-only fill of the same byte count reaches on this box
    extra = {}
    if rank == 0 and world == 1:
        def timed(fn, n):
            for _ in range(2):
                fn()
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, b in pairs:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            return float(np.mean([a.elapsed_time(b) for a, b in pairs]))

        fill = torch.empty(bytes_per_launch // 4, dtype=torch.float32, device=dev)
        fill_ms = timed(lambda: fill.zero_(), 5)
        extra["fill_same_bytes_gbs"] = bytes_per_launch / (fill_ms * 1e-3) / 1e9
        # the same stores (layout, workgroup shape, width) without any kinematics: what the write
        # pattern of a column-major J per instance allows on this box
        for _ in range(2):
            pb.store_pattern_kernel_ms(fill)
        sp_ms = float(np.mean([pb.store_pattern_kernel_ms(fill) for _ in range(5)]))
        extra["store_pattern_gbs"] = B * 4 * M * P / (sp_ms * 1e-3) / 1e9
        del fill
        BL = 32768
        if args.config == "cfg2" and B < BL:
            rhL, pbL, _, thetaL = make_device_problem(rig, parents, BL, local_rank, seed + 1)
            jacL = torch.empty((BL, P, M), dtype=torch.float32, device=dev)
            resL = torch.empty((BL, M), dtype=torch.float32, device=dev)
            errL = torch.empty((BL,), dtype=torch.float64, device=dev)
            for _ in range(2):
                pbL.eval_jacobian(thetaL, jacL, resL, errL)
            msL = float(np.mean([pbL.eval_jacobian_kernel_ms(thetaL, jacL, resL, errL) for _ in range(5)]))
            gbsL = BL * algorithmic_bytes_per_instance(M, P, Kp_, Ko_) / (msL * 1e-3) / 1e9
            extra["at_batch_32768"] = {"achieved": gbsL, "frac": gbsL / HBM_PEAK_GBS, "ms_per_launch": msL}
            del jacL, resL, errL, pbL, rhL

    if rank == 0:
        solves = float(B) * world * args.steps
        line = {
            "metric": "character IK solves/sec (72-joint, 10 GN iters)",
            "value": solves / elapsed,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": desc,
                "batch_per_gpu": B,
                "global_batch": B * world,
                "joints": rig.num_joints,
                "params": P,
                "rows": M,
                "gn_iterations": args.iterations,
                "line_search": args.line_search,
                "regularization": 0.05,
                "sharding": f"{world} x {B} independent instances, one all-reduce of residual norms per solve",
            },
            "check": {"sum_final_error": total_err, "sum_iterations": total_it, "failed_instances": failed},
            "roofline": {
                "kernel": "fkJacobianKernel<true> (mmx_eval_jacobian: FK + dense J/r assembly)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "bytes_per_launch": bytes_per_launch,
                "ms_per_launch": jac_ms,
                "timing": "HIP events attached to the kernel's dispatch packet on the launch stream (hipExtLaunchKernelGGL)",
                "ms_per_launch_recorded_events": jac_ms_recorded,
                "batch": B,
                **extra,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(rig, parents, args.cpu_sample, 12345, opt)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
