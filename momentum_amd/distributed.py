"""Multi-GPU plumbing of the batched IK path: instances are fully independent (per-instance
convergence test, momentum/solver/solver.cpp:98; the reference batches with independent tasks,
pymomentum/tensor_ik/tensor_ik.cpp:127-177), so a batch shards into contiguous blocks, one
process per GPU, with NO collective on the data path.  The only exchange is one all-reduce of
the per-batch residual norms per solve (sum of final errors, sum of iterations, number of failed
instances): RCCL over xGMI on the GPUs ("nccl" backend on ROCm), gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch


def env_rank() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str = "nccl"):
    """Initialises the process group when WORLD_SIZE > 1; returns torch.distributed or None."""
    rank, world, _ = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of rank `rank`: ceil(total / world) instances per rank, the
    last ranks may be short or empty (strong scaling of a fixed batch)."""
    per = (int(total) + world - 1) // world
    begin = min(rank * per, total)
    return begin, min(begin + per, total)


def reduce_norms(dist, norms: torch.Tensor) -> torch.Tensor:
    """In-place sum all-reduce of the residual-norm vector (<= 64 bytes: latency bound; called
    once per solve, never per iteration)."""
    if dist is not None:
        if norms.is_cuda and dist.get_backend() == "gloo":  # CPU rendezvous of the tests: stage through the host
            host = norms.cpu()
            dist.all_reduce(host)
            norms.copy_(host)
        else:
            dist.all_reduce(norms)
    return norms


def reduce_min_max(dist, value: float, device) -> Tuple[float, float]:
    """(min, max) over ranks of a host scalar (the per-rank rates the bench reports next to the aggregate)."""
    if dist is None:
        return float(value), float(value)
    t = torch.tensor([value, -value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t[0].item()), float(-t[1].item())


def reduce_max(dist, value: float, device) -> float:
    """Max over ranks of a host scalar (the timed region of the bench)."""
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
