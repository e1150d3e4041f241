"""The N > 1 path on CPU: two gloo ranks shard a batch, reduce residual norms and agree on the
max-over-ranks timing -- the same helpers bench.py drives over RCCL on the GPUs."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["MMX_ROOT"])
import torch
from momentum_amd import distributed as D
rank, world, local = D.env_rank()
dist = D.init("gloo")
assert dist is not None and dist.get_world_size() == world == 2
total = 1001
b, e = D.shard_range(total, rank, world)
# every instance is owned by exactly one rank
owned = torch.zeros(total, dtype=torch.int64)
owned[b:e] = 1
dist.all_reduce(owned)
assert bool((owned == 1).all()), "shards must partition the batch"
# residual norms: sum over ranks of (sum error, sum iterations, failed)
norms = torch.tensor([float(e - b) * 0.5, float(e - b) * 10, float(rank)], dtype=torch.float64)
D.reduce_norms(dist, norms)
assert abs(norms[0].item() - total * 0.5) < 1e-9 and norms[1].item() == total * 10 and norms[2].item() == 1.0
t = D.reduce_max(dist, 1.0 + rank, torch.device("cpu"))
assert t == 2.0
dist.barrier()
dist.destroy_process_group()
sys.stdout.write(f"rank {rank} ok\n"); sys.stdout.flush()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_shard_and_reduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MMX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script)]  # fmt: skip
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_shard_range_edges():
    from momentum_amd.distributed import shard_range

    assert shard_range(10, 0, 1) == (0, 10)
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]  # ragged / empty shards
    assert shard_range(0, 0, 2) == (0, 0)
