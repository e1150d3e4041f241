"""bench.py with two ranks (torch.distributed.run, one node), exactly as the driver launches it for
N > 1, but over gloo and with both ranks on the one GPU of the test box: checks the sharding, the
residual-norm all-reduce, the max-over-ranks timing and that rank 0 prints ONE JSON line whose
`value` is the whole-job aggregate."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_over_gloo():
    port = str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "1024", "--backend", "gloo", "--no-cpu-baseline"]  # fmt: skip
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2048 and d["config"]["batch_per_gpu"] == 1024
    # both shards were solved and reduced: 2 x 1024 instances x 10 iterations, nobody failed
    assert d["check"]["sum_iterations"] == 2 * 1024 * 10 and d["check"]["failed_instances"] == 0
    assert d["value"] > 0 and abs(d["value"] - 2048 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-4 * d["value"]  # (the line carries six significant digits)
    ex = d["config"]["exchange"]
    assert ex["per_rank_solves_per_s"][0] <= ex["per_rank_solves_per_s"][1] and ex["per_rank_solves_per_s"][0] * 2 >= d["value"] * 0.999
    # the N = 1 line of the same command carries the same workload and per-GPU batch: what a scaling run compares
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "1024", "--no-cpu-baseline",
                          "--no-extra-configs", "--check-instances", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)  # fmt: skip
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["config"]["workload"] == d["config"]["workload"] and d1["config"]["batch_per_gpu"] == d["config"]["batch_per_gpu"]
    assert d1["metric"] == d["metric"] and d1["unit"] == d["unit"] and d1["config"]["global_batch"] * 2 == d["config"]["global_batch"]


def test_bench_config4_is_the_weak_scaling_shard():
    """--config cfg4 = BASELINE configs[3]: 32768 instances per GPU of cfg2's problem (262144 over eight GPUs)."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.CONFIGS["cfg4"][2] == 32768 and bench.CONFIGS["cfg4"][:2] == bench.CONFIGS["cfg2"][:2] and bench.CONFIGS["cfg4"][3] == bench.CONFIGS["cfg2"][3]
