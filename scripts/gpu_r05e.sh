#!/bin/bash
# round 5, call E: per-workgroup latency of the headline kernel at 128 VGPRs (occ4 variant, LDS unchanged: still three
# workgroups per CU, so the rate IS 3 / latency), the precision policy's table, the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05e; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05e_ab main occ4 2>&1 | grep -v amdgpu.ids | tee $out/ab_occ4.txt
timeout 900 python scripts/diag_auto_table.py 1024 < /dev/null 2>&1 | grep -v amdgpu.ids > $out/auto_table.txt; tail -14 $out/auto_table.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=line < /dev/null 2>&1 | tail -25 > $out/pytest_gpu.txt; tail -25 $out/pytest_gpu.txt
