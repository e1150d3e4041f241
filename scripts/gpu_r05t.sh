#!/bin/bash
# round 5, call T: the lazily read kernel arguments as the default of the four-workgroup instantiations: the three lines they
# carry, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05t; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05t_h main 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05t_c3 main 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--line-search 2" bash scripts/gpu_ab.sh r05t_ls main 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--batch 32768" bash scripts/gpu_ab.sh r05t_b32 main 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x < /dev/null 2>&1 | tail -15 > $out/pytest_gpu.txt; tail -15 $out/pytest_gpu.txt
