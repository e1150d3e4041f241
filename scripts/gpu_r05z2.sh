#!/bin/bash
# round 5, call Z2: the lazily loaded descriptors' pointers typed global (variant globalptr), dot4 as two packed multiply-adds
# (variant pkdot); the test of the two argument forms of the solve
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05z2; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05z2_h main globalptr pkdot 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05z2_c3 main globalptr pkdot 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--line-search 2" bash scripts/gpu_ab.sh r05z2_ls main globalptr 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
timeout 600 python -m pytest tests/test_gpu_function_weights.py -m gpu -q --tb=short < /dev/null 2>&1 | tail -12 | tee $out/pytest_fw.txt
