#!/bin/bash
# round 5, call R: the generic-rule instantiations (line searches) at four workgroups per CU (variant gen4): A/B on the batched
# driver's line-search line and the LM schedule's (cfg3), then the full GPU suite on the default library
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05r; mkdir -p $out
export TMPDIR=/tmp
BENCH_ARGS="--line-search 2" bash scripts/gpu_ab.sh r05r_ls main gen4 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05r_c3 main gen4 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
bash scripts/gpu_ab.sh r05r_h main 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=line -x < /dev/null 2>&1 | tail -8 > $out/pytest_gpu.txt; tail -8 $out/pytest_gpu.txt
