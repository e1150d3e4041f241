#!/bin/bash
# round 5, call Y: the default library (solve kernels without machine LICM / loop strength reduction) against a variant built
# from the same sources with the same flags (is the default path itself any different?), then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05y; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05y_h main same 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--line-search 2" bash scripts/gpu_ab.sh r05y_ls main same 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--dtype f64" bash scripts/gpu_ab.sh r05y_f64 main same 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -x < /dev/null 2>&1 | tail -15 > $out/pytest_gpu.txt; tail -15 $out/pytest_gpu.txt
