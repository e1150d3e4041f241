#!/bin/bash
# round 5, call W: the library compiled without the machine-level loop-invariant code motion (variant nomlicm: hoisted per-lane
# address arithmetic is what the one-launch solve spills at 128 registers): every route
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05w; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05w_h main nomlicm 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05w_c3 main nomlicm 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--line-search 2" bash scripts/gpu_ab.sh r05w_ls main nomlicm 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--config cfg5" bash scripts/gpu_ab.sh r05w_c5 main nomlicm 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--dtype f64" bash scripts/gpu_ab.sh r05w_f64 main nomlicm 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
