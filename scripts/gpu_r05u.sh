#!/bin/bash
# round 5, call U: the triangular solves with the chain cut to the newest block (variant ahead), the panel's row updates two
# columns per instruction (variant pkchain): A/B on the headline and on cfg3
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05u2; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05u2_h main ahead pkchain 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05u2_c3 main ahead pkchain 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
