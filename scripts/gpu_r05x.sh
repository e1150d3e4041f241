#!/bin/bash
# round 5, call X: loop strength reduction off as well (variant nolsr: no spilled vector register left in the four-workgroup
# instantiations), then the full forms of the tile products / operand buffers (nolsrfat) and of the solves (nolsrfatall) back
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05x; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05x_h main nolsr nolsrfat nolsrfatall 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05x_c3 main nolsr nolsrfat 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--line-search 2" bash scripts/gpu_ab.sh r05x_ls main nolsr nolsrfat 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--config cfg5" bash scripts/gpu_ab.sh r05x_c5 main nolsr 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--dtype f64" bash scripts/gpu_ab.sh r05x_f64 main nolsr 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
