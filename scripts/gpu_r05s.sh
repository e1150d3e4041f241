#!/bin/bash
# round 5, call S: the resident factor kernel at eight waves per workgroup (variant fact8) on cfg5; kernel arguments behind one
# pointer (variant argptr) now that the one-launch solve runs at 128 registers; phase clocks of both routes
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05s; mkdir -p $out
export TMPDIR=/tmp
BENCH_ARGS="--config cfg5" bash scripts/gpu_ab.sh r05s_c5 main fact8 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
bash scripts/gpu_ab.sh r05s_h main argptr 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05s_c3 main argptr 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--line-search 2" bash scripts/gpu_ab.sh r05s_ls main argptr 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null > /dev/null 2> $out/clocks_fused.txt; tail -32 $out/clocks_fused.txt
timeout 300 python scripts/diag_wide_clocks.py cfg5 8192 > $out/clocks_wide_main.txt 2>&1; tail -40 $out/clocks_wide_main.txt
MMX_LIB=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_fact8.so timeout 300 python scripts/diag_wide_clocks.py cfg5 8192 > $out/clocks_wide_fact8.txt 2>&1; tail -40 $out/clocks_wide_fact8.txt
