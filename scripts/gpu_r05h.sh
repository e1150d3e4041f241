#!/bin/bash
# round 5, call H: pipelined walk of the transform's CSR (A/B), parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05h; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05h_ab main nopipe 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--batch 32768" bash scripts/gpu_ab.sh r05h_ab32k main nopipe 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_trust_region.py -m gpu -q --tb=line < /dev/null 2>&1 | tail -8 > $out/pytest_sel.txt; tail -8 $out/pytest_sel.txt
