#!/bin/bash
# round 5, call B: A/B of the diagnostics' cost, the GPU suite with one line per failure, calibration check
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05b; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05b_ab main nodiag 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=line < /dev/null 2>&1 | tail -70 > $out/pytest_gpu.txt; tail -70 $out/pytest_gpu.txt
timeout 600 python scripts/diag_precision.py estimate < /dev/null 2>&1 | grep -v amdgpu.ids > $out/estimate.txt; grep -c . $out/estimate.txt
