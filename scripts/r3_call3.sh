#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== cfg5 diag (main)"; timeout 300 python scripts/diag_cfg5.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8
echo "== cfg5 diag (nofloor)"; MMX_LIB=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_nofloor.so timeout 300 python scripts/diag_cfg5.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8
echo "== A/B floor"; BENCH_ARGS="" timeout 600 bash scripts/gpu_ab.sh abfloor main nofloor
echo "== suite"
timeout 900 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --deselect tests/test_gpu_weak_damping.py < /dev/null > gpurun_out/r3_suite.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_suite.txt | cut -c1-250 | head -20
