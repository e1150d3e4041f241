#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
F='^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
echo "== ik3"; timeout 120 python scripts/diag_ik3.py 2>&1 | grep -v "$F" | grep -A5 "^fused status\|^wide status\|^explicit_jacobian status" | grep -v "theta\|solve list" | head -40
echo "== weak damping"
timeout 600 python -m pytest tests/test_gpu_weak_damping.py -q --no-header -p no:cacheprovider < /dev/null > gpurun_out/r3_weak_all.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_weak_all.txt | cut -c1-250 | tail -30
echo "== function weights + per-instance + trust region"
timeout 600 python -m pytest tests/test_gpu_function_weights.py tests/test_gpu_per_instance.py tests/test_gpu_trust_region.py tests/test_cpp_shell.py -q --no-header -p no:cacheprovider < /dev/null > gpurun_out/r3_fw.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r3_fw.txt | cut -c1-300 | tail -30
echo "== suite (rest)"
timeout 900 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider --deselect tests/test_gpu_weak_damping.py --deselect tests/test_gpu_function_weights.py < /dev/null > gpurun_out/r3_suite.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_suite.txt | cut -c1-250 | head -20
