#!/bin/bash
# Round 3, GPU call 1: the weak-damping parity tests on the default library, then the staged variants of round 2
# (correctness of their own tests, then the A/B bench).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== weak damping"
timeout 600 python -m pytest tests/test_gpu_weak_damping.py -q -x --no-header -p no:cacheprovider < /dev/null > gpurun_out/r3_weak.txt 2>&1
tail -30 gpurun_out/r3_weak.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_weak_damping.py -q --no-header -p no:cacheprovider < /dev/null > gpurun_out/r3_weak_all.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_weak_all.txt | cut -c1-250 | tail -40
echo "== staged tests (MMX_TEST_STAGED=1)"
MMX_TEST_STAGED=1 timeout 300 python -m pytest tests/test_gpu_parity.py -q -k "plain_gauss_newton or wide_solve_variants" --no-header < /dev/null 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | cut -c1-250 | tail -10
echo "== staged env switches over the parity files"
MMX_FUSED_PLAIN=1 MMX_CHOL_LEAN=1 MMX_TREE_NE_WAVES=8 timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_parameter_rows.py tests/test_gpu_trust_region.py -q --no-header < /dev/null 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | cut -c1-250 | tail -10
echo "== A/B"
SKIP_CLOCKS=1 VAR=lookahead timeout 900 bash scripts/gpu_variant_ab.sh
