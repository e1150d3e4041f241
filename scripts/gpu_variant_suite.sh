#!/bin/bash
# Everything a build variant has to pass before it becomes the default (run on the GPU box, after
# `MMX_BUILD_VARIANT=$VAR python -m momentum_amd.build` here): the whole -m gpu suite and the randomised sweeps with the
# variant loaded through MMX_LIB (the C++ shell programs link the default library and are unaffected), then the A/B bench.
#   gpurun --timeout 900 -- 'VAR=lookahead bash scripts/gpu_variant_suite.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
VAR=${VAR:-lookahead}
export MMX_LIB=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$VAR.so
[ -f "$MMX_LIB" ] || { echo "missing $MMX_LIB"; exit 1; }
export MMX_TEST_STAGED=1 # the tests of the staged per-rule instantiations too
timeout 400 python -m pytest tests -q -m gpu < /dev/null > gpurun_out/variant_${VAR}_suite.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/variant_${VAR}_suite.txt | cut -c1-200 | tail -8
MMX_FUZZ_SEEDS=200 MMX_FUZZ_JMAX=110 timeout 300 python -m pytest tests/test_gpu_fuzz.py -q -k "not wide" < /dev/null 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200 | tail -4
MMX_FUSED_PLAIN=1 MMX_CHOL_LEAN=1 MMX_TREE_NE_WAVES=8 timeout 400 python -m pytest tests -q -m gpu < /dev/null 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | cut -c1-200 | tail -8
unset MMX_LIB
INV_MORE=1 VAR=$VAR timeout 200 bash scripts/gpu_variant_ab.sh
