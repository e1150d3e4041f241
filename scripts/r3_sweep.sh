cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
MMX_FUZZ_SEEDS=150 timeout 400 python -m pytest tests/test_gpu_fuzz.py -k "not wide" -q --no-header -p no:cacheprovider < /dev/null 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200 | tail -5
MMX_TEST_ROUTE=prefer_wide MMX_FUZZ_SEEDS=150 MMX_FUZZ_JMAX=110 timeout 400 python -m pytest tests/test_gpu_fuzz.py -k "not wide" -q --no-header -p no:cacheprovider < /dev/null 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200 | tail -5
MMX_FUZZ_WIDE_SEEDS=64 MMX_FUZZ_WIDE_JMAX=195 timeout 400 python -m pytest tests/test_gpu_fuzz.py -k "wide" -q --no-header -p no:cacheprovider < /dev/null 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200 | tail -5
