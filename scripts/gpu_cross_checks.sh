#!/bin/bash
# One-off cross-checks beyond the suite (run on the GPU box): large seed sweeps of the randomised tests, also with every
# solve sent through the wide route where it applies (MMX_TEST_ROUTE=prefer_wide, tests/conftest.py), and the whole GPU suite
# that way (expected there: only tests that pin or compare routes themselves can differ).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # tag, env..., -- pytest args
  tag=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout 900 python -m pytest "$@" -q < /dev/null 2>&1 | grep -E "passed|failed|^FAILED|AssertionError" | cut -c1-220 | tail -12 > gpurun_out/cross_$tag.txt
  echo "== $tag: $(tail -1 gpurun_out/cross_$tag.txt)"
}
run fuzz_small MMX_FUZZ_SEEDS=300 -- tests/test_gpu_fuzz.py -k "not wide"
run fuzz_mid MMX_FUZZ_SEEDS=300 MMX_FUZZ_JMAX=110 -- tests/test_gpu_fuzz.py -k "not wide"
run fuzz_mid_forced_wide MMX_TEST_ROUTE=prefer_wide MMX_FUZZ_SEEDS=300 MMX_FUZZ_JMAX=110 -- tests/test_gpu_fuzz.py -k "not wide"
run fuzz_wide MMX_FUZZ_WIDE_SEEDS=128 MMX_FUZZ_WIDE_JMAX=195 -- tests/test_gpu_fuzz.py -k "wide"
run suite_forced_wide MMX_TEST_ROUTE=prefer_wide -- tests -m gpu
