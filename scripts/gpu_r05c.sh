#!/bin/bash
# round 5, call C: A/B of the diagnostics' cost (registers instead of LDS), precision tests, the weak-damping table, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05c; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05c_ab main nodiag 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
timeout 900 python -m pytest tests/test_gpu_precision.py tests/test_bench_two_ranks.py tests/test_gpu_baseline_parity.py -m gpu -q --tb=line < /dev/null 2>&1 | tail -30 > $out/pytest_sel.txt; tail -30 $out/pytest_sel.txt
timeout 1200 python bench.py < /dev/null > $out/bench_default.json 2> $out/bench_default.err; tail -12 $out/bench_default.err; wc -c $out/bench_default.json
cp gpurun_out/bench_details.json $out/bench_details.json
