#!/bin/bash
# GPU call: test suite, default bench line, phase clocks of the fused kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/gpu_tests.log
tail -15 gpurun_out/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
tail -c 6000 gpurun_out/bench_default.json
MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 > /dev/null 2> gpurun_out/phase_clocks_cfg2.txt
tail -30 gpurun_out/phase_clocks_cfg2.txt
