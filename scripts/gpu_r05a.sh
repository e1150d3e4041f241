#!/bin/bash
# round 5, call A: calibration data of the precision estimate, the LM schedule's step histories, the GPU suite, a quick bench
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05a; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python scripts/diag_precision.py estimate < /dev/null 2>&1 | grep -v amdgpu.ids > $out/estimate.txt; tail -50 $out/estimate.txt
timeout 300 python scripts/diag_precision.py lm 65536 16384 < /dev/null 2>&1 | grep -v amdgpu.ids > $out/lm.txt; tail -5 $out/lm.txt
timeout 1200 python -m pytest tests -m gpu -q < /dev/null 2>&1 | tail -60 > $out/pytest_gpu.txt; tail -60 $out/pytest_gpu.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-extra-configs --check-instances 1024 < /dev/null > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json | head -c 3000
