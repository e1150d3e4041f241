#!/bin/bash
# development round trip: GPU tests (all failures listed), A/B throughput of library variants, phase clocks
# usage: bash scripts/gpu_dev.sh tag "pytest args" variant...
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; pyargs=$2; shift; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$pyargs" != "none" ]; then
timeout 1200 python -m pytest tests -m gpu -q $pyargs < /dev/null > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/${tag}_tests.log | tail -25
fi
bash scripts/gpu_ab.sh ${tag}_ab "$@"
bash scripts/gpu_clocks.sh ${tag}_clk main
