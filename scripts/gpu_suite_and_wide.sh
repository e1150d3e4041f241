#!/bin/bash
# full GPU suite (summary + assertion lines), the headline A/B line, cfg5 / cfg2_all numbers, cfg5 kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-suite}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider < /dev/null 2>&1 | grep -E "^E  .*(Assert|assert)|FAILED|passed|failed" | cut -c1-300 | tail -40 > gpurun_out/${tag}_pytest_gpu.txt
cat gpurun_out/${tag}_pytest_gpu.txt
bash scripts/gpu_ab.sh $tag main 2>&1 | tail -2
bash scripts/gpu_sparse_ab.sh $tag main 2>&1 | grep -v "^{" | tail -14
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 4 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocpd_stats.py gpurun_out/${tag}_prof 2>/dev/null | head -12
