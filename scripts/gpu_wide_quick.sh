#!/bin/bash
# wide-route tests, cfg5 / cfg2_all / headline numbers, cfg5 kernel trace: bash scripts/gpu_wide_quick.sh tag
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-wq}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_trust_region.py tests/test_gpu_weak_damping.py tests/test_gpu_parameter_rows.py -q -x --no-header -p no:cacheprovider < /dev/null 2>&1 | grep -E "^E  .*(Assert|assert)|FAILED|passed|failed" | cut -c1-300 | tail -8
bash scripts/gpu_ab.sh $tag main 2>&1 | tail -1
bash scripts/gpu_sparse_ab.sh $tag main 2>&1 | grep -v "^{" | tail -10
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -o cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 4 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python scripts/rocpd_stats.py gpurun_out/${tag}_prof/cfg5_results.db 2>/dev/null | cut -c1-150 | head -7
