#!/bin/bash
# kernel trace (per-kernel durations) of a short bench run: bash scripts/gpu_trace.sh tag [bench args...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-trace}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extra-configs --no-cpu-baseline --check-instances 0 "$@" < /dev/null > $out/bench.json 2> $out/err.txt
echo "rocprof rc=$?"
db=$(find $out -name "*.db" | head -1)
if [ -n "$db" ]; then timeout 120 python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py "$db" < /dev/null | head -24 | tee $out/kernel_stats.txt; fi
