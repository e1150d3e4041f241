#!/bin/bash
# one PMC pass over an arbitrary bench.py command line, per-kernel averages: bash scripts/gpu_pmc2.sh tag "COUNTERS" bench args...
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; counters=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $counters --output-format csv -d $out -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 "$@" < /dev/null > $out/bench.json 2> $out/err.txt
echo "rocprof rc=$?"
timeout 120 python $GRAFT_REPO_ROOT/scripts/pmc_summary.py $(find $out -name "*counter_collection.csv") < /dev/null | tee $out/summary.txt
