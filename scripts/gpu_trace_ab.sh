#!/bin/bash
# cfg5 numbers + per-kernel times for library variants: bash scripts/gpu_trace_ab.sh tag variant...
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
  MMX_LIB=$lib timeout 300 python bench.py --config cfg5 --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 256 < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v cfg5 %.4g solves/s  parity %.3g' % (d['value'], d['check'].get('max_rel_theta_vs_oracle_f64',-1)))"
  cd /tmp && MMX_LIB=$lib rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_${v}_prof -o cfg5 -- python $GRAFT_REPO_ROOT/bench.py --config cfg5 --steps 4 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT && python scripts/rocpd_stats.py gpurun_out/${tag}_${v}_prof/cfg5_results.db 2>/dev/null | cut -c1-150 | grep -E "Factor|Finish|treeNormal|treeRefine"
done
