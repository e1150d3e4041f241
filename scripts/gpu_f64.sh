#!/bin/bash
# the double route: tests + timing.  usage: bash scripts/gpu_f64.sh [notests]
cd "$GRAFT_REPO_ROOT" || exit 1
if [ "$1" != "notests" ]; then
  timeout 600 python -m pytest tests/test_gpu_f64.py tests/test_real_rig.py -m gpu -q 2>&1 | grep -a "^E  *Assert\|^E  *assert\|passed\|failed\|^FAILED" | cut -c1-300
fi
for args in "" "--line-search 2 --lambda 1e-5"; do
timeout 300 python bench.py --dtype f64 --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 256 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f64 cfg2@4096 [$args]: %.4g solves/s ms/step %.2f' % (d['value'], d['ms_per_step']), {k:v for k,v in d['check'].items() if k in ('max_rel_theta_vs_oracle_f64','pass','within_bound')})"
done
