#!/bin/bash
# cfg5 at the test's batch (1024 of seed 20240611) and the bench's: parity max / count above the bound, per library variant
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
  MMX_LIB=$lib timeout 600 python - "$v" < /dev/null <<'PY'
import sys, torch, numpy as np
import bench
from momentum_amd._abi import GnOptions
from tests.test_gpu_baseline_parity import _solve_and_check
for cfg, B, n in (("cfg5", 1024, 1024), ("cfg2_all", 2048, 1024)):
    chk, out, db = _solve_and_check(torch, cfg, B, n)
    print(sys.argv[1], cfg, "max %.3g p99 %.3g median %.3g above %d" % (chk["max_rel_theta_vs_oracle_f64"], chk["p99_rel_theta_vs_oracle_f64"], chk["median_rel_theta_vs_oracle_f64"], chk["num_above_bound"]), chk.get("above_bound_instances"), chk.get("above_bound_float_oracle_rel"))
PY
  MMX_LIB=$lib timeout 300 python bench.py --config cfg5 --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 1024 < /dev/null 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['check']; print('$v', 'bench cfg5 %.4g solves/s' % d['value'], 'max %.3g above %s' % (c.get('max_rel_theta_vs_oracle_f64',-1), c.get('num_above_bound')))"
done
