#!/bin/bash
# fused solve against the wide path on a probe configuration: bash scripts/gpu_route.sh config...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in "$@"; do
for v in fused wide; do
  if [ $v = wide ]; then export MMX_FORCE_WIDE=1; unset MMX_PREFER_FUSED; else export MMX_PREFER_FUSED=1; unset MMX_FORCE_WIDE; fi
  timeout 300 python bench.py --config $c --steps 5 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 256 < /dev/null > gpurun_out/route_${c}_$v.json 2> gpurun_out/route_${c}_$v.err
  python - $c $v gpurun_out/route_${c}_$v.json < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[3]))
    print(sys.argv[1], sys.argv[2], "%.4g solves/s" % d["value"], "ms/step %.3f" % d["ms_per_step"], "parity", d["check"]["max_rel_theta_vs_oracle_f64"], "n", d["config"].get("solved_parameters"))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
done
done
