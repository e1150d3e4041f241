#!/bin/bash
# A/B of library variants on the wide-route configs: bash scripts/gpu_sparse_ab.sh tag variant...
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
  for cfg in cfg5 cfg2_all; do
  MMX_LIB=$lib timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 256 < /dev/null > gpurun_out/${tag}_${v}_$cfg.json 2> gpurun_out/${tag}_${v}_$cfg.err
  python - "$v $cfg" "gpurun_out/${tag}_${v}_$cfg.json" < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-18s value %.4g solves/s  ms/step %.3f  parity max %.3g  failed %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["check"].get("max_rel_theta_vs_oracle_f64",-1), d["check"]["failed_instances"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
done
done
for v in "$@"; do
  if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
  echo "== $v"; MMX_LIB=$lib MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --config cfg5 --steps 1 --warmup 0 --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null 2>&1 | grep -E "factor|solve|tree NE: F|G extras"
done
