#!/bin/bash
# A/B of library variants on one box: bash scripts/gpu_ab.sh tag variant1 variant2 ...  ("main" = libmmx_hip.so)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
  MMX_LIB=$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline --check-instances 256 ${BENCH_ARGS} < /dev/null > gpurun_out/${tag}_$v.json 2> gpurun_out/${tag}_$v.err
  python - "$v" "gpurun_out/${tag}_$v.json" < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-10s value %.4g solves/s  ms/step %.3f  parity max %.3g  failed %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["check"].get("max_rel",-1), d["check"]["failed_instances"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
done
