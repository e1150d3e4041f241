#!/bin/bash
# A/B of library variants on the line-search / LM configurations: bash scripts/gpu_ab_ls.sh variant...
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
  for args in "--config cfg2 --line-search 2" "--config cfg2 --line-search 1" "--config cfg3 --batch 32768" "--config cfg2"; do
    MMX_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 2 --no-extra-configs --no-cpu-baseline --check-instances 512 $args < /dev/null > gpurun_out/abls.json 2> gpurun_out/abls.err
    python - "$v" "$args" < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open("gpurun_out/abls.json"))
    print("%-8s %-34s %.4g solves/s  parity max %.3g pass %s" % (sys.argv[1], sys.argv[2], d["value"], d["check"]["max_rel_theta_vs_oracle_f64"], d["check"]["pass"]))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done
