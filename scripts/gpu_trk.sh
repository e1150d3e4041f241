#!/bin/bash
# tracker-shaped config through the fused general rows and through the explicit-Jacobian kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 1 0; do
  MMX_FUSED_GENERAL=$m timeout 300 python bench.py --config cfg2_tracker --steps 10 --warmup 2 --no-extra-configs --no-cpu-baseline --check-instances 1024 < /dev/null > gpurun_out/trk_$m.json 2> gpurun_out/trk_$m.err
  python - $m gpurun_out/trk_$m.json < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("fused_general=%s value %.4g solves/s ms/step %.3f check %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: d["check"][k] for k in ("max_rel_theta_vs_oracle_f64","p99_rel_theta_vs_oracle_f64","failed_instances") if k in d["check"]}))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(sys.argv[2].replace(".json",".err")).read()[-1500:])
PY
done
