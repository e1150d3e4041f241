#!/bin/bash
# GPU call: full GPU test suite (all failures listed), headline bench, phase clocks of the fused kernel.
# usage: gpurun -- 'bash scripts/gpu_quick.sh [tag] [pytest args...]'
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-quick}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q "$@" > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/${tag}_tests.log | tail -25
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra-configs --cpu-sample 2048 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${tag}_bench.json"))
    print("value %.4g solves/s  ms/step %.3f  parity max %.3g  jac frac %.3f" % (d["value"], d["ms_per_step"], d["check"].get("max_rel_theta_vs_oracle_f64",-1), d["roofline"]["frac"]))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/${tag}_bench.err").read()[-2000:])
PY
MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 > /dev/null 2> gpurun_out/${tag}_phase_clocks.txt
tail -26 gpurun_out/${tag}_phase_clocks.txt
