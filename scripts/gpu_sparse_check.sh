#!/bin/bash
# wide-route checks + cfg5 / cfg2_all numbers on one box: bash scripts/gpu_sparse_check.sh tag
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-sparse}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_trust_region.py tests/test_gpu_parameter_rows.py tests/test_gpu_function_weights.py tests/test_gpu_weak_damping.py tests/test_gpu_ellipsoid.py tests/test_gpu_joint_blocks.py tests/test_gpu_per_instance.py -q -x --no-header -p no:cacheprovider < /dev/null 2>&1 | tail -15 > gpurun_out/${tag}_pytest.txt
cat gpurun_out/${tag}_pytest.txt
for cfg in cfg5 cfg2_all; do
  timeout 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 256 < /dev/null > gpurun_out/${tag}_$cfg.json 2> gpurun_out/${tag}_$cfg.err
  python - "$cfg" "gpurun_out/${tag}_$cfg.json" < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-10s value %.4g solves/s  ms/step %.3f  parity max %.3g  failed %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["check"].get("max_rel_theta_vs_oracle_f64",-1), d["check"]["failed_instances"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --config cfg5 --steps 1 --warmup 0 --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null 2>&1 | grep -v "^{" | tail -30 > gpurun_out/${tag}_clocks.txt
cat gpurun_out/${tag}_clocks.txt
