#!/bin/bash
# one GPU call: noise diagnostic, full GPU suite, quick bench.  usage: bash scripts/gpu_check.sh tag [diag] [tests] [bench]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for what in "$@"; do
case $what in
diag)
  timeout 200 python scripts/diag_step_noise.py p128 1e-3 256 fused 2>&1 | grep -v amdgpu.ids > $out/diag_fused.txt
  timeout 200 python scripts/diag_step_noise.py p128 1e-3 256 wide 2>&1 | grep -v amdgpu.ids > $out/diag_wide.txt
  tail -7 $out/diag_fused.txt; tail -3 $out/diag_wide.txt;;
tests)
  timeout 900 python -m pytest tests -m gpu -q -x < /dev/null 2>&1 | tail -15 > $out/pytest_gpu.txt; tail -15 $out/pytest_gpu.txt;;
testsall)
  timeout 1200 python -m pytest tests -m gpu -q < /dev/null 2>&1 | tail -40 > $out/pytest_gpu.txt; tail -40 $out/pytest_gpu.txt;;
bench)
  timeout 600 python bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline --check-instances 1024 < /dev/null > $out/bench.json 2> $out/bench.err
  python - $out/bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value %.4g solves/s  ms/step %.3f  roofline %.4f  check %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v for k,v in d["check"].items() if not isinstance(v,(list,dict))}))
PY
  ;;
benchfull)
  timeout 1200 python bench.py < /dev/null > $out/bench_default.json 2> $out/bench_default.err
  python - $out/bench_default.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value %.4g roofline %.4f cpu %s" % (d["value"], d["roofline"]["frac"], d["cpu_baseline"]["value"]))
for k,v in d.get("configs",{}).items():
    c=v.get("check",{})
    print("  %-50s %.4g solves/s  check pass %s within %s/%s max %.3g" % (k, v.get("value",0), c.get("pass"), c.get("instances_within_bound"), c.get("instances_checked"), c.get("max_rel_theta_vs_oracle_f64",-1)))
PY
  ;;
esac
done
