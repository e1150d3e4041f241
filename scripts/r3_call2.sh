#!/bin/bash
# Round 3, GPU call 2: whole -m gpu suite after the clean-up + pivot floor, weak-damping report, FK noise, headline bench.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --no-header -p no:cacheprovider -x --deselect tests/test_gpu_weak_damping.py < /dev/null > gpurun_out/r3_suite.txt 2>&1
tail -5 gpurun_out/r3_suite.txt | cut -c1-300
grep -E "^FAILED|^ERROR" gpurun_out/r3_suite.txt | cut -c1-250 | head
timeout 600 python -m pytest tests/test_gpu_weak_damping.py -q --no-header -p no:cacheprovider < /dev/null > gpurun_out/r3_weak_all.txt 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3_weak_all.txt | cut -c1-250 | tail -40
timeout 120 python scripts/diag_fk_noise.py 2>&1 | tail -20
timeout 300 python bench.py --no-extra-configs > gpurun_out/r3_bench_quick.json 2> gpurun_out/r3_bench_quick.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_bench_quick.json"))
print("value %.4g ms/step %.3f check %s roofline %.3f cpu %.4g" % (d["value"], d["ms_per_step"], d["check"].get("max_rel_theta_vs_oracle_f64"), d["roofline"]["frac"], d["cpu_baseline"]["value"]))
print(d["cpu_baseline"]["sample"])
PY
timeout 120 python bench.py --no-extra-configs --no-cpu-baseline --config cfg5 --steps 4 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 %.4g solves/s check %s' % (d['value'], d['check'].get('max_rel_theta_vs_oracle_f64')))"
