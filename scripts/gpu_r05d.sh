#!/bin/bash
# round 5, call D: the wide route's normal equations at two workgroups per CU (A/B on cfg5), parity, traffic measured in the run
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05d; mkdir -p $out
export TMPDIR=/tmp
BENCH_ARGS="--config cfg5" bash scripts/gpu_ab.sh r05d_ab main necompact_off 2>&1 | grep -v amdgpu.ids | tee $out/ab_cfg5.txt
for b in 2048 4096; do
  timeout 300 python bench.py --config cfg5 --batch $b --steps 10 --warmup 3 --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg5 batch', d['config']['batch_per_gpu'], 'solves/s %.4g' % d['value'])" | tee -a $out/ab_cfg5.txt
done
timeout 900 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_tile_structure.py tests/test_gpu_precision.py -m gpu -q --tb=line -k "config5 or tile or auto or lm_schedule" < /dev/null 2>&1 | tail -15 > $out/pytest_sel.txt; tail -15 $out/pytest_sel.txt
timeout 600 python bench.py --measure-traffic --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null > $out/bench_traffic.json 2> $out/bench_traffic.err; python -c "import json;d=json.load(open('$out/bench_traffic.json'));print(d['roofline'])"
