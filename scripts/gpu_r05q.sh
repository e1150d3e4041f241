#!/bin/bash
# round 5, call Q: the transform walked from row records (one L2 round trip per walk): A/B on the headline and cfg5, parity
# subset, the oracle's phase probe on the box's CPU
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05q; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05q_ab main norowrec 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg5" bash scripts/gpu_ab.sh r05q_ab5 main norowrec 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py tests/test_gpu_determinism.py tests/test_gpu_fuzz.py tests/test_real_rig.py -m gpu -q --tb=line < /dev/null 2>&1 | tail -6 > $out/pytest_sel.txt; tail -6 $out/pytest_sel.txt
timeout 600 python scripts/probes/oracle_phase_probe.py > $out/oracle_phase_probe.txt 2>&1; cat $out/oracle_phase_probe.txt
