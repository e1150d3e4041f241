#!/bin/bash
# round 5, call G: whole GPU suite on the lifetime-shared carve, phase clocks of the one-launch solve, A/B 4 vs 3 workgroups
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05g; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=line < /dev/null 2>&1 | tail -25 > $out/pytest_gpu.txt; tail -25 $out/pytest_gpu.txt
MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null > /dev/null 2> $out/phase_clocks.txt; grep -A 30 "phase clocks" $out/phase_clocks.txt | head -40
bash scripts/gpu_ab.sh r05g_ab main occ3 2>&1 | grep -v amdgpu.ids | tee $out/ab_occ.txt
