#!/bin/bash
# round 5, call F: the one-launch solve at four workgroups per CU (lifetime-shared LDS carve): A/B against the same carve at
# three, the whole GPU suite, the CPU probe of the oracle's kernels
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05f; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05f_ab main occ3 2>&1 | grep -v amdgpu.ids | tee $out/ab_occ.txt
BENCH_ARGS="--batch 32768" bash scripts/gpu_ab.sh r05f_ab32k main occ3 2>&1 | grep -v amdgpu.ids | tee -a $out/ab_occ.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=line -x < /dev/null 2>&1 | tail -25 > $out/pytest_gpu.txt; tail -25 $out/pytest_gpu.txt
bash scripts/probes/cpu_syrk_probe.sh > $out/cpu_probe.txt 2>&1; cat $out/cpu_probe.txt
