#!/bin/bash
# round 5, call Z: three register-hungry variants again now that the solve kernels do not spill (compiled without machine LICM /
# loop strength reduction): unit payload requested before FK (earlyunit), the transform's walk requested a phase ahead
# (csrpipe), the solves' chain cut to the newest block (ahead)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05z; mkdir -p $out
export TMPDIR=/tmp
bash scripts/gpu_ab.sh r05z_h main earlyunit csrpipe ahead 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05z_c3 main earlyunit csrpipe ahead 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
