#!/bin/bash
# phase clocks of the fused kernel for library variants: bash scripts/gpu_clocks.sh tag variant...
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
  MMX_LIB=$lib MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 ${BENCH_ARGS} < /dev/null > gpurun_out/${tag}_$v.json 2> gpurun_out/${tag}_$v.txt
  echo "== $v"; tail -26 gpurun_out/${tag}_$v.txt
done
