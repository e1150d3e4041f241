#!/bin/bash
# round 5, call V: where the waves of a workgroup land (probe), the logical wave index rotated per workgroup (variant rotwave),
# the packed panel updates on the wide route (main against variant nopkchain on cfg5)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05v2; mkdir -p $out
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 scripts/probes/wave_simd_placement.hip -o /tmp/wave_simd 2> /dev/null && timeout 120 /tmp/wave_simd 2>&1 | grep -v amdgpu.ids | tee $out/wave_simd_placement.txt
bash scripts/gpu_ab.sh r05v2_h main rotwave nopkchain 2>&1 | grep -v amdgpu.ids | tee $out/ab.txt
BENCH_ARGS="--config cfg3" bash scripts/gpu_ab.sh r05v2_c3 main rotwave 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
BENCH_ARGS="--config cfg5" bash scripts/gpu_ab.sh r05v2_c5 main nopkchain 2>&1 | grep -v amdgpu.ids | tee -a $out/ab.txt
