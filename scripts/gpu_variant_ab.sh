#!/bin/bash
# A/B of one build variant (VAR=name of MMX_BUILD_VARIANT, e.g. lookahead): headline bench of the variant against the default library on one
# box (each run carries its own parity check against the CPU oracle), then the variant's phase clocks.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # tag lib extra bench args...
  tag=$1; lib=$2; shift 2
  MMX_LIB=$GRAFT_REPO_ROOT/momentum_amd/$lib timeout 60 python bench.py --steps 30 --warmup 5 --no-extra-configs --no-cpu-baseline --check-instances 256 "$@" < /dev/null > gpurun_out/inv_$tag.json 2> gpurun_out/inv_$tag.err
  python - "$tag" "gpurun_out/inv_$tag.json" < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2]))
    print("%-14s value %.4g solves/s  ms/step %.3f  parity max %.3g  failed %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["check"].get("max_rel_theta_vs_oracle_f64",-1), d["check"]["failed_instances"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run ${VAR:-lookahead}_1 libmmx_hip_${VAR:-lookahead}.so
run main_1 libmmx_hip.so
# the per-rule instantiations (MMX_FUSED_PLAIN=1) on both libraries: plain Gauss-Newton, then the LM schedule of cfg3
export MMX_FUSED_PLAIN=1
run main_plain libmmx_hip.so
run ${VAR:-lookahead}_plain libmmx_hip_${VAR:-lookahead}.so
run main_lm_perrule libmmx_hip.so --config cfg3
unset MMX_FUSED_PLAIN
run main_lm libmmx_hip.so --config cfg3
# the staged kernels of the wide path (MMX_CHOL_LEAN=1, MMX_TREE_NE_WAVES=8) on cfg5
MMX_CHOL_LEAN=1 run main_cfg5_lean libmmx_hip.so --config cfg5 --steps 5 --warmup 2
MMX_TREE_NE_WAVES=8 run main_cfg5_ne8 libmmx_hip.so --config cfg5 --steps 5 --warmup 2
MMX_CHOL_LEAN=1 MMX_TREE_NE_WAVES=8 run main_cfg5_both libmmx_hip.so --config cfg5 --steps 5 --warmup 2
run main_cfg5 libmmx_hip.so --config cfg5 --steps 5 --warmup 2
if [ -n "$INV_MORE" ]; then
  run ${VAR:-lookahead}_2 libmmx_hip_${VAR:-lookahead}.so
  run ${VAR:-lookahead}_ls libmmx_hip_${VAR:-lookahead}.so --line-search 2
  run main_ls libmmx_hip.so --line-search 2
  MMX_LIB=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_${VAR:-lookahead}.so timeout 60 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_parity.py -q < /dev/null 2>&1 | grep -E "passed|failed|^FAILED|Error" | cut -c1-200 | tail -4
fi
[ -n "$SKIP_CLOCKS" ] || bash scripts/gpu_clocks.sh ${VAR:-lookahead}clk ${VAR:-lookahead} 2>&1 | grep -E "total|H cholesky|I solve|J solve|H.bc|H.b |H.d"
