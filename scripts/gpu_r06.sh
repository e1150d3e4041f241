#!/bin/bash
# Round 6 GPU call script: bash scripts/gpu_r06.sh <step> [args]   (run through gpurun; outputs under gpurun_out/)
set -u
mkdir -p gpurun_out
step=${1:-mixed}
shift || true
case "$step" in
  mixed)  # the mixed-precision table (B instances per row, default 1024)
    timeout 1500 python scripts/diag_mixed_table.py "${1:-1024}" > gpurun_out/mixed_table.log 2>&1; tail -40 gpurun_out/mixed_table.log ;;
  mixedvar)  # the same table for A/B libraries: mixedvar <variant> ... (momentum_amd/libmmx_hip_<variant>.so)
    for v in "$@"; do
      MMX_LIB=$PWD/momentum_amd/libmmx_hip_$v.so MMX_TABLE_TAG=_$v timeout 900 python scripts/diag_mixed_table.py 1024 > gpurun_out/mixed_table_$v.log 2>&1
      echo "== $v"; grep -E "^cfg" gpurun_out/mixed_table_$v.log | cut -c1-60 | head -3
    done ;;
  mixedtol)  # the table for a list of CG tolerances: mixedtol <tol> ...
    for t in "$@"; do
      MMX_MIXED_TOL=$t MMX_TABLE_TAG=_tol$t timeout 900 python scripts/diag_mixed_table.py 1024 > gpurun_out/mixed_table_tol$t.log 2>&1
      echo "== tol $t"; grep -cE "^cfg" gpurun_out/mixed_table_tol$t.log
    done ;;
  mixedrate)  # rates of the precision routes + the mixed kernel's phase clocks
    timeout 900 python scripts/diag_mixed_rate.py > gpurun_out/mixed_rate.log 2>&1; cat gpurun_out/mixed_rate.log
    MMX_PHASE_CLOCKS=1 timeout 300 python scripts/diag_mixed_rate.py > gpurun_out/mixed_clocks.log 2>&1; cat gpurun_out/mixed_clocks.log ;;
  ab)  # A/B on one box: ab <variant> [more bench args]: the default library against momentum_amd/libmmx_hip_<variant>.so, alternating
    v=$1; shift
    B="python bench.py --no-extra-configs --no-cpu-baseline --no-measure-traffic --check-instances 1024 --steps 30 --warmup 3 --details gpurun_out/ab_details.json"
    for rep in 1 2; do
      for lib in default $v; do
        if [ $lib = default ]; then unset MMX_LIB; else export MMX_LIB=$PWD/momentum_amd/libmmx_hip_$v.so; fi
        for cfg in "--config cfg2" "--config cfg3 --steps 4" "--config cfg2 --line-search 2" "--config cfg2 --batch 32768 --steps 8"; do
          timeout 300 $B $cfg "$@" 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$lib', '$cfg', 'value %.4g' % l['value'], 'max_rel', l['check'].get('max_rel'), 'median', l['check'].get('median_rel'), l['check'].get('within_bound'))"
        done
      done
    done ;;
  tests)  # the GPU suite (optionally -k expression)
    if [ $# -eq 0 ]; then set -- tests; fi
    timeout 2400 python -m pytest "$@" -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log ;;
  bench)  # the driver's command (no arguments = the default run: its line and side file are the ones profiles/ keeps)
    if [ $# -eq 0 ]; then
      SECONDS=0; timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
      echo "Elapsed ${SECONDS}s" >> gpurun_out/bench_default.err; cp gpurun_out/bench_details.json gpurun_out/bench_default_details.json
      wc -c gpurun_out/bench_default.json; tail -c 1500 gpurun_out/bench_default.json; grep -E "^\[bench\]|Elapsed|Error|Traceback" gpurun_out/bench_default.err
    else
      timeout 900 python bench.py --details gpurun_out/bench_other_details.json "$@" > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 3000 gpurun_out/bench.json; tail -20 gpurun_out/bench.err
    fi ;;
  *) echo "unknown step $step"; exit 2 ;;
esac
