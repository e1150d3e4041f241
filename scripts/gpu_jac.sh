#!/bin/bash
# J-assembly A/B: two-kernel (default) vs one-kernel form, at B = 4096 and 32768 (bench.py's roofline section)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
for mode in two one; do
  if [ $mode = one ]; then export MMX_JAC_ONE_KERNEL=1; else unset MMX_JAC_ONE_KERNEL; fi
  timeout 300 python bench.py --steps 3 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null > gpurun_out/${tag}_$mode.json 2> gpurun_out/${tag}_$mode.err
  python - "$mode" "gpurun_out/${tag}_$mode.json" < /dev/null <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[2])); r=d["roofline"]
    print("%-4s B=4096: %.1f us %.0f GB/s frac %.3f (recorded-event %.1f us) | store pattern %.0f fill %.0f | B=32768: %.1f us frac %.3f" % (sys.argv[1], 1e3*r["ms_per_launch"], r["achieved"], r["frac"], 1e3*r["ms_per_launch_recorded_events"], r["store_pattern_gbs"], r["fill_same_bytes_gbs"], 1e3*r["at_batch_32768"]["ms_per_launch"], r["at_batch_32768"]["frac"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
