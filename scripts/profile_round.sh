#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace of the default bench command and the
# PMC passes of the J-assembly kernel (counters in their own runs, no tracing), then writes
# text/JSON summaries under gpurun_out/profiles_<tag>/ (copied into profiles/ by hand).
set -u
TAG=${1:-r01}
BATCH=${2:-4096}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
python $R/scripts/rocpd_stats.py $O/trace/bench_kernel_trace.csv > $O/${TAG}_bench_kernel_stats.txt 2>&1
for pass in "WRITE_SIZE" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --output-format csv -d $O/pmc -o $name -- python $R/scripts/jac_only.py --batch $BATCH --launches 5 > $O/pmc_$name.log 2>&1
done
python $R/scripts/pmc_summary.py $O/pmc/*_counter_collection.csv > $O/${TAG}_pmc_jacobian_B${BATCH}.txt 2>&1
python - <<PY
import csv, glob, json
vals = {}
for f in glob.glob("$O/pmc/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fkJacobianKernel<true" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in vals.items()}
# WRITE_SIZE / FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide streaming reads by 2x
# (MI355X_MICROARCH.md, HBM section) -- the read side of this kernel is ~1% of its traffic.
traffic = 1024.0 * (avg.get("WRITE_SIZE", 0.0) + 2.0 * avg.get("FETCH_SIZE", 0.0))
json.dump({"config": "cfg2", "batch": $BATCH, "hbm_bytes_per_launch": traffic, "write_size_kib": avg.get("WRITE_SIZE"),
           "fetch_size_kib_raw": avg.get("FETCH_SIZE"), "counters": avg}, open("$O/pmc_jacobian.json", "w"), indent=1)
print(open("$O/pmc_jacobian.json").read()[:400])
PY
tail -1 $O/bench_under_rocprof.log | cut -c1-300
head -8 $O/${TAG}_bench_kernel_stats.txt
