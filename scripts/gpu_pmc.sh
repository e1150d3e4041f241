#!/bin/bash
# PMC pass over a short headline run: bash scripts/gpu_pmc.sh tag "COUNTER1 COUNTER2 ..." [lib variant]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; counters=$2; v=${3:-main}
if [ "$v" = "main" ]; then lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip.so; else lib=$GRAFT_REPO_ROOT/momentum_amd/libmmx_hip_$v.so; fi
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 -L < /dev/null 2>/dev/null | grep -i -E "icache|ifetch|INST_CACHE" | head -40 > $out/counters_available.txt
MMX_LIB=$lib timeout 600 rocprofv3 --pmc $counters --output-format csv -d $out -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 < /dev/null > $out/bench.json 2> $out/err.txt
echo "rocprof rc=$?"
f=$(find $out -name "*counter_collection.csv" | head -1)
echo "file: $f"
if [ -n "$f" ]; then
python - "$f" < /dev/null <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "fusedSolve" in k or "fkJacobian" in k or "jacobianColumns" in k:
        print(k)
        for c,vals in sorted(v.items()): print("   %-28s n=%d avg=%.4g" % (c,len(vals),sum(vals)/len(vals)))
PY
fi
head -40 $out/counters_available.txt
