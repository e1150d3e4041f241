#!/bin/bash
# kernel trace of a short headline run: per-kernel durations and launch resources
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-trace}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extra-configs --no-cpu-baseline --check-instances 0 > $GRAFT_REPO_ROOT/gpurun_out/$tag/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/$tag/err.txt
cd $GRAFT_REPO_ROOT/gpurun_out/$tag
ls -R . | head -30
f=$(find . -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
print(rows[0].keys())
for r in rows:
    if "fusedSolve" in r["Kernel_Name"]:
        print({k:r[k] for k in r if k in ("Kernel_Name","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size","Grid_Size","Start_Timestamp","End_Timestamp")}, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, "us")
        break
PY
f2=$(find . -name "*kernel_stats.csv" | head -1); head -8 $f2
