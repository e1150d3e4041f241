#!/bin/bash
# round profiles: kernel stats of the default bench run, PMC passes of the fused solve and the J-assembly kernel
# usage: bash scripts/gpu_profiles.sh r02
cd "$GRAFT_REPO_ROOT" || exit 1
r=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$r
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extra-configs --no-cpu-baseline --check-instances 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- $B < /dev/null > $out/bench_traced.json 2> $out/trace_err.txt
db=$(find $out/trace -name "*.db" | head -1)
[ -n "$db" ] && timeout 120 python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py "$db" < /dev/null > $out/${r}_bench_kernel_stats.txt
pass() { # name, counters
  timeout 600 rocprofv3 --pmc $2 --output-format csv -d $out/$1 -o pmc -- $B < /dev/null > /dev/null 2> $out/$1.err
}
pass pmc1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
pass pmc2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
pass pmc3 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
pass pmc4 "FETCH_SIZE"
pass pmc5 "WRITE_SIZE"
timeout 120 python - $out $r < /dev/null <<'PY'
import csv,sys,glob,collections,json
out,r=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name=row["Kernel_Name"]
        key=name.split("(")[0]+" ["+row.get("Grid_Size","?")+" threads]"
        agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out+"/"+r+"_pmc_bench.txt","w") as fo:
    for k,v in sorted(agg.items()):
        if "mmx::" not in k: continue
        fo.write(k+"\n")
        for c,vals in sorted(v.items()): fo.write("   %-28s n=%-3d avg=%.4g\n" % (c,len(vals),sum(vals)/len(vals)))
print(open(out+"/"+r+"_pmc_bench.txt").read()[:6000])
PY
head -12 $out/${r}_bench_kernel_stats.txt | cut -c1-140
