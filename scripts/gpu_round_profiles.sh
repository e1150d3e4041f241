#!/bin/bash
# end-of-round evidence: full GPU suite, the default bench line, kernel traces and PMC passes of the headline (cfg2) and
# of the wide solve (cfg5).  usage: bash scripts/gpu_round_profiles.sh r02   -> gpurun_out/round_r02/
cd "$GRAFT_REPO_ROOT" || exit 1
r=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/round_$r
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q < /dev/null 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -6 > $out/pytest_gpu.txt
# the DRIVER's command (no arguments): its line and ITS side file are the ones profiles/ keeps (round 5's committed side file was
# a later profiling pass's: every other bench.py call below writes its details elsewhere)
timeout 1200 python bench.py --details $out/${r}_bench_details.json < /dev/null > $out/${r}_bench_default.json 2> $out/bench_default.err
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extra-configs --no-cpu-baseline --check-instances 0 --no-measure-traffic --details $out/profiling_pass_details.json"
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace2 -o t -- $B < /dev/null > /dev/null 2> $out/trace2.err
db=$(find $out/trace2 -name "*.db" | head -1)
[ -n "$db" ] && timeout 120 python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py "$db" < /dev/null > $out/${r}_bench_kernel_stats.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace5 -o t -- $B --config cfg5 --steps 3 --warmup 1 < /dev/null > /dev/null 2> $out/trace5.err
db=$(find $out/trace5 -name "*.db" | head -1)
[ -n "$db" ] && timeout 120 python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py "$db" < /dev/null > $out/${r}_cfg5_kernel_stats.txt
pass() { # dir, counters, bench args...
  d=$1; c=$2; shift; shift
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $out/$d -o pmc -- $B "$@" < /dev/null > /dev/null 2> $out/$d.err
}
pass pmc2a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
pass pmc2b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
pass pmc2c "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
pass pmc2d "FETCH_SIZE"
pass pmc2e "WRITE_SIZE"
# the mixed-precision instantiation (fusedSolveKernel<6,0,false,false,-1,false,true>): utilisation counters and a kernel trace
pass pmcMa "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" --precision mixed
pass pmcMb "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" --precision mixed
pass pmcMc "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE" --precision mixed
timeout 300 rocprofv3 --kernel-trace --stats -d $out/traceM -o t -- $B --precision mixed < /dev/null > /dev/null 2> $out/traceM.err
db=$(find $out/traceM -name "*.db" | head -1)
[ -n "$db" ] && timeout 120 python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py "$db" < /dev/null > $out/${r}_mixed_kernel_stats.txt
pass pmc5a "FETCH_SIZE" --config cfg5 --batch 2048 --steps 2 --warmup 1
pass pmc5b "WRITE_SIZE" --config cfg5 --batch 2048 --steps 2 --warmup 1
pass pmc5c "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT" --config cfg5 --batch 2048 --steps 2 --warmup 1
timeout 120 python - $out $r < /dev/null <<'PY'
import csv,sys,glob,collections
out,r=sys.argv[1],sys.argv[2]
for tag,pat in (("pmc_bench","pmc2"),("pmc_cfg5","pmc5"),("pmc_mixed","pmcM")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out+"/"+pat+"*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            key=row["Kernel_Name"].split("(")[0]+" ["+row.get("Grid_Size","?")+" threads]"
            agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    with open(out+"/"+r+"_"+tag+".txt","w") as fo:
        for k,v in sorted(agg.items()):
            if "mmx::" not in k: continue
            fo.write(k+"\n")
            for c,vals in sorted(v.items()): fo.write("   %-28s n=%-3d avg=%.4g\n" % (c,len(vals),sum(vals)/len(vals)))
PY
timeout 60 python $GRAFT_REPO_ROOT/scripts/pmc_json.py $out $r < /dev/null
# two-line derivation of the roofline fractions from the kernel traces alone: algorithmic bytes per instance (bench.py's
# algorithmic_bytes_per_instance: write J and r, read theta / constraint payload / parents) x instances per launch
# / average duration of the J-assembly kernel at that grid / 8 TB/s
timeout 60 python - $out $r < /dev/null <<'PY'
import re,sys
out,r=sys.argv[1],sys.argv[2]
def avg_us(path, wg):
    for line in open(path):
        if line.startswith(f"[{wg} wg]") and "fkJacobianKernel<true" in line:
            f=line.split()
            # columns: ... calls total_ms avg_us min_us max_us pct (from the right)
            m=re.search(r"fkJacobianKernel<[^>]*>", line)
            return int(f[-6]), float(f[-4]), (m.group(0).replace(" ", "") if m else "fkJacobianKernel<true,...>")
    return None
rows=[]
for tag,path,wg,M,P,Kp,Ko in (("cfg2 (BASELINE configs[1], B = 4096)", f"{out}/{r}_bench_kernel_stats.txt", 4096, 192, 128, 16, 16),
                            ("cfg5 (BASELINE configs[4], B = 8192)", f"{out}/{r}_cfg5_kernel_stats.txt", 8192, 900, 300, 150, 50)):
    try:
        got=avg_us(path, wg)
    except OSError:
        got=None
    if not got: continue
    calls,us,kname=got
    bpi=4*(M*P+M)+4*P+4*(7*Kp+9*Ko)+4*(Kp+Ko)
    gbs=bpi*wg/(us*1e-6)/1e9
    rows.append(f"{tag}: {kname} at {wg} workgroups: {calls} launches, average {us:.2f} us; {bpi} B per instance x {wg} = {bpi*wg} B per launch -> {gbs:.0f} GB/s = {gbs/8000:.3f} of the 8 TB/s HBM peak")
open(f"{out}/{r}_roofline.txt","w").write("\n".join(rows)+"\n")
print("\n".join(rows))
PY
cat $out/pytest_gpu.txt; head -8 $out/${r}_bench_kernel_stats.txt | cut -c1-140; head -8 $out/${r}_cfg5_kernel_stats.txt | cut -c1-140
cd $GRAFT_REPO_ROOT
MMX_TABLE_TAG=_round timeout 900 python scripts/diag_mixed_table.py 1024 < /dev/null 2>&1 | grep -v amdgpu.ids > $out/mixed_table.txt
cp gpurun_out/mixed_table_round.json $out/${r}_weak_damping.json 2> /dev/null
timeout 600 python scripts/diag_mixed_rate.py < /dev/null 2>&1 | grep -v amdgpu.ids > $out/${r}_mixed_rate.txt
MMX_PHASE_CLOCKS=1 timeout 300 python scripts/diag_mixed_rate.py < /dev/null 2>&1 | grep -v amdgpu.ids > $out/${r}_mixed_phase_clocks.txt
MMX_PHASE_CLOCKS=1 timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --no-measure-traffic --check-instances 0 --steps 1 --warmup 0 --line-search 2 --details $out/profiling_pass_details.json < /dev/null 2>&1 | grep -v "amdgpu.ids\|^{" > $out/${r}_phase_clocks.txt
timeout 300 python scripts/diag_precision.py lm 65536 16384 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -2 > $out/lm_steps.txt
(timeout 120 python scripts/diag_lm_estimate.py < /dev/null 2>&1 | grep cfg3; timeout 900 python scripts/diag_precision.py estimate < /dev/null 2>&1 | grep -E "^cfg|^p128") > $out/${r}_precision_estimate.txt
timeout 300 python bench.py --config cfg4 --steps 6 --warmup 2 --no-extra-configs --no-cpu-baseline --check-instances 256 < /dev/null > $out/${r}_bench_cfg4.json 2> /dev/null
timeout 200 python scripts/diag_determinism.py 6 < /dev/null 2>&1 | grep -v amdgpu.ids > $out/determinism.txt
bash scripts/resource_usage.sh > $out/${r}_kernel_resource_usage.txt 2>&1
# LDS bank conflicts by phase (counter passes over launches cut after successive stamps), eager against graph replay, AUTO's trace
bash scripts/lds_conflicts_by_phase.sh f32 > /dev/null 2>&1; cp gpurun_out/lds_phase/lds_conflicts_by_phase.txt $out/${r}_lds_conflicts_by_phase.txt 2> /dev/null
rm -rf gpurun_out/lds_phase/p_*
timeout 600 python scripts/diag_graph_rate.py < /dev/null 2>&1 | grep -v amdgpu.ids > $out/${r}_graph_rate.txt
(cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats -d $out/traceA -o t -- $B --precision auto < /dev/null > /dev/null 2> $out/traceA.err
 db=$(find $out/traceA -name "*.db" | head -1); [ -n "$db" ] && timeout 120 python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py "$db" < /dev/null > $out/${r}_auto_kernel_stats_after.txt)
# gpurun copies back at most 64 MiB: the raw traces and counter CSVs stay on the box, their summaries above are what is kept
rm -rf $out/trace2 $out/trace5 $out/traceM $out/traceA $out/pmc2? $out/pmcM? $out/pmc5? gpurun_out/lds_phase/p_*
python - $out/${r}_bench_default.json < /dev/null <<'PY'
import json,sys
raw=open(sys.argv[1]).read()
d=json.loads(raw)
print("bytes", len(raw), "value", d["value"], "roofline", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["single_thread_value"])
for k,v in d["configs"].items():
    print(" ", k, v.get("solves_per_s"), v.get("within_bound"), v.get("max_rel"), v.get("pass"), v.get("gpu_over_cpu"))
PY
