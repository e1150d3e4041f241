#!/bin/bash
# LDS bank conflicts of the one-launch solve, phase by phase (the verdicts' "0.25 and un-attributed" item): counter passes over
# launches of the CLOCKED instantiation (fusedSolveKernel<6, 2, ...>, MMX_PHASE_CLOCKS=1) whose workgroups end at successive
# workgroup-uniform stamps of their first iteration (MMX_PHASE_STOP=<stamp>, mmx_capi.hip armPhaseStop); a phase's share is the
# difference of two neighbouring passes.  One more pass without the stop gives the whole solve (ten iterations).
#   usage (GPU box): bash scripts/lds_conflicts_by_phase.sh [precision]  ->  gpurun_out/lds_phase/lds_conflicts_by_phase.txt
cd "$GRAFT_REPO_ROOT" || exit 1
prec=${1:-f32}
out=$GRAFT_REPO_ROOT/gpurun_out/lds_phase
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-extra-configs --no-cpu-baseline --check-instances 0 --no-measure-traffic --precision $prec --details $out/details.json"
C=${MMX_PHASE_COUNTERS:-"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"}
# stamps in program order: 1 FK | 2 units | 15 own sums | 3 subtree sums | 4 slot tables | 5 g | 6 H assembly (records, terms, pull) |
# 7 factor | 8 first solve | 9 refinement | 10 update (end of iteration 1)
for stop in 1 2 15 3 4 5 6 7 8 9 10 none; do
  if [ $stop = none ]; then
    MMX_PHASE_CLOCKS=1 timeout 200 rocprofv3 --pmc $C --output-format csv -d $out/p_$stop -o pmc -- $B < /dev/null > /dev/null 2> $out/p_$stop.err
  else
    MMX_PHASE_CLOCKS=1 MMX_PHASE_STOP=$stop timeout 200 rocprofv3 --pmc $C --output-format csv -d $out/p_$stop -o pmc -- $B < /dev/null > /dev/null 2> $out/p_$stop.err
  fi
done
python - $out $prec < /dev/null <<'PY' | tee $out/lds_conflicts_by_phase.txt
import collections, csv, glob, sys
out, prec = sys.argv[1], sys.argv[2]
order = [("1", "A,B forward kinematics"), ("2", "C units"), ("15", "D own sums"), ("3", "D subtree sums"), ("4", "E slot tables"), ("5", "F g = J^T r"),
         ("6", "G H assembly (records, terms, pull)"), ("7", "H factor (panels, chain, MFMA updates)"), ("8", "I first solve"), ("9", "J refinement"), ("10", "K update"), ("none", "whole solve (ten iterations)")]
rows = {}
for stop, _ in order:
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/p_{stop}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "fusedSolveKernel<6, 2" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    rows[stop] = {k: sum(v) / len(v) for k, v in agg.items()}
names = ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"]
print(f"LDS bank conflicts by phase, first iteration of the clocked instantiation (precision {prec}): per-launch averages, a phase = the difference")
print("of the passes cut after it and after its predecessor (scripts/lds_conflicts_by_phase.sh)")
print("%-42s %14s %14s %8s %12s %14s %14s" % ("phase", "BANK_CONFLICT", "IDX_ACTIVE", "ratio", "INSTS_LDS", "WAIT_INST_LDS", "WAVE_CYCLES"))
prev = {k: 0.0 for k in names}
tot1 = rows.get("10", {})
for stop, label in order:
    r = rows.get(stop) or {}
    if not r:
        print("%-42s (no data)" % label)
        continue
    if stop == "none":
        d = r
    else:
        d = {k: r.get(k, 0.0) - prev.get(k, 0.0) for k in names}
        prev = {k: r.get(k, 0.0) for k in names}
    ratio = d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"] if d.get("SQ_LDS_IDX_ACTIVE") else float("nan")
    share = 100.0 * d["SQ_LDS_BANK_CONFLICT"] / tot1["SQ_LDS_BANK_CONFLICT"] if stop != "none" and tot1.get("SQ_LDS_BANK_CONFLICT") else float("nan")
    print("%-42s %14.4g %14.4g %8.3f %12.4g %14.4g %14.4g   %5.1f %% of the iteration's conflicts" % (
        label, d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"], ratio, d.get("SQ_INSTS_LDS", 0), d.get("SQ_WAIT_INST_LDS", 0), d.get("SQ_WAVE_CYCLES", 0), share))
PY
