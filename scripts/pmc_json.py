#!/usr/bin/env python3
"""profiles/pmc_fused.json and profiles/pmc_jacobian.json (what bench.py quotes in roofline_fused.pmc / roofline.traffic)
from the counter_collection CSVs of a profile round:  pmc_json.py <round dir> <round tag>  -> <round dir>/pmc_*.json.
HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE (KB): on gfx950 FETCH_SIZE reports half of a wide streaming read
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section); the two counters come from separate passes."""
import collections, csv, glob, json, os, sys

out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc2*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        key = (row["Kernel_Name"].split("(")[0], int(row.get("Grid_Size", 0) or 0))
        agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
avg = lambda v: sum(v) / len(v) if v else None
for (name, grid), c in agg.items():
    g = lambda k: avg(c.get(k, []))
    if "fusedSolveKernel<6, 0, false, false, 0" in name and grid == 4096 * 256:  # (+ ", true>": the lazy-argument form)
        d = {
            "kernel": name.replace("void mmx::", "").replace(" ", ""), "config": "cfg2", "batch": 4096,
            "source": f"profiles/{tag}_pmc_bench.txt (rocprofv3 --pmc, separate passes, averages over the launches of one bench run)",
            "lds_bank_conflict_ratio": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
            "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT"), "lds_idx_active_cycles": g("SQ_LDS_IDX_ACTIVE"),
            "valu_active_over_wave_cycles": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
            "wait_any_over_wave_cycles": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
            "wait_inst_any_over_wave_cycles": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
            "valu_insts": g("SQ_INSTS_VALU"), "mfma_f32_insts": g("SQ_INSTS_VALU_MFMA_F32"), "lds_insts": g("SQ_INSTS_LDS"), "salu_insts": g("SQ_INSTS_SALU"),
            "round2": {"lds_bank_conflict_ratio": 0.226, "valu_active_over_wave_cycles": 0.148, "wait_any_over_wave_cycles": 0.666, "valu_insts": 6.31e8, "salu_insts": 2.179e8},
            "round5_before_the_compiler_flags_and_packed_chain": {"lds_bank_conflict_ratio": 0.261, "valu_active_over_wave_cycles": 0.132, "wait_any_over_wave_cycles": 0.671, "valu_insts": 6.92e8, "salu_insts": 2.575e8},
        }
        json.dump(d, open(os.path.join(out, "pmc_fused.json"), "w"), indent=1)
        print("pmc_fused:", {k: (round(v, 4) if isinstance(v, float) and v < 10 else v) for k, v in d.items() if k not in ("source", "round2")})
    if "fkJacobianKernel<true, 4, true>" in name and grid == 4096 * 256:
        w, f = g("WRITE_SIZE"), g("FETCH_SIZE")
        if w and f:
            hbm = (w + 2 * f) * 1024
            alg = 4096 * 100736
            d = {"config": "cfg2", "batch": 4096, "kernel": "fkJacobianKernel<true,4,true>", "hbm_bytes_per_launch": hbm, "write_size_kb": w, "fetch_size_kb_uncorrected": f,
                 "note": f"WRITE_SIZE + 2 x FETCH_SIZE (gfx950 FETCH_SIZE reports half of a wide streaming read, MI355X_MICROARCH.md), KB -> bytes; profiles/{tag}_pmc_bench.txt",
                 "ratio_to_algorithmic": hbm / alg}
            json.dump(d, open(os.path.join(out, "pmc_jacobian.json"), "w"), indent=1)
            print("pmc_jacobian:", d["hbm_bytes_per_launch"], d["ratio_to_algorithmic"])
