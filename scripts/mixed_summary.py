"""Prints one line per row of a gpurun_out/mixed_table*.json (scripts/diag_mixed_table.py)."""
import json
import sys

for path in sys.argv[1:]:
    t = json.load(open(path))
    print("==", path)
    for k in sorted(t, key=lambda k: (k.split()[0], -float(k.split()[1].split("=")[1]), k.split()[2])):
        r = t[k]
        sp = r["solves_per_s"]
        print(
            f"{k:38s} stable {r['sane_and_stable']:5d} f32> {r['f32_above_1e-5']:4d} mix> {r['mixed_above_1e-5']:4d} max {r['mixed_max_rel'] or 0:.2e} med {r['mixed_median_rel'] or 0:.2e} "
            f"cg-unconv {r['mixed_cg_unconverged']:4d} cg/it {r.get('mixed_operator_applications_per_iteration', 0):.2f} otherLS {r['mixed_other_line_search_decision']:4d} esc64 {r['auto_escalated_f64']:4d} auto> {r['auto_above_1e-5']:4d} | f32 {sp['f32']:.3g} mixed {sp['mixed']:.3g} auto {sp['auto']:.3g} f64 {sp['f64']:.3g}"
        )
