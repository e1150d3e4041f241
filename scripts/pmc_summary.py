#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more files)."""
import csv
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "mmx" not in k and "fkJac" not in k:
        continue
    print(k[:90])
    for c, v in sorted(cs.items()):
        print(f"   {c:<26} n={len(v):<4} avg={sum(v)/len(v):.4g}")
