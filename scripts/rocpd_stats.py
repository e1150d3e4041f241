#!/usr/bin/env python3
"""Per-kernel summary (calls / total / avg / min / max / %) from a rocprofv3 rocpd sqlite database
or a *_kernel_trace.csv.  Used to produce the text summaries committed under profiles/.  Launches of
one kernel with different grid sizes are different workloads (bench.py also runs the J-assembly
kernel at the 32768-instance shard size for context), so a CSV trace is summarised per
(kernel, workgroups) pair."""
import csv
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, d.end - d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id"
    ).fetchall()
    return rows


def from_csv(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if name.startswith("void mmx::") or name.startswith("mmx::"):
            wg = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            name = f"[{wg} wg] " + name
        rows.append((name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return rows


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(list)
    for name, dur in rows:
        agg[name].append(dur)
    total = sum(sum(v) for v in agg.values())
    print(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:<70} {len(v):>6} {sum(v)/1e6:>10.3f} {sum(v)/len(v)/1e3:>10.2f} {min(v)/1e3:>10.2f} {max(v)/1e3:>10.2f} {100*sum(v)/total:>6.1f}")


if __name__ == "__main__":
    main()
