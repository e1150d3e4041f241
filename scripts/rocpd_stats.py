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


def _label(name, grid_x, wg_x):
    """Launches of one kernel with different grids are different workloads: one line per (kernel, workgroups)."""
    if "mmx" in name:  # (the sqlite database holds mangled names: _ZN3mmx...)
        return f"[{int(grid_x) // max(int(wg_x), 1)} wg] " + name
    return name


def from_db(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    # rocprofv3's rocpd schema suffixes its tables with the run's GUID; views without the suffix exist in newer versions
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
    disp = next((t for t in tables if t == "rocpd_kernel_dispatch"), None) or next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    sym = next((t for t in tables if t == "rocpd_info_kernel_symbol"), None) or next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = {r[1] for r in cur.execute(f"pragma table_info({disp})")}
    grid = "d.grid_size_x" if "grid_size_x" in cols else ("d.grid_x" if "grid_x" in cols else "0")
    wg = "d.workgroup_size_x" if "workgroup_size_x" in cols else ("d.workgroup_x" if "workgroup_x" in cols else "1")
    rows = cur.execute(f"select s.kernel_name, d.end - d.start, {grid}, {wg} from {disp} d join {sym} s on d.kernel_id = s.id").fetchall()
    return [(_label(name, g, w), dur) for name, dur, g, w in rows]


def from_csv(path):
    rows = []
    for r in csv.DictReader(open(path)):
        name = _label(r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"])
        rows.append((name, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return rows


def demangle(names):
    import shutil
    import subprocess

    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    path = sys.argv[1]
    rows = from_db(path) if path.endswith(".db") else from_csv(path)
    agg = defaultdict(list)
    for name, dur in rows:
        agg[name].append(dur)
    pretty = demangle([n.split("] ", 1)[-1] for n in agg])
    agg = {(n.split("] ", 1)[0] + "] " if "] " in n else "") + pretty[n.split("] ", 1)[-1]]: v for n, v in agg.items()}
    total = sum(sum(v) for v in agg.values())
    print(f"{'kernel':<70} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:<70} {len(v):>6} {sum(v)/1e6:>10.3f} {sum(v)/len(v)/1e3:>10.2f} {min(v)/1e3:>10.2f} {max(v)/1e3:>10.2f} {100*sum(v)/total:>6.1f}")


if __name__ == "__main__":
    main()
