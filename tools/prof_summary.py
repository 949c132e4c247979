"""Summarise rocprofv3 output directories (kernel-trace stats and --pmc counter passes) into small text/CSV files
for profiles/.   usage: prof_summary.py <rocprof_dir> [<out_prefix>]"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
lines = []
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    lines.append(f"# kernel stats: {os.path.relpath(f, d)}")
    lines += [l.rstrip() for l in open(f)]
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        key = (r["Kernel_Name"].split("(")[0][:60], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append(f"# counters (mean per dispatch; FETCH_SIZE/WRITE_SIZE in KB as reported): {os.path.relpath(f, d)}")
    lines.append("kernel,grid,wg,dispatches," + "counter=mean ...")
    for key, cs in sorted(agg.items()):
        n = max(len(v) for v in cs.values())
        lines.append(",".join(map(str, key)) + f",{n}," + " ".join(f"{k}={sum(v) / len(v):.4g}" for k, v in sorted(cs.items())))
txt = "\n".join(lines) + "\n"
if out:
    open(out, "w").write(txt)
else:
    sys.stdout.write(txt)
