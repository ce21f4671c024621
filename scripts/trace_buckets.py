"""Bucket a rocprofv3 kernel trace (csv) by kernel name and grid size: per-step launches, mean duration, total per step.
Usage: python scripts/trace_buckets.py <kernel_trace.csv> <steps> [name-substring ...]"""
import csv, sys, collections
path, steps = sys.argv[1], int(sys.argv[2])
subs = sys.argv[3:]
b = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if subs and not any(s in n for s in subs):
        continue
    short = n.split("(")[0][:60]
    b[(short, "x".join(str(int(r.get("Grid_Size_" + a, 1)) // max(int(r.get("Workgroup_Size_" + a, 1)), 1)) for a in "XYZ"), r.get("Workgroup_Size_X", "?"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(b.items(), key=lambda kv: -sum(kv[1]))
tot = 0
for (n, g, wg), d in rows[:60]:
    tot += sum(d)
    print(f"{sum(d)/steps/1e3:8.1f} us/step  {len(d)/steps:5.1f} x {sum(d)/len(d)/1e3:7.1f} us  grid {g:>13} wg {wg:>4}  {n}")
print(f"total {tot/steps/1e3:.1f} us/step")
