"""Summarise a rocprofv3 --kernel-trace CSV by (kernel, grid): calls, avg/min us.  usage: trace_summary.py DIR [filter]"""
import csv, glob, sys, collections
rows = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        grid = (r.get("Grid_Size_X") or r.get("Grid_Size", "?"), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        rows[(name[:60], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v):10.1f} us total  n={len(v):4d} avg={sum(v)/len(v):8.1f} min={min(v):8.1f}  grid={'x'.join(map(str, grid))}  {name}")
