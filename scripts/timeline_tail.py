"""Dump the last `ms` milliseconds of the backward (before the first optimizer kernel) of the last step.
usage: timeline_tail.py DIR [ms]"""
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", "")))
ev.sort()
ends = [e[1] for e in ev if "FusedAdam" in e[3]]
marks = [ends[0]]
for t in ends[1:]:
    if t - marks[-1] > 3e6: marks.append(t)
    else: marks[-1] = t
t0, t1 = marks[-2], marks[-1]
step = [e for e in ev if t0 < e[0] <= t1]
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 2e6
first_opt = next(e[0] for e in step if "multi_tensor" in e[3] or "LpNorm" in e[3])
qs = sorted({e[2] for e in step})
for e in step:
    if first_opt - win <= e[0] <= first_opt + 0.3e6:
        name = e[3]
        for a, b in (("_Z", ""), ("IDF16b", ""), ("void ", "")):
            name = name.replace(a, b)
        print(f"{(e[0] - first_opt) / 1e3:9.1f} +{(e[1] - e[0]) / 1e3:7.1f} us  {'    ' * qs.index(e[2])}q{e[2]} {name[:50]} g={e[4]}x{e[5]}")
