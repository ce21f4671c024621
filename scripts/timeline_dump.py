"""Dump a window of the last step's kernel timeline (both queues).  usage: timeline_dump.py DIR start_kernel_substr [ms]"""
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
key = sys.argv[2]
win = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 1.5e6
s0 = next(e[0] for e in step if key in e[3])
qs = sorted({e[2] for e in step})
for e in step:
    if s0 - 50e3 <= e[0] <= s0 + win:
        name = e[3]
        for a, b in (("_Z", ""), ("IDF16b", ""), ("void ", "")):
            name = name.replace(a, b)
        col = qs.index(e[2])
        print(f"{(e[0] - s0) / 1e3:9.1f} +{(e[1] - e[0]) / 1e3:7.1f} us  {'    ' * col}q{e[2]} {name[:46]} g={e[4]}x{e[5]}")
