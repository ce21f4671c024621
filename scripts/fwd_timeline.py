"""Forward-pass kernel list of the LAST overlapped step of a rocprofv3 --kernel-trace CSV: per kernel start, duration, gap
to the previous main-queue kernel.  usage: fwd_timeline.py DIR [min_us]"""
import csv, glob, sys, re
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
ev.sort()
ends = [e[1] for e in ev if "FusedAdam" in e[3]]
marks = [ends[0]]
for t in ends[1:]:
    if t - marks[-1] > 3e6: marks.append(t)
    else: marks[-1] = t
k = int(sys.argv[3]) if len(sys.argv) > 3 else -2
t0, t1 = marks[k - 1], marks[k]
step = [e for e in ev if t0 < e[0] <= t1]
qs = {}
for e in step: qs[e[2]] = qs.get(e[2], 0) + 1
mainq = max(qs, key=qs.get)
L = [e for e in step if e[2] == mainq]
stop = next((i for i, e in enumerate(L) if "head_bwd" in e[3] or "bce_dice_bwd" in e[3]), len(L))
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 0
short = lambda n: re.sub(r"^_Z\d+", "", re.sub(r"^void ", "", n))[:40]
prev = None
tot = gap = 0.0
for e in L[:stop]:
    d = (e[1] - e[0]) / 1e3
    g = (e[0] - prev) / 1e3 if prev else 0.0
    prev = e[1]
    tot += d; gap += max(g, 0)
    if d >= minus or g > 15:
        print(f"{(e[0]-t0)/1e3:9.1f} {d:8.1f} gap {g:6.1f}  {short(e[3])}")
print(f"forward: {len(L[:stop])} kernels, busy {tot/1e3:.2f} ms, gaps {gap/1e3:.2f} ms, span {(L[stop-1][1]-L[0][0])/1e6:.2f} ms")
