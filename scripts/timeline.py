"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py: per queue busy time, main-queue gaps, and time per
kernel family inside the LAST step.  usage: timeline.py DIR"""
import csv, glob, sys, collections
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"], r.get("Grid_Size_X", "")))
ev.sort()
# the last optimizer kernel marks step ends
ends = [e[1] for e in ev if "FusedAdam" in e[3]]
if len(ends) < 2:
    sys.exit("no steps found")
# group adam launches into steps (gaps > 2 ms)
marks = [ends[0]]
for t in ends[1:]:
    if t - marks[-1] > 3e6: marks.append(t)
    else: marks[-1] = t
t0, t1 = marks[-2], marks[-1]
step = [e for e in ev if t0 < e[0] <= t1]
print(f"last step: {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels")
byq = collections.defaultdict(list)
for e in step: byq[e[2]].append(e)
for q, L in byq.items():
    busy = sum(e[1] - e[0] for e in L)
    print(f"queue {q}: {len(L)} kernels, busy {busy / 1e6:.2f} ms")
mainq = max(byq, key=lambda q: len(byq[q]))
L = byq[mainq]
gaps = [(L[i + 1][0] - L[i][1], L[i][3][:50], L[i + 1][3][:50]) for i in range(len(L) - 1)]
tot_gap = sum(g[0] for g in gaps if g[0] > 0)
print(f"main queue {mainq}: total idle gaps {tot_gap / 1e6:.2f} ms; gaps > 20 us:")
for g in sorted(gaps, reverse=True)[:15]:
    print(f"   {g[0] / 1e3:8.1f} us  after {g[1]}  before {g[2]}")
# small-gap histogram
small = [g[0] for g in gaps if 0 < g[0] <= 20000]
print(f"   {len(small)} gaps <= 20 us, sum {sum(small) / 1e6:.2f} ms, mean {sum(small) / max(len(small), 1) / 1e3:.1f} us")
fam = collections.defaultdict(lambda: [0, 0.0])
for e in L:
    k = e[3].split("(")[0][:48]
    fam[k][0] += 1; fam[k][1] += (e[1] - e[0]) / 1e3
print("main queue kernel families:")
for k, (n, us) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"   {us / 1e3:7.2f} ms n={n:4d} avg={us / n:7.1f} us  {k}")
