"""Timeline of ONE train step from a rocprofv3 --kernel-trace CSV: every main-queue kernel with its start offset, duration and the
side-queue kernels that overlap it (how much of the main chain's time is spent next to which weight-gradient kernel).
usage: step_timeline.py DIR_OR_CSV [step_index] > timeline.txt"""
import csv, glob, sys, os, collections, re
src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
ev = []
for f in files:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
ev.sort()
ends = [e[1] for e in ev if "adamw_multi_kernel" in e[3]]
marks = [ends[0]]
for t in ends[1:]:
    if t - marks[-1] > 3e6: marks.append(t)
    else: marks[-1] = t
k = int(sys.argv[2]) if len(sys.argv) > 2 else -2
t0, t1 = marks[k - 1], marks[k]
step = [e for e in ev if t0 < e[0] <= t1]
qs = collections.Counter(e[2] for e in step)
mainq = qs.most_common(1)[0][0]
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    n = n.replace("bool _Accum", "bf16")
    m = re.match(r"_Z\d+([A-Za-z_0-9]+?)I", n)
    return (m.group(1) if m else n)[:44]
side = [e for e in step if e[2] != mainq]
print(f"step wall {(t1 - t0) / 1e6:.3f} ms; main queue {mainq}: {qs[mainq]} kernels; other queues: {sum(v for q, v in qs.items() if q != mainq)}")
prev_end = None
tot_gap = 0.0
for e in step:
    if e[2] != mainq:
        continue
    ov = collections.defaultdict(float)
    for s in side:
        o = min(e[1], s[1]) - max(e[0], s[0])
        if o > 0:
            ov[short(s[3])] += o / 1e3
    gap = (e[0] - prev_end) / 1e3 if prev_end is not None else 0.0
    tot_gap += max(gap, 0)
    prev_end = e[1]
    o = " ".join(f"{n}:{v:.0f}" for n, v in sorted(ov.items(), key=lambda kv: -kv[1]))
    print(f"{(e[0] - t0) / 1e3:9.1f} +{gap:6.1f} {(e[1] - e[0]) / 1e3:8.1f} us  {short(e[3]):44s} | {o}")
print("main-queue gaps total", round(tot_gap, 1), "us")
