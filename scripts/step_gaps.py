"""Where is the main HIP queue idle inside one train step?  From a rocprofv3 --kernel-trace CSV: the last complete step
(bounded by the optimizer kernels), split at the first loss-backward kernel; per phase the wall time, the main queue's kernel
time, its gaps (start - previous end) and the side queues' kernel time.  usage: step_gaps.py DIR_OR_CSV [step_index]"""
import csv, glob, sys, os, collections
src = sys.argv[1]
files = [src] if os.path.isfile(src) else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)
ev = []
for f in files:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
ev.sort()
ends = [e[1] for e in ev if "adamw_multi_kernel" in e[3] or "FusedAdam" in e[3] or "adamw_flat_kernel" in e[3]]
marks = [ends[0]]
for t in ends[1:]:
    if t - marks[-1] > 3e6: marks.append(t)
    else: marks[-1] = t
k = int(sys.argv[2]) if len(sys.argv) > 2 else -2
t0, t1 = marks[k - 1], marks[k]
step = [e for e in ev if t0 < e[0] <= t1]
qs = collections.Counter(e[2] for e in step)
mainq = qs.most_common(1)[0][0]
L = [e for e in step if e[2] == mainq]
i_bwd = next((i for i, e in enumerate(L) if "head_bwd" in e[3] or "bce_dice_bwd" in e[3] or "loss_bwd" in e[3]), len(L))
i_opt = next((i for i, e in enumerate(L) if "LpNorm" in e[3] or "adamw" in e[3]), len(L))
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels, queues {dict(qs)}")
def phase(name, evs, a, b):
    dur = sum(e[1] - e[0] for e in evs) / 1e3
    gaps = [max(evs[i][0] - evs[i - 1][1], 0) / 1e3 for i in range(1, len(evs))]
    h = collections.Counter("<2" if g < 2 else "2-5" if g < 5 else "5-10" if g < 10 else "10-30" if g < 30 else ">30" for g in gaps)
    side = sum(min(e[1], b) - max(e[0], a) for e in step if e[2] != mainq and e[1] > a and e[0] < b) / 1e3
    print(f"{name}: wall {(b - a) / 1e3:8.1f} us | main queue: {len(evs)} kernels, busy {dur:8.1f} us, gaps {sum(gaps):7.1f} us {dict(h)} | other queues busy {side:8.1f} us")
    big = sorted(((g, evs[i + 1][3][:50], evs[i][3][:50]) for i, g in enumerate(gaps)), reverse=True)[:6]
    for g, after, before in big:
        print(f"      gap {g:6.1f} us before {after}   (after {before})")
if i_bwd > 0:
    phase("forward ", L[:i_bwd], L[0][0], L[i_bwd][0] if i_bwd < len(L) else t1)
if i_bwd < i_opt:
    phase("backward", L[i_bwd:i_opt], L[i_bwd][0], L[i_opt][0] if i_opt < len(L) else t1)
if i_opt < len(L):
    phase("optimize", L[i_opt:], L[i_opt][0], t1)
def table(title, evs):
    b = collections.defaultdict(list)
    for e in evs:
        b[e[3].split("(")[0][:70]].append((e[1] - e[0]) / 1e3)
    print(title)
    for n, d in sorted(b.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {sum(d):8.1f} us  {len(d):3d} x {sum(d) / len(d):7.1f}  {n}")
if os.environ.get("STEP_GAPS_BWD"):
    table("backward kernels of this step (main queue):", L[i_bwd:i_opt])
    table("backward kernels of this step (other queues):", [e for e in step if e[2] != mainq and L[i_bwd][0] <= e[0] < (L[i_opt][0] if i_opt < len(L) else t1)])
if os.environ.get("STEP_GAPS_FWD"):
    b = collections.defaultdict(list)
    for e in L[:i_bwd]:
        b[e[3].split("(")[0][:70]].append((e[1] - e[0]) / 1e3)
    print("forward kernels of this step (main queue):")
    for n, d in sorted(b.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {sum(d):8.1f} us  {len(d):3d} x {sum(d) / len(d):7.1f}  {n}")
