"""Print the top of a rocprofv3 kernel_stats CSV: share, calls, avg us, total ms per step.  usage: kstats.py CSV steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
t = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {t / steps / 1e6:.2f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 16]:
    print(f"{float(r['TotalDurationNs']) / t * 100:5.1f}% calls={r['Calls']:>5} avg={float(r['AverageNs']) / 1e3:7.1f}us per-step={float(r['TotalDurationNs']) / steps / 1e6:6.2f}ms {r['Name'][:64]}")
