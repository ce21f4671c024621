"""Summarise a rocprofv3 --pmc (SQ counters) CSV per kernel: wave-cycle breakdown and MFMA / LDS utilisation.
usage: pmc_sq.py DIR"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:44], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    n = max(cnt[k], 1)
    gui = c.get("GRBM_GUI_ACTIVE", 0) / n
    line = f"{k[0]:44s} g={k[1]:>9s} n={n:3d} wave_cyc/launch={wc / n:.3e}"
    for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
        if name in c: line += f" {name[3:]}={c[name] / wc * 100:.0f}%"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and gui > 0:
        # busy cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs -> cycles/XCD = gui/8; 1024 SIMDs
        line += f" mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES'] / n / (gui / 8 * 1024) * 100:.0f}%"
    if "SQ_LDS_IDX_ACTIVE" in c and gui > 0:
        line += f" lds_active={c['SQ_LDS_IDX_ACTIVE'] / n / (gui / 8 * 256) * 100:.0f}% conflict={c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c['SQ_LDS_IDX_ACTIVE'], 1) * 100:.0f}%"
    print(line)
