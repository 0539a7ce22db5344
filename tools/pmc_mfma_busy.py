#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
busy % = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.
usage: python tools/pmc_mfma_busy.py <dir> <out.txt>"""
import collections, csv, glob, sys
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
rows = []
for k, c in d.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    rows.append((gui, k, n[k], 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * gui)))
tot = sum(r[0] for r in rows)
with open(sys.argv[2], "w") as f:
    f.write("kernel | launches | share of GPU-active cycles | matrix pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles)\n")
    for gui, k, cnt, busy in sorted(rows, reverse=True)[:24]:
        f.write(f"{k[:110]:110s} | {cnt:6d} | {100 * gui / tot:5.1f} % | {busy:5.1f} %\n")
print(open(sys.argv[2]).read())
