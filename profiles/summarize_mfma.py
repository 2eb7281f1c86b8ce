#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel from one rocprofv3 PMC pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES):
util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); both counters are summed over the chip by
rocprofv3 (MFMA busy over the 1024 SIMDs, GUI_ACTIVE over the 8 XCDs).  usage: summarize_mfma.py <pmc_dir> <out.json>"""
import collections, csv, glob, json, os, sys

d, out = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]); ga = sum(c["GRBM_GUI_ACTIVE"])
    if mf <= 0 or ga <= 0:
        continue
    rows.append({"kernel": k[:110], "launches": len(c["GRBM_GUI_ACTIVE"]),
                 "mfma_busy_cycles": mf, "gui_active_cycles_per_xcd": ga / 8,
                 "mfma_pipe_utilisation": round(mf / (ga / 8 * 1024), 4)})
rows.sort(key=lambda r: -r["mfma_busy_cycles"])
json.dump(rows, open(out, "w"), indent=1)
for r in rows[:12]:
    print(r["mfma_pipe_utilisation"], r["launches"], r["kernel"][:90])
