#!/usr/bin/env python3
"""MFMA-busy fraction per kernel from a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE:

    busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * GRBM_GUI_ACTIVE / 8 * 256)

GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back), so GRBM_GUI_ACTIVE / 8 is the kernel's
duration in shader cycles; SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip and counts, per CU, the cycles in which any of
its four matrix pipes is busy ... (calibrated below against the kernel's known MFMA count).

    python tools/pmc_mfma.py <dir of the pass> [commit] > profiles/r02_pmc_mfma_busy.json
"""
import collections
import csv
import glob
import json
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", row["Kernel_Name"]).split("(")[0]
        a = acc[name][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
out = {"_commit": sys.argv[2] if len(sys.argv) > 2 else "",
       "_formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs); gpu_cycles_per_launch = GRBM_GUI_ACTIVE / 8"}
for k, d in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    busy = d["SQ_VALU_MFMA_BUSY_CYCLES"][0] / d["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    gui = d["GRBM_GUI_ACTIVE"][0] / d["GRBM_GUI_ACTIVE"][1] / 8.0
    if busy <= 0:
        continue
    out[k] = {"launches": d["GRBM_GUI_ACTIVE"][1], "mfma_busy_cycles_per_launch": round(busy), "gpu_cycles_per_launch": round(gui),
              "mfma_busy_frac": round(busy / (gui * 1024), 4)}
print(json.dumps(out, indent=1))
