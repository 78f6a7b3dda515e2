#!/usr/bin/env python3
"""In-order view of the last launches of a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous kernel's
end, grid and name -- for latency-bound sequences such as the batch-16 eval forward.

    python tools/prof_sequence.py <..._kernel_trace.csv> <launches per pass> [passes=1]
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-per * passes:]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    grid = "x".join(str(int(r[k]) // max(int(r[w]), 1)) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")))
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  grid {grid:>12s}  {nm[:70]}")
    prev = e
    busy += e - s
print(f"span {(prev - t0) / 1e3:.1f} us over {passes} pass(es), kernel-busy {busy / 1e3:.1f} us")
