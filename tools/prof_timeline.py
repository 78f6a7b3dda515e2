#!/usr/bin/env python3
"""Two-stream timeline of ONE training step from a rocprofv3 --kernel-trace CSV: per queue the busy time, the time both queues
run kernels / only one does / none does, and every idle gap of a queue above a threshold with the kernels around it.

    python tools/prof_timeline.py <..._kernel_trace.csv> <launches per step> [step from the end = 2] [gap us = 4]
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
gap_us = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) - per * back: len(rows) - per * (back - 1)]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
name = lambda r: re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][:64]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
queues = {}
for r in rows:
    queues.setdefault(r[qkey] if qkey else "0", []).append(r)
print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(rows)} launches, queues: " + ", ".join(f"{q}: {len(v)} launches" for q, v in queues.items()))
# coverage: sweep over interval endpoints
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cover = {}
depth, prev = 0, t0
for t, d in ev:
    cover[min(depth, 3)] = cover.get(min(depth, 3), 0) + (t - prev)
    depth += d; prev = t
print("time with k kernels in flight: " + ", ".join(f"k={k}: {v / 1e3:.1f} us" for k, v in sorted(cover.items())))
for q, v in queues.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in v)
    print(f"\nqueue {q}: busy {busy / 1e3:.1f} us; idle gaps > {gap_us} us:")
    prev_r = None
    for r in v:
        if prev_r is not None:
            g = (int(r["Start_Timestamp"]) - int(prev_r["End_Timestamp"])) / 1e3
            if g > gap_us:
                print(f"  at {(int(prev_r['End_Timestamp']) - t0) / 1e3:8.1f} us  gap {g:7.1f}  after {name(prev_r)}  before {name(r)}")
        prev_r = r
if len(sys.argv) > 5:          # full listing
    for r in rows:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{r[qkey] if qkey else 0} {name(r)}")
