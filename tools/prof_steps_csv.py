#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py --mode train: launches per step, wall vs. busy time
and per-kernel time per step over the last N steps (a step ends with adam_kernel).

    python tools/prof_steps_csv.py <..._kernel_trace.csv> [steps=8] > profiles/r02_train_b64_per_step.txt
"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0] for r in rows]
adam = [i for i, nm in enumerate(names) if nm.startswith("adam_kernel")]
s0, s1 = adam[-n - 1] + 1, adam[-1] + 1
acc = collections.defaultdict(lambda: [0, 0])
for r, nm in zip(rows[s0:s1], names[s0:s1]):
    acc[nm][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc[nm][1] += 1
wall = (int(rows[s1 - 1]["End_Timestamp"]) - int(rows[s0]["Start_Timestamp"])) / n
busy = sum(v[0] for v in acc.values()) / n
print(f"steps {n}  launches/step {(s1 - s0) / n:.1f}  wall {wall / 1e3:.1f} us/step  kernel-busy {busy / 1e3:.1f} us/step")
print(f"{'kernel':62s} {'calls':>6s} {'us/step':>9s} {'avg us':>8s} {'share':>6s}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k[:62]:62s} {v[1] / n:6.1f} {v[0] / n / 1e3:9.1f} {v[0] / v[1] / 1e3:8.1f} {100 * v[0] / n / busy:5.1f}%")
