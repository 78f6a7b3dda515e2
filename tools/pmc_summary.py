#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from the counter_collection CSV(s) under a directory.
    python tools/pmc_summary.py gpurun_out/pmc_dir [name-filter]"""
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
        if len(sys.argv) > 2 and sys.argv[2] not in name:
            continue
        a = acc[name][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: round(v[0] / v[1], 1) for c, v in d.items()} | {"launches": max(v[1] for v in d.values())} for k, d in acc.items()}
print(json.dumps(out, indent=1))
