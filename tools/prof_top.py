#!/usr/bin/env python3
"""Prints the top kernels of a rocprofv3 --stats CSV:  python tools/prof_top.py <..._kernel_stats.csv> [n]"""
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for r in rows[1:1 + n]:
    print(f"{r[0][:66]:66s} calls {r[1]:>5s}  avg {float(r[3]) / 1e3:9.1f} us  {r[4]:>6s} %")
