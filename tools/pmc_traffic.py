#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes over bench.py (one with FETCH_SIZE, one with WRITE_SIZE: they do not fit one
pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots') into per-kernel HBM bytes per launch:

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024

FETCH_SIZE is doubled: on gfx950 it reports half the bytes of wide (16 B/lane) streaming reads
(MI355X_MICROARCH.md 'HBM').  Writes profiles/r02_pmc_traffic.json, which bench.py reads for `roofline.traffic`.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [out.json]
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"fetch_size_kb_per_launch": round(f, 1), "write_size_kb_per_launch": round(w, 1),
                  "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
    path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "profiles", "r02_pmc_traffic.json")
    if len(sys.argv) > 4:
        out["_commit"] = sys.argv[4]            # the commit the counters were taken at (bench.py quotes it with `traffic`)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, dict)), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k[:70]:70s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
