#!/usr/bin/env python3
"""One table per kernel from the rocprofv3 passes of tools/pmc_bf16.sh (a directory of passes, each with a *counter_collection.csv,
plus kt/ with the kernel trace): average duration, HBM bytes per launch (FETCH_SIZE doubled: the gfx950 correction of
MI355X_MICROARCH.md for wide streaming reads; WRITE_SIZE as is), MFMA-busy fraction, LDS bank-conflict share, issue / wait shares.

    python tools/pmc_table.py <dir> [--json out.json]
"""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
name_of = lambda r: re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = cnt[name_of(row)][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(root + "/kt/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[len(rows) // 4:]:                      # skip the first passes (cold caches, first-use set-up)
        d = dur[name_of(r)]
        d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
avg = lambda d, k: (d[k][0] / d[k][1]) if k in d and d[k][1] else None
out = {}
print(f"{'kernel':58s} {'us':>7s} {'HBM MB':>8s} {'TB/s':>6s} {'mfma':>6s} {'ldsconf':>7s} {'valu':>6s} {'waitinst':>8s} {'waitany':>7s}")
for k in sorted(dur, key=lambda k: -dur[k][0]):
    us = dur[k][0] / dur[k][1]
    c = cnt.get(k, {})
    fetch, write = avg(c, "FETCH_SIZE"), avg(c, "WRITE_SIZE")
    mb = None if fetch is None or write is None else (2 * fetch + write) * 1024 / 1e6        # counters are in KB; MB / us = TB/s
    busy, gui = avg(c, "SQ_VALU_MFMA_BUSY_CYCLES"), avg(c, "GRBM_GUI_ACTIVE")
    mf = None if not busy or not gui else busy / (gui / 8 * 1024)
    conf, idx = avg(c, "SQ_LDS_BANK_CONFLICT"), avg(c, "SQ_LDS_IDX_ACTIVE")
    lc = None if not idx else conf / idx
    wc = avg(c, "SQ_WAVE_CYCLES")
    valu = None if not wc else avg(c, "SQ_ACTIVE_INST_VALU") / wc
    wi = None if not wc else avg(c, "SQ_WAIT_INST_ANY") / wc
    wa = None if not wc or avg(c, "SQ_WAIT_ANY") is None else avg(c, "SQ_WAIT_ANY") / wc
    f = lambda v, w, p: (f"{v:{w}.{p}f}" if v is not None else " " * (w - 1) + "-")
    print(f"{k[:58]:58s} {us:7.1f} {f(mb, 8, 1)} {f(None if mb is None else mb / us, 6, 2)} {f(mf, 6, 3)} {f(lc, 7, 3)} {f(valu, 6, 3)} {f(wi, 8, 3)} {f(wa, 7, 3)}   x{dur[k][1]}")
    out[k] = {"launches": dur[k][1], "avg_us": round(us, 2), "hbm_MB_per_launch": None if mb is None else round(mb, 2),
              "mfma_busy_frac": None if mf is None else round(mf, 4), "lds_conflict_share": None if lc is None else round(lc, 4),
              "valu_active_share": None if valu is None else round(valu, 4), "wait_inst_share": None if wi is None else round(wi, 4)}
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
