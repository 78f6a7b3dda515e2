#!/usr/bin/env python3
"""Per-step kernel breakdown from a rocprofv3 --kernel-trace database (rocpd .db) of bench.py --mode train:
launches per step, wall vs. busy time, and per-kernel time per step over the last N steps (a step ends with
adam_kernel).

    python tools/prof_steps.py gpurun_out/prof/<...>_results.db [steps=8] > profiles/r01_train_b64_per_step.txt
"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = list(db.execute("select name,start,end from kernels order by start"))
    names = [re.sub(r"^void ", "", r[0]).split("(")[0] for r in rows]
    adam = [i for i, n in enumerate(names) if n.startswith("adam_kernel")]
    s0, s1 = adam[-nsteps - 1] + 1, adam[-1] + 1
    seg, segn = rows[s0:s1], names[s0:s1]
    wall = (seg[-1][2] - seg[0][1]) / nsteps
    busy = sum(r[2] - r[1] for r in seg) / nsteps
    print(f"steps {nsteps}  launches/step {len(seg) / nsteps:.1f}  wall {wall / 1e3:.1f} us/step  kernel-busy {busy / 1e3:.1f} us/step")
    acc = collections.defaultdict(lambda: [0, 0])
    for n, r in zip(segn, seg):
        acc[n][0] += r[2] - r[1]
        acc[n][1] += 1
    print(f"{'kernel':62s} {'calls':>6s} {'us/step':>9s} {'avg us':>8s} {'share':>6s}")
    for n, (t, k) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"{n[:62]:62s} {k / nsteps:6.1f} {t / nsteps / 1e3:9.1f} {t / k / 1e3:8.1f} {100 * t / nsteps / busy:5.1f}%")


if __name__ == "__main__":
    main()
