#!/usr/bin/env python3
"""Instruction histogram of one kernel of libsvs_hip.so (static count by class):  python tools/isa_hist.py mr_pass_kernel  [lib]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import LLVM, device_code_objects  # noqa: E402

pat = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "svs_unet_pytorch_amd", "libsvs_hip.so")
with tempfile.TemporaryDirectory() as wd:
    for co in device_code_objects(lib, wd):
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", "--demangle", co], capture_output=True, text=True, check=True).stdout
        sym, hist = None, None
        def flush():
            if sym and pat in sym and hist:
                tot = sum(hist.values())
                cls = collections.Counter()
                for k, v in hist.items():
                    c = "valu" if k.startswith("v_") else "ds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if k.startswith("s_") else "other"
                    if k.startswith("v_mfma"): c = "mfma"
                    if k in ("s_waitcnt", "s_nop", "s_barrier"): c = k
                    cls[c] += v
                print(f"{sym[:110]}\n  total {tot}  " + "  ".join(f"{k} {v}" for k, v in cls.most_common()))
                print("  top: " + "  ".join(f"{k} {v}" for k, v in hist.most_common(18)))
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                flush()
                sym, hist = m.group(1), collections.Counter()
                continue
            m = re.match(r"^\s+([a-z_0-9]+)\s", line)
            if m and hist is not None:
                hist[m.group(1)] += 1
        flush()
