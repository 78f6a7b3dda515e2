#!/usr/bin/env python3
"""Repeats tests/test_gpu_parallel.py's two-rank-vs-emulation comparison (full objective) and reports which tensors differ.
    python tools/stress_2rank.py [iterations=8] [full=1]        (SVS_MFMA_SPLIT etc. from the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.multiprocessing as mp
import test_gpu_parallel as T
from svs_unet_pytorch_amd import _lib

def once(full):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = T._free_port()
    procs = [ctx.Process(target=T._worker, args=(r, 2, port, out, full)) for r in range(2)]
    for p in procs: p.start()
    got = {}
    for _ in range(2):
        rank, losses, flat, bn = out.get(timeout=600)
        got[rank] = (losses, flat, bn)
    for p in procs: p.join(timeout=120)
    models = [T._fresh_model() for _ in range(2)]
    shards = [T._shard(r) for r in range(2)]
    extras = [T._extras(r, full) for r in range(2)]
    for r, m in enumerate(models):
        m.rank = r; m.optim.grad_scale = 0.5
    for _ in range(T.STEPS):
        for r, m in enumerate(models):
            m.optim.zero_grad(); m.fwd_bwd(*shards[r], loss_scale=T.SCALE, **extras[r])
        total = models[0]._gflat + models[1]._gflat
        for m in models:
            m._gflat.copy_(total); m.optim.step()
    torch.cuda.synchronize()
    if os.environ.get("SVS_STRESS_EMU_TWICE"):                      # is the single-process emulation itself reproducible?
        again = [T._fresh_model() for _ in range(2)]
        for r, m in enumerate(again):
            m.rank = r; m.optim.grad_scale = 0.5
        for _ in range(T.STEPS):
            for r, m in enumerate(again):
                m.optim.zero_grad(); m.fwd_bwd(*shards[r], loss_scale=T.SCALE, **extras[r])
            total = again[0]._gflat + again[1]._gflat
            for m in again:
                m._gflat.copy_(total); m.optim.step()
        torch.cuda.synchronize()
        print("   emulation reproducible:", bool(torch.equal(again[0]._flat, models[0]._flat)), flush=True)
    offs = [int(_lib.lib().svs_unet_param_offset(i)) for i in range(47)]
    res = []
    for r in range(2):
        mine = models[r]._flat.cpu().numpy()
        bad = np.nonzero(got[r][1] != mine)[0]
        tens = sorted(set(int(np.searchsorted(offs, b, side="right") - 1) for b in bad))
        res.append((bad.size, tens[:12], float(np.abs(got[r][1] - mine).max()) if bad.size else 0.0))
    res.append(("ranks equal", bool(np.array_equal(got[0][1], got[1][1]))))
    return res

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    full = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    for i in range(n):
        print(i, once(full), flush=True)
