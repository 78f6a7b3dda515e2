#!/usr/bin/env python3
"""Same-device, same-process A/B of one tuning switch on the whole train step (devices differ by several %, so two gpurun
calls cannot be compared):  python tools/ab_tune.py TRAIN_ONE_STREAM 1 [-1] [--batch 64] [--rounds 4] [--steps 30]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svs_unet_pytorch_amd import _lib, synth  # noqa: E402
from svs_unet_pytorch_amd.model import ALPHA_L1, UNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("a", type=int)
ap.add_argument("b", type=int, nargs="?", default=-1)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--eval", action="store_true", help="time the eval forward instead of the train step")
args = ap.parse_args()
B = args.batch
model = UNet()
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=False).items()})
model.to("cuda").train()
if args.eval:
    model.eval()
mix = torch.empty((B, 1, 512, 128), device="cuda")
voc = torch.empty_like(mix)
_lib.check(_lib.lib().svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), B, 512, 128, 0, _lib.stream_ptr()))
def one():
    if args.eval:
        with torch.no_grad():
            model(mix)
    else:
        model.train_step(mix, voc, loss_scale=ALPHA_L1)


res = {args.a: [], args.b: []}
for r in range(args.rounds):
    for val in (args.a, args.b):
        _lib.tuning(args.name, val)
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one()
        torch.cuda.synchronize()
        res[val].append(1e3 * (time.perf_counter() - t0) / args.steps)
for val, v in res.items():
    print(f"{args.name}={val}: median {np.median(v):.4f} ms  min {min(v):.4f}  all {[round(x, 4) for x in v]}")
