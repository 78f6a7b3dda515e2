#!/usr/bin/env python3
"""Experiment: the batch-16 eval forward as two independent half-batches on two streams (tiles are independent in eval mode),
against one launch chain for the whole batch.  Same device, same process.   python tools/eval_two_streams.py [--batch 16]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svs_unet_pytorch_amd import _lib, synth  # noqa: E402
from svs_unet_pytorch_amd.model import UNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--parts", type=int, default=2)
args = ap.parse_args()
B = args.batch
L = _lib.lib()
model = UNet()
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=True).items()})
model.to("cuda").eval()
mix = torch.empty((B, 1, 512, 128), device="cuda")
voc = torch.empty_like(mix)
_lib.check(L.svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), B, 512, 128, 0, _lib.stream_ptr()))
with torch.no_grad():
    ref = model(mix).clone()
prepared = model._prepared
P = args.parts
hb = B // P
streams = [torch.cuda.Stream() for _ in range(P)]
wss = [torch.empty(int(L.svs_unet_eval_workspace_bytes(hb, 512, 128)), dtype=torch.uint8, device="cuda") for _ in range(P)]
ws_full = torch.empty(int(L.svs_unet_eval_workspace_bytes(B, 512, 128)), dtype=torch.uint8, device="cuda")
out = torch.empty_like(mix)
main = torch.cuda.current_stream()
fork = torch.cuda.Event()
joins = [torch.cuda.Event() for _ in range(P)]


def whole():
    _lib.check(L.svs_unet_forward_eval(prepared.data_ptr(), mix.data_ptr(), out.data_ptr(), B, 512, 128, ws_full.data_ptr(), ws_full.numel(), main.cuda_stream))


def halves():
    fork.record(main)
    for i, s in enumerate(streams):
        s.wait_event(fork)
        _lib.check(L.svs_unet_forward_eval(prepared.data_ptr(), mix[i * hb:].data_ptr(), out[i * hb:].data_ptr(), hb, 512, 128, wss[i].data_ptr(), wss[i].numel(),
                                           s.cuda_stream))
        joins[i].record(s)
        main.wait_event(joins[i])


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for rnd in range(3):
    tw = timed(whole)
    whole(); torch.cuda.synchronize(); dw = (out - ref).abs().max().item()
    th = timed(halves)
    halves(); torch.cuda.synchronize(); dh = (out - ref).abs().max().item()
    print(f"round {rnd}: whole batch {tw:.4f} ms (max diff {dw:.1e})   {P} x {hb} tiles on {P} streams {th:.4f} ms (max diff {dh:.1e})")
