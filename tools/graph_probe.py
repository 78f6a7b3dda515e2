#!/usr/bin/env python3
"""How much of the step is launch overhead?  Captures one train step (B=64) into a hipGraph and replays it (the replayed
steps reuse the captured dropout seed / Adam step scalars, so this is a timing probe only, not a training loop)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd.model import UNet  # noqa: E402


def main():
    dev = "cuda"
    B = 64
    model = UNet().to(dev).train()
    mix = torch.rand((B, 1, 512, 128), device=dev)
    voc = mix * 0.5
    for _ in range(5):
        model.train_step(mix, voc, 166.66)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        model.train_step(mix, voc, 166.66)
    torch.cuda.synchronize()
    print("eager  ms/step %.3f" % ((time.perf_counter() - t0) / 30 * 1e3))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model.train_step(mix, voc, 166.66)
        with torch.cuda.graph(g, stream=s):
            model.train_step(mix, voc, 166.66)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    print("graph  ms/step %.3f" % ((time.perf_counter() - t0) / 30 * 1e3))


if __name__ == "__main__":
    main()
