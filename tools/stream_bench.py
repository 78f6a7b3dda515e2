#!/usr/bin/env python3
"""End-to-end on-GPU separation rate (BASELINE configs[4], fp32 path): stereo waveform -> STFT -> normalise -> U-Net mask
on all tiles in one batch -> masked magnitude x mixture phase -> iSTFT -> peak-normalise, nothing leaving HBM.

    python tools/stream_bench.py [--seconds 240] [--rate 44100]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd import synth  # noqa: E402
from svs_unet_pytorch_amd.model import UNet  # noqa: E402
from svs_unet_pytorch_amd.streaming import separate_waveform  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16"))
    a = ap.parse_args()
    n = int(a.seconds * a.rate)
    model = UNet()
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state().items()})
    model.to("cuda").eval()
    y = torch.from_numpy(np.stack([synth.audio(n, 20), synth.audio(n, 21)])).to("cuda")
    for _ in range(2):
        out = separate_waveform(model, y, precision=a.precision)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        out = separate_waveform(model, y, precision=a.precision)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    frames = 1 + n // 768
    tiles = 2 * ((frames + 127) // 128)
    print(f"[{a.precision}] {a.seconds:.0f} s of stereo audio at {a.rate} Hz ({tiles} tiles): {dt * 1e3:.2f} ms end to end = "
          f"{a.seconds / dt:.0f}x real time, {tiles / dt:.0f} tiles/s incl. STFT / iSTFT; output {tuple(out.shape)}")


if __name__ == "__main__":
    main()
