#!/usr/bin/env python3
"""Bandwidth of the signal kernels (csrc/stft.hip, csrc/mrstft.hip), HIP events on the launch stream.

    python tools/signal_bench.py [--seconds 240] [--rate 44100] [--batch 64]

STFT: stereo waveform -> network tiles + frame-major phasors (one launch).  iSTFT: tiles x mask x phasors -> waveform (one
launch).  Algorithmic bytes: samples in/out (4 B), magnitude tiles (4 B / bin, DC row dropped), phasors (8 B / bin),
mask (4 B / bin) -- each counted once.  MR-STFT: the training loss on B waveforms of 97,536 samples.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd import _lib, synth  # noqa: E402
from svs_unet_pytorch_amd.data import istft_from_tiles, specific_istft, stft_to_tiles  # noqa: E402

HBM_ACHIEVABLE_TBS = 6.29          # /opt/skills/guides/MI355X_MICROARCH.md (float4 copy)


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def timed_isolated(fn, reps=10):
    """One launch at a time on an idle device (synchronise, event, launch, event): start-to-end of a single launch plus the
    event overhead -- the upper companion of `timed`, whose back-to-back launches overlap each other's ramp-up and tail."""
    fn()
    tot = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps


def signal_record(seconds=240.0, rate=44100, batch=64):
    n = int(seconds * rate)
    L = _lib.lib()
    y = torch.from_numpy(np.stack([synth.audio(n, 20), synth.audio(n, 21)])).to("cuda")
    tiles, phase, peak, T = stft_to_tiles(y)
    mask = torch.rand_like(tiles)
    C, n_tiles = tiles.shape[:2]
    # raw launches (no per-channel max / scale around them)
    ph = torch.view_as_real(phase).contiguous()
    part = torch.empty((C, int(L.svs_stft_groups(n_tiles * 128))), device="cuda")
    fwd = lambda: L.svs_stft_tiles(y.data_ptr(), n, C, 1024, 768, tiles.data_ptr(), n_tiles * 512 * 128, 128, 512, 1, n_tiles * 128,
                                   ph.data_ptr(), 1, part.data_ptr(), _lib.stream_ptr())
    out = torch.empty((C, 768 * (T - 1)), device="cuda")
    part2 = torch.empty((C, int(L.svs_istft_groups(768, T, C))), device="cuda")
    inv = lambda: L.svs_istft_tiles(tiles.data_ptr(), n_tiles * 512 * 128, 128, 512, 1, mask.data_ptr(), 0, ph.data_ptr(), 1, C, 1024, 768, T,
                                    out.data_ptr(), part2.data_ptr(), _lib.stream_ptr())
    ms_f, ms_i = timed(fwd), timed(inv)
    iso_f, iso_i = timed_isolated(fwd), timed_isolated(inv)
    bytes_f = C * (n * 4 + T * 512 * 4 + T * 513 * 8)
    bytes_i = C * (T * 512 * 4 * 2 + T * 513 * 8 + 768 * (T - 1) * 4)
    rec = {"audio_seconds": seconds, "channels": C, "frames_per_channel": T,
           "stft": {"ms": round(ms_f, 4), "algorithmic_MB": round(bytes_f / 1e6, 1), "GBps": round(bytes_f / ms_f / 1e6, 1),
                    "frac_of_6.29TBps": round(bytes_f / ms_f / 1e9 / HBM_ACHIEVABLE_TBS, 3)},
           "istft": {"ms": round(ms_i, 4), "algorithmic_MB": round(bytes_i / 1e6, 1), "GBps": round(bytes_i / ms_i / 1e6, 1),
                     "frac_of_6.29TBps": round(bytes_i / ms_i / 1e9 / HBM_ACHIEVABLE_TBS, 3)},
           "timing": "ms = HIP events around 20 back-to-back launches on the launch stream (what a pipeline of launches sustains: successive "
                     "launches overlap each other's ramp-up and tail); ms_single_launch = one launch at a time on an idle device, start to end "
                     "incl. the event pair (what rocprofv3's per-kernel average of profiles/r0N_signal_*_kernel_stats.csv corresponds to)"}
    rec["stft"]["ms_single_launch"] = round(iso_f, 4)
    rec["istft"]["ms_single_launch"] = round(iso_i, 4)
    # training-side pieces at batch B: specific_istft (train.py:33-60) and the MR-STFT loss with gradient (train.py:293)
    B, Tt = batch, 128
    mag = torch.rand((B, 1, 512, Tt), device="cuda")
    ang = (torch.rand((B, 1, 512, Tt), device="cuda") - 0.5) * 6.28
    rec["specific_istft_ms"] = round(timed(lambda: specific_istft(mag, ang)), 4)
    Lw = 768 * (Tt - 1)
    x = (torch.rand((B, Lw), device="cuda") - 0.5) * 0.4
    yy = x * 0.7 + (torch.rand((B, Lw), device="cuda") - 0.5) * 0.2
    ws = torch.empty(int(L.svs_mrstft_workspace_bytes(B, Lw)), dtype=torch.uint8, device="cuda")
    loss, dx = torch.zeros(1, device="cuda"), torch.empty_like(x)
    rec["mrstft_fwd_bwd_ms"] = round(timed(lambda: L.svs_mrstft_loss_fwd_bwd(x.data_ptr(), yy.data_ptr(), B, Lw, 1.0, loss.data_ptr(), dx.data_ptr(),
                                                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr()), reps=5), 4)
    rec["mrstft_value_only_ms"] = round(timed(lambda: L.svs_mrstft_loss_fwd_bwd(x.data_ptr(), yy.data_ptr(), B, Lw, 1.0, loss.data_ptr(), None,
                                                                               ws.data_ptr(), ws.numel(), _lib.stream_ptr()), reps=5), 4)
    rec["mrstft_batch"] = B
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--rate", type=int, default=44100)
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    print(json.dumps(signal_record(a.seconds, a.rate, a.batch)))
