#!/usr/bin/env python3
"""A/B two builds of libsvs_hip.so on the single-channel kernels (deconv6 forward = svs_out_block_fwd), interleaved
in one process on one device.

    python tools/ab_c1.py tools/bin/libsvs_hip_base.so svs_unet_pytorch_amd/libsvs_hip.so [--batch 64]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd import _lib  # noqa: E402
from tools.ab_libs import timeit  # noqa: E402


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for name in ("svs_out_block_fwd",):
        fn = getattr(h, name)
        fn.restype, fn.argtypes = _lib._SIGS[name]
    return h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib_a")
    ap.add_argument("lib_b")
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    torch.zeros(1, device="cuda")
    libs = [load(a.lib_a), load(a.lib_b)]
    B, H, W, C = a.batch, 256, 64, 32
    x = torch.rand((B, H, W, C), device="cuda") - 0.5
    w = (torch.rand((C, 1, 5, 5), device="cuda") - 0.5) * 0.2
    b = torch.rand(1, device="cuda")
    ys = [torch.empty((B, 2 * H, 2 * W), device="cuda") for _ in libs]
    S = _lib.stream_ptr
    runs = [lambda L=L, y=y: L.svs_out_block_fwd(x.data_ptr(), C, B, H, W, C, w.data_ptr(), b.data_ptr(), y.data_ptr(), 2 * H, 2 * W, 1, S())
            for L, y in zip(libs, ys)]
    for r in runs:
        assert r() == 0
    torch.cuda.synchronize()
    best = [1e9, 1e9]
    for _ in range(5):
        for i, r in enumerate(runs):
            best[i] = min(best[i], timeit(r, 20))
    mb = (x.numel() + ys[0].numel()) * 4 / 1e6
    print(f"deconv6.fwd  A {best[0] * 1e3:7.1f} us ({mb / best[0] / 1e3:.2f} TB/s)   B {best[1] * 1e3:7.1f} us ({mb / best[1] / 1e3:.2f} TB/s)"
          f"   maxdiff {(ys[0] - ys[1]).abs().max().item():.1e}")


if __name__ == "__main__":
    main()
