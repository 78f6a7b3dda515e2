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
    for name in ("svs_out_block_fwd", "svs_enc_block_fwd", "svs_dec_block_bwd_data", "svs_enc_block_bwd_weight", "svs_dec_block_bwd_weight", "svs_block_bwd_weight_workspace_bytes"):
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
    # the single-channel weight gradients: conv1 (S = dy 16 ch, L = mix) and deconv6 (S = x 32 ch, L = d_logit)
    for tag, cs in (("conv1.bwd_weight", 16), ("deconv6.bwd_weight", 32)):
        sm = torch.rand((B, H, W, cs), device="cuda") - 0.5
        lg = torch.rand((B, 2 * H, 2 * W), device="cuda") - 0.5
        dws = [torch.empty(cs * 25, device="cuda") for _ in libs]
        wss = [torch.empty(int(L.svs_block_bwd_weight_workspace_bytes(B, H, W, cs, 1)) + 256, dtype=torch.uint8, device="cuda") for L in libs]
        if cs == 16:
            rs = [lambda L=L, dw=dw, ws=ws: L.svs_enc_block_bwd_weight(sm.data_ptr(), cs, B, H, W, cs, lg.data_ptr(), 1, 2 * H, 2 * W, 1, dw.data_ptr(),
                                                                       None, ws.data_ptr(), ws.numel(), S()) for L, dw, ws in zip(libs, dws, wss)]
        else:
            rs = [lambda L=L, dw=dw, ws=ws: L.svs_dec_block_bwd_weight(sm.data_ptr(), cs, B, H, W, cs, lg.data_ptr(), 1, 2 * H, 2 * W, 1, dw.data_ptr(),
                                                                       None, ws.data_ptr(), ws.numel(), S()) for L, dw, ws in zip(libs, dws, wss)]
        for r in rs:
            assert r() == 0
        torch.cuda.synchronize()
        bb = [1e9, 1e9]
        for _ in range(5):
            for i, r in enumerate(rs):
                bb[i] = min(bb[i], timeit(r, 20))
        rel = ((dws[0] - dws[1]).abs().max() / dws[0].abs().max()).item()
        print(f"{tag:20s} A {bb[0] * 1e3:7.1f} us   B {bb[1] * 1e3:7.1f} us   reldiff {rel:.1e}")
    # the single-channel convolutions: conv1 forward (1 -> 16) and deconv6 backward-data (1 -> 32)
    img = torch.rand((B, 2 * H, 2 * W), device="cuda")
    for tag, n in (("conv1.fwd", 16), ("deconv6.bwd_data", 32)):
        wt = (torch.rand((n, 25), device="cuda") - 0.5) * 0.2
        outs = [torch.empty((B, H, W, n), device="cuda") for _ in libs]
        if n == 16:
            rs = [lambda L=L, o=o: L.svs_enc_block_fwd(img.data_ptr(), 1, B, 2 * H, 2 * W, 1, wt.data_ptr(), None, None, None, 0.0, o.data_ptr(),
                                                       n, n, 0, None, 0, S()) for L, o in zip(libs, outs)]
        else:
            rs = [lambda L=L, o=o: L.svs_dec_block_bwd_data(img.data_ptr(), 1, B, 2 * H, 2 * W, 1, wt.data_ptr(), o.data_ptr(), n, H, W, n, 0,
                                                            None, 0, S()) for L, o in zip(libs, outs)]
        for r in rs:
            assert r() == 0
        torch.cuda.synchronize()
        bb = [1e9, 1e9]
        for _ in range(5):
            for i, r in enumerate(rs):
                bb[i] = min(bb[i], timeit(r, 20))
        rel = ((outs[0] - outs[1]).abs().max() / outs[0].abs().max()).item()
        print(f"{tag:20s} A {bb[0] * 1e3:7.1f} us   B {bb[1] * 1e3:7.1f} us   reldiff {rel:.1e}")
    print(f"deconv6.fwd  A {best[0] * 1e3:7.1f} us ({mb / best[0] / 1e3:.2f} TB/s)   B {best[1] * 1e3:7.1f} us ({mb / best[1] / 1e3:.2f} TB/s)"
          f"   maxdiff {(ys[0] - ys[1]).abs().max().item():.1e}")


if __name__ == "__main__":
    main()
