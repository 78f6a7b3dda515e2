#!/usr/bin/env python3
"""Runs ONE GEMM call of the training step repeatedly (for rocprofv3 --pmc / --kernel-trace runs).
    python tools/one_layer.py conv4.fwd [--batch 64] [--reps 20]
Names: convK.fwd|bwd_data|bwd_weight (K=2..6), deconvJ.fwd|bwd_data|bwd_weight (J=1..5)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd import _lib  # noqa: E402

CH = (1, 16, 32, 64, 128, 256, 512)
DEC = ((512, 256), (512, 128), (256, 64), (128, 32), (64, 16))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    hw = [(512, 128)]
    for _ in range(6):
        hw.append(((hw[-1][0] + 1) // 2, (hw[-1][1] + 1) // 2))
    layer, op = a.name.split(".")
    B, L, dev = a.batch, _lib.lib(), "cuda"
    if layer.startswith("conv"):
        k = int(layer[4:])
        small, cs, large, cl, enc = hw[k], CH[k], hw[k - 1], CH[k - 1], True      # small grid carries cs channels
    else:
        j = int(layer[6:]) - 1
        small, cs, large, cl, enc = hw[6 - j], DEC[j][0], hw[5 - j], DEC[j][1], False
    ws = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
    s = torch.rand((B, *small, cs), device=dev) - 0.5
    l = torch.rand((B, *large, cl), device=dev) - 0.5
    w = (torch.rand(cs * cl * 25, device=dev) - 0.5) * 0.05
    S = _lib.stream_ptr
    if op == "bwd_weight":
        dw = torch.empty(cs * cl * 25, device=dev)
        run = lambda: L.svs_enc_block_bwd_weight(s.data_ptr(), cs, B, *small, cs, l.data_ptr(), cl, *large, cl, dw.data_ptr(), None,
                                                 ws.data_ptr(), ws.numel(), S())
    elif (op == "fwd") == enc:       # gather: large image -> small grid
        y = torch.empty_like(s)
        run = lambda: L.svs_enc_block_fwd(l.data_ptr(), cl, B, *large, cl, w.data_ptr(), None, None, None, 0.0, y.data_ptr(), cs, cs, 0,
                                          ws.data_ptr(), ws.numel(), S())
    else:                            # parity: small grid -> large image
        y = torch.empty_like(l)
        run = lambda: L.svs_dec_block_fwd(s.data_ptr(), cs, B, *small, cs, w.data_ptr(), None, None, None, 0.0, y.data_ptr(), cl, *large,
                                          cl, 0, ws.data_ptr(), ws.numel(), S())
    _lib.check(run(), a.name)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    gflop = 2.0 * B * small[0] * small[1] * cs * cl * 25 / 1e9
    t = e0.elapsed_time(e1) / a.reps
    print(f"{a.name} B={B}: {t * 1e3:.1f} us  {gflop / t:.1f} TFLOP/s")
    if os.environ.get("SVS_CONV_DIAG"):
        off = (ws.numel() - (32 << 20)) & ~255
        d = ws[off:off + (16 << 20)].view(torch.int64).view(-1, 4).cpu().double()
        d = d[d[:, 3] > 0]
        tot = d[:, 3].mean().item()
        print(f"  diag over {d.shape[0]} waves: wave lifetime in loop {tot:.0f} cyc; pre(reads+stores+load issue) {d[:, 0].mean().item() / tot:.1%}"
              f"  mfma block {d[:, 1].mean().item() / tot:.1%}  barrier {d[:, 2].mean().item() / tot:.1%}")


if __name__ == "__main__":
    main()
