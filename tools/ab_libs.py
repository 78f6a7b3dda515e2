#!/usr/bin/env python3
"""A/B two builds of libsvs_hip.so on every conv GEMM call of a training step, interleaved in ONE process on
ONE device (devices differ by up to ~12%, so numbers from different gpurun calls cannot be compared).

    python tools/ab_libs.py tools/bin/libsvs_hip_A.so svs_unet_pytorch_amd/libsvs_hip.so [--batch 64]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd import _lib  # noqa: E402

CH = (1, 16, 32, 64, 128, 256, 512)
DEC = ((512, 256), (512, 128), (256, 64), (128, 32), (64, 16))


def load(path):
    h = ctypes.CDLL(os.path.abspath(path))
    for name in ("svs_enc_block_fwd", "svs_dec_block_fwd", "svs_enc_block_bwd_weight"):
        fn = getattr(h, name)
        fn.restype, fn.argtypes = _lib._SIGS[name]
    return h


def timeit(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib_a")
    ap.add_argument("lib_b")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--wgrad", action="store_true")
    a = ap.parse_args()
    torch.zeros(1, device="cuda")
    libs = [load(a.lib_a), load(a.lib_b)]
    B, dev = a.batch, "cuda"
    hw = [(512, 128)]
    for _ in range(6):
        hw.append(((hw[-1][0] + 1) // 2, (hw[-1][1] + 1) // 2))
    calls = []
    for k in range(2, 7):
        calls.append((f"conv{k}.fwd", "gather", (*hw[k - 1], CH[k - 1]), (*hw[k], CH[k])))
        calls.append((f"conv{k}.bwd_data", "parity", (*hw[k], CH[k]), (*hw[k - 1], CH[k - 1])))
    for j, (c, n) in enumerate(DEC):
        calls.append((f"deconv{j + 1}.fwd", "parity", (*hw[6 - j], c), (*hw[5 - j], n)))
        calls.append((f"deconv{j + 1}.bwd_data", "gather", (*hw[5 - j], n), (*hw[6 - j], c)))
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    S = _lib.stream_ptr
    tot = [0.0, 0.0]
    for name, mode, (h, w, C), (ho, wo, N) in calls:
        x = torch.rand((B, h, w, C), device=dev) - 0.5
        wp = (torch.rand(N * C * 25, device=dev) - 0.5) * 0.05
        ys = [torch.empty((B, ho, wo, N), device=dev) for _ in libs]
        runs = []
        for L, y in zip(libs, ys):
            if mode == "gather":
                runs.append(lambda L=L, y=y: L.svs_enc_block_fwd(x.data_ptr(), C, B, h, w, C, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(),
                                                                 N, N, 0, ws.data_ptr(), ws.numel(), S()))
            else:
                runs.append(lambda L=L, y=y: L.svs_dec_block_fwd(x.data_ptr(), C, B, h, w, C, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(),
                                                                 N, ho, wo, N, 0, ws.data_ptr(), ws.numel(), S()))
        for r in runs:
            assert r() == 0
        torch.cuda.synchronize()
        same = (ys[0] - ys[1]).abs().max().item()
        best = [1e9, 1e9]
        for _ in range(3):
            for i, r in enumerate(runs):
                best[i] = min(best[i], timeit(r))
        gflop = 2.0 * B * (ho * wo if mode == "gather" else h * w) * N * C * 25 / 1e9
        tot[0] += best[0]
        tot[1] += best[1]
        print(f"{name:18s} A {best[0] * 1e3:7.1f} us {gflop / best[0]:6.1f} TF   B {best[1] * 1e3:7.1f} us {gflop / best[1]:6.1f} TF   B/A time {best[1] / best[0]:.3f}  maxdiff {same:.1e}",
              flush=True)
    print(f"TOTAL conv A {tot[0] * 1e3:.1f} us   B {tot[1] * 1e3:.1f} us   B/A {tot[1] / tot[0]:.3f}")
    if not a.wgrad:
        return
    tot = [0.0, 0.0]
    wg = [(f"conv{k}.bwd_weight", hw[k], CH[k], hw[k - 1], CH[k - 1]) for k in range(2, 7)]
    wg += [(f"deconv{j + 1}.bwd_weight", hw[6 - j], c, hw[5 - j], n) for j, (c, n) in enumerate(DEC)]
    for name, (hs, wsz), cs, (hl, wl), cl in wg:
        sm = torch.rand((B, hs, wsz, cs), device=dev) - 0.5
        lg = torch.rand((B, hl, wl, cl), device=dev) - 0.5
        dws = [torch.empty(cs * cl * 25, device=dev) for _ in libs]
        runs = [lambda L=L, dw=dw: L.svs_enc_block_bwd_weight(sm.data_ptr(), cs, B, hs, wsz, cs, lg.data_ptr(), cl, hl, wl, cl, dw.data_ptr(),
                                                              None, ws.data_ptr(), ws.numel(), S()) for L, dw in zip(libs, dws)]
        for r in runs:
            assert r() == 0
        torch.cuda.synchronize()
        same = ((dws[0] - dws[1]).abs().max() / dws[0].abs().max()).item()
        best = [1e9, 1e9]
        for _ in range(3):
            for i, r in enumerate(runs):
                best[i] = min(best[i], timeit(r))
        gflop = 2.0 * B * hs * wsz * cs * cl * 25 / 1e9
        tot[0] += best[0]
        tot[1] += best[1]
        print(f"{name:20s} A {best[0] * 1e3:7.1f} us {gflop / best[0]:6.1f} TF   B {best[1] * 1e3:7.1f} us {gflop / best[1]:6.1f} TF   B/A time {best[1] / best[0]:.3f}  reldiff {same:.1e}",
              flush=True)
    print(f"TOTAL wgrad A {tot[0] * 1e3:.1f} us   B {tot[1] * 1e3:.1f} us   B/A {tot[1] / tot[0]:.3f}")


if __name__ == "__main__":
    main()
