#!/usr/bin/env python3
"""Times every MFMA GEMM call of one training step (forward, backward-data, backward-weight of each
layer) at a given batch, for every tile configuration / K-split the library can be forced into
(CONV_CFG, CONV_KSPLIT, WGRAD_KSPLIT: planner overrides set through svs_tuning_set).
Used to derive the planner's tables; writes gpurun_out/gemm_sweep_B<B>.txt.

    python tools/gemm_sweep.py [--batch 64] [--quick]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd import _lib  # noqa: E402

CH = (1, 16, 32, 64, 128, 256, 512)
DEC = ((512, 256), (512, 128), (256, 64), (128, 32), (64, 16))
CFG = {0: (128, 128), 1: (128, 64), 2: (256, 32), 3: (256, 16), 4: (32, 128), 5: (64, 64), 6: (64, 128)}


def sizes(H=512, W=128):
    hw = [(H, W)]
    for _ in range(6):
        hw.append(((hw[-1][0] + 1) // 2, (hw[-1][1] + 1) // 2))
    return hw


def timeit(fn, reps=8):
    for _ in range(2):
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated substrings of call names to run")
    ap.add_argument("--cfgs", default="", help="comma-separated config ids to try (default all)")
    ap.add_argument("--no-wgrad", action="store_true")
    ap.add_argument("--no-conv", action="store_true")
    ap.add_argument("--no-ks", action="store_true", help="skip the K-split sweep of the weight gradients")
    ap.add_argument("--ab", action="store_true", help="A/B an environment switch per call (see --ab-env)")
    ap.add_argument("--ab-env", default="CONV_KORDER", help="switch toggled by --ab")
    ap.add_argument("--ab-vals", default="0,1", help="comma-separated values of the switch")
    ap.add_argument("--ks", default="1,2,4,8", help="K-splits to try for the conv GEMMs")
    ap.add_argument("--fwd-only", action="store_true", help="forward calls only (eval)")
    args = ap.parse_args()
    args.ks_list = tuple(int(t) for t in args.ks.split(","))
    B = args.batch
    L = _lib.lib()
    dev = "cuda"
    hw = sizes()
    out = open(os.path.join(ROOT, "gpurun_out", f"gemm_sweep_B{B}.txt"), "w")

    def emit(s):
        print(s, flush=True)
        out.write(s + "\n")
        out.flush()

    calls = []   # (name, mode, (h, w, C), (ho, wo, N))
    for k in range(2, 7):
        calls.append((f"conv{k}.fwd", "gather", (*hw[k - 1], CH[k - 1]), (*hw[k], CH[k])))
        calls.append((f"conv{k}.bwd_data", "parity", (*hw[k], CH[k]), (*hw[k - 1], CH[k - 1])))
    for j, (c, n) in enumerate(DEC):
        calls.append((f"deconv{j + 1}.fwd", "parity", (*hw[6 - j], c), (*hw[5 - j], n)))
        calls.append((f"deconv{j + 1}.bwd_data", "gather", (*hw[5 - j], n), (*hw[6 - j], c)))
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    only = [t for t in args.only.split(",") if t]
    cfgs = [int(t) for t in args.cfgs.split(",") if t]
    for name, mode, (h, w, C), (ho, wo, N) in calls:
        if args.no_conv or (only and not any(t in name for t in only)) or (args.fwd_only and not name.endswith(".fwd")):
            continue
        x = torch.rand((B, h, w, C), device=dev) - 0.5
        wp = (torch.rand(N * C * 25, device=dev) - 0.5) * 0.05
        y = torch.empty((B, ho, wo, N), device=dev)
        gflop = 2.0 * B * (ho * wo if mode == "gather" else h * w) * N * C * 25 / 1e9
        if mode == "gather":
            run = lambda: L.svs_enc_block_fwd(x.data_ptr(), C, B, h, w, C, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), N, N, 0,
                                              ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        else:
            run = lambda: L.svs_dec_block_fwd(x.data_ptr(), C, B, h, w, C, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), N, ho, wo, N, 0,
                                              ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.tuning("CONV_CFG")
        _lib.tuning("CONV_KSPLIT")
        _lib.check(run(), name)
        base = timeit(run)
        emit(f"{name:18s} {mode:6s} in {h}x{w}x{C} out {ho}x{wo}x{N} {gflop:7.2f} GFLOP  default {base * 1e3:8.1f} us {gflop / base:6.1f} TF")
        best = (base, "default")
        for cfg, (bm, bn) in CFG.items():
            if N % bn or (cfgs and cfg not in cfgs):
                continue
            for ks in (args.ks_list if not args.quick else (1, 4)):
                _lib.tuning("CONV_CFG", cfg)
                _lib.tuning("CONV_KSPLIT", ks)
                if run() != 0:
                    continue
                t = timeit(run)
                emit(f"    cfg{cfg} {bm:3d}x{bn:3d} ks{ks:<3d} {t * 1e3:8.1f} us {gflop / t:6.1f} TF")
                if t < best[0]:
                    best = (t, f"cfg{cfg} ks{ks}")
        _lib.tuning("CONV_CFG")
        _lib.tuning("CONV_KSPLIT")
        if args.ab:
            res = []
            for rnd in range(3):               # interleaved A/B in one process (guide rule 24)
                for ko in args.ab_vals.split(","):
                    _lib.tuning(args.ab_env, int(ko))
                    run()
                    res.append((ko, timeit(run)))
            _lib.tuning(args.ab_env)
            emit(f"    A/B {args.ab_env}: " + "  ".join(f"={v} {min(t for k, t in res if k == v) * 1e3:7.1f} us" for v in args.ab_vals.split(",")))
        emit(f"  -> best {best[1]} {best[0] * 1e3:.1f} us {gflop / best[0]:.1f} TF")
    _lib.tuning("CONV_CFG")
    _lib.tuning("CONV_KSPLIT")

    wg = []
    if args.no_wgrad:
        out.close()
        return
    for k in range(2, 7):
        wg.append((f"conv{k}.bwd_weight", hw[k], CH[k], hw[k - 1], CH[k - 1]))
    for j, (c, n) in enumerate(DEC):
        wg.append((f"deconv{j + 1}.bwd_weight", hw[6 - j], c, hw[5 - j], n))
    for name, (hs, wsz), cs, (hl, wl), cl in wg:
        if only and not any(t in name for t in only):
            continue
        s = torch.rand((B, hs, wsz, cs), device=dev) - 0.5
        l = torch.rand((B, hl, wl, cl), device=dev) - 0.5
        dw = torch.empty(cs * cl * 25, device=dev)
        gflop = 2.0 * B * hs * wsz * cs * cl * 25 / 1e9
        run = lambda: L.svs_enc_block_bwd_weight(s.data_ptr(), cs, B, hs, wsz, cs, l.data_ptr(), cl, hl, wl, cl, dw.data_ptr(), None,
                                                 ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.tuning("WGRAD_KSPLIT")
        _lib.check(run(), name)
        base = timeit(run)
        emit(f"{name:20s} S {hs}x{wsz}x{cs} L {hl}x{wl}x{cl} {gflop:7.2f} GFLOP  default {base * 1e3:8.1f} us {gflop / base:6.1f} TF")
        if args.ab:
            res = []
            for rnd in range(3):
                for ko in args.ab_vals.split(","):
                    _lib.tuning(args.ab_env, int(ko))
                    run()
                    res.append((ko, timeit(run)))
            _lib.tuning(args.ab_env)
            emit(f"    A/B {args.ab_env}: " + "  ".join(f"={v} {min(t for k, t in res if k == v) * 1e3:7.1f} us" for v in args.ab_vals.split(",")))
        for ks in (() if args.no_ks else (1, 2, 4, 8, 16, 32, 64, 128, 256)):
            _lib.tuning("WGRAD_KSPLIT", ks)
            if run() != 0:
                continue
            t = timeit(run)
            emit(f"    ks{ks:<4d} {t * 1e3:8.1f} us {gflop / t:6.1f} TF")
    _lib.tuning("WGRAD_KSPLIT")
    out.close()


if __name__ == "__main__":
    main()
