#!/usr/bin/env python3
"""Every conv GEMM call of a train step at batch B, default planner against CONV_BALANCE=0 (uniform K-splits) and against a
forced 128x128 / 64x64 tile: the results must agree to fp32 summation noise.   python tools/check_balance.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from svs_unet_pytorch_amd import _lib
L = _lib.lib(); S = _lib.stream_ptr
ws = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")
for B in [int(a) for a in sys.argv[1:]] or [64, 128]:
    for name, kind, (h, w, c, ho, wo, n), gf in bench.gemm_calls(B, "train"):
        if kind == 2:
            continue
        torch.manual_seed(1)
        x = torch.rand((B, h, w, c), device="cuda") - 0.5
        wp = (torch.rand(n * c * 25, device="cuda") - 0.5) * 0.05
        outs = {}
        for tag, sw in (("default", {}), ("uniform", {"CONV_BALANCE": 0}), ("cfg5", {"CONV_BALANCE": 0, "CONV_CFG": 5}), ("cfg0", {"CONV_BALANCE": 0, "CONV_CFG": 0})):
            _lib.tuning("*", -1)
            for k, v in sw.items():
                _lib.tuning(k, v)
            y = torch.full((B, ho, wo, n), float("nan"), device="cuda")
            fn = L.svs_enc_block_fwd if kind == 0 else L.svs_dec_block_fwd
            if kind == 0:
                rc = fn(x.data_ptr(), c, B, h, w, c, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), n, n, 0, ws.data_ptr(), ws.numel(), S())
            else:
                rc = fn(x.data_ptr(), c, B, h, w, c, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), n, ho, wo, n, 0, ws.data_ptr(), ws.numel(), S())
            if rc == 0:
                outs[tag] = y
        _lib.tuning("*", -1)
        ref = outs["default"]
        scale = ref.abs().max().item()
        msg = "  ".join(f"{t}: {(o - ref).abs().max().item() / scale:.2e}" for t, o in outs.items() if t != "default")
        bad = any(not torch.isfinite(o).all().item() for o in outs.values())
        print(f"B{B} {name:18s} max|y| {scale:.3f}  rel diff vs default -> {msg} {'NAN!' if bad else ''}", flush=True)
