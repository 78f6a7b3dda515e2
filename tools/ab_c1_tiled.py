#!/usr/bin/env python3
"""Same-process A/B of the two forms of the single-channel convolution (conv1 forward, deconv6 backward-data) at batch B:
    python tools/ab_c1_tiled.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from svs_unet_pytorch_amd import _lib
L = _lib.lib(); S = _lib.stream_ptr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.rand((B, 512, 128), device="cuda")
ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for N in (16, 32):
    w = torch.rand(N * 25, device="cuda") - 0.5
    bias = torch.rand(N, device="cuda")
    y = torch.empty((B, 256, 64, N), device="cuda")
    run = lambda: _lib.check(L.svs_enc_block_fwd(x.data_ptr(), 1, B, 512, 128, 1, w.data_ptr(), bias.data_ptr(), None, None, 0.0, y.data_ptr(), N, N, 0,
                                                 ws.data_ptr(), ws.numel(), S()))
    res = {}
    outs = {}
    for rnd in range(3):
        for v in (0, -1):
            _lib.tuning("CONV_C1_TILED", v)
            res.setdefault(v, []).append(timeit(run))
            outs[v] = y.clone()
    mb = (B * 512 * 128 * 4 + y.numel() * 4) / 1e6
    print(f"N={N}: thread-per-pixel {min(res[0]):.1f} us ({mb / min(res[0]):.2f} TB/s)  tiled {min(res[-1]):.1f} us ({mb / min(res[-1]):.2f} TB/s)  "
          f"max |diff| {(outs[0] - outs[-1]).abs().max().item():.2e}")
