#!/usr/bin/env python3
"""conv3 / deconv3 of the bf16 network: LDS-window form (default) against the GEMM form (SVS_BF16_CONV3_WINDOW=0 /
SVS_BF16_DECONV3_WINDOW=0) on the whole forward, several tile geometries (both accumulate in fp32 in different orders and round to bf16 once: the masks agree to a few bf16 ulps of the
intermediate activations), and the time of a 216-tile forward either way."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svs_unet_pytorch_amd import _lib, synth  # noqa: E402
from svs_unet_pytorch_amd.model import UNet  # noqa: E402

model = UNet()
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state().items()})
model.to("cuda").eval()
model.eval_precision = "bf16"
bad = 0
SWITCH = sys.argv[1] if len(sys.argv) > 1 else "BF16_CONV3_WINDOW"
for shape in ((4, 1, 512, 128), (3, 1, 513, 128), (2, 1, 512, 100), (1, 1, 300, 77), (16, 1, 512, 128), (130, 1, 512, 128)):
    torch.manual_seed(1)
    x = torch.rand(shape, device="cuda")
    with torch.no_grad():
        _lib.tuning(SWITCH, 0)
        ref = model(x).clone()
        _lib.tuning(SWITCH, -1)
        out = model(x).clone()
        model.eval_precision = "fp32"
        f32 = model(x).clone()
        model.eval_precision = "bf16"
    d = (out - ref).abs()
    print(f"{shape}: window vs GEMM form max {d.max().item():.2e} mean {d.mean().item():.2e};  vs fp32: window mean {(out - f32).abs().mean().item():.2e}, GEMM mean {(ref - f32).abs().mean().item():.2e}")
    if d.mean().item() > 2e-4 or d.max().item() > 2e-2:
        bad += 1
x = torch.rand((216, 1, 512, 128), device="cuda")
for val in (0, -1, 0, -1):
    _lib.tuning(SWITCH, val)
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            model(x)
        b.record()
        torch.cuda.synchronize()
    print(f"{SWITCH}={val}: 216-tile bf16 forward {a.elapsed_time(b) / 20:.4f} ms")
print("OK" if not bad else "FAILED")
sys.exit(1 if bad else 0)
