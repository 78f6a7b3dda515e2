#!/usr/bin/env python3
"""One process, two streams: the MR-STFT loss (value + gradient) repeated on fixed inputs while another stream keeps the GPU
busy with (a) the bf16 network, (b) fp32 train steps in split-bf16 mode, (c) fp32 train steps, (d) nothing.  Counts
repetitions whose gradient differs bitwise from the first one.    python tools/stress_mr_concurrent.py [reps=200]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_unet_pytorch_amd import _lib, synth
from svs_unet_pytorch_amd.model import UNet
L = _lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, n = 8, 97536
y = torch.from_numpy((synth.uniform(60, B * n).reshape(B, n) - 0.5) * 0.5).cuda()
x = (y * 0.8 + torch.from_numpy((synth.uniform(61, B * n).reshape(B, n) - 0.5) * 0.1).cuda()).contiguous()
ws = torch.empty(int(L.svs_mrstft_workspace_bytes(B, n)) + 4096, dtype=torch.uint8, device="cuda")
model = UNet()
model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=False).items()})
model.to("cuda")
tiles = torch.rand((64, 1, 512, 128), device="cuda")
voc = tiles * 0.5
s_mr, s_bg = torch.cuda.Stream(), torch.cuda.Stream()
def mr_once():
    loss = torch.zeros(1, device="cuda"); d = torch.empty_like(x)
    _lib.check(L.svs_mrstft_loss_fwd_bwd(x.data_ptr(), y.data_ptr(), B, n, 1.0, loss.data_ptr(), d.data_ptr(), ws.data_ptr(), ws.numel(), s_mr.cuda_stream))
    return loss, d
import ctypes
BG = None
bgpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libbg_kernels.so")
if os.path.exists(bgpath):
    BG = ctypes.CDLL(bgpath)
    BG.bg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
bg_out = torch.empty(2048 * 256, device="cuda")
SYN = {"syn: bf16 MFMA, no LDS": 0, "syn: fp32 MFMA, no LDS": 1, "syn: LDS traffic, no MFMA": 2, "syn: cvt_pk_bf16 + VALU": 3, "syn: bf16 MFMA + LDS": 4}
labels = ["nothing"] + (list(SYN) if BG is not None else []) + ["bf16 network", "fp32 train steps", "split-bf16 train steps"]
for label in labels:
    stop = False
    def background():
        with torch.cuda.stream(s_bg):
            while not stop:
                if label in SYN:
                    for _ in range(4): BG.bg_launch(SYN[label], bg_out.data_ptr(), 2048, 400, s_bg.cuda_stream)
                elif label == "bf16 network":
                    model.eval(); model.eval_precision = "bf16"
                    with torch.no_grad():
                        for _ in range(4): model(tiles)
                elif label != "nothing":
                    _lib.tuning("MFMA_SPLIT", 1 if label.startswith("split") else -1)
                    model.train(); model.eval_precision = "fp32"
                    for _ in range(2): model.train_step(tiles, voc, loss_scale=166.66)
                s_bg.synchronize()
    th = threading.Thread(target=background); th.start()
    with torch.cuda.stream(s_mr):
        l0, d0 = mr_once(); s_mr.synchronize()
        bad = 0
        for i in range(reps):
            l, d = mr_once(); s_mr.synchronize()
            if not torch.equal(d, d0) or l.item() != l0.item():
                bad += 1
                if bad <= 3: print(f"   [{label}] rep {i}: {int((d != d0).sum())} elements differ, max |d| {(d - d0).abs().max().item():.3e}, loss {l.item()} vs {l0.item()}", flush=True)
    stop = True; th.join(); torch.cuda.synchronize()
    _lib.tuning("MFMA_SPLIT", -1)
    print(f"background = {label}: {bad} of {reps} MR-STFT repetitions differ from the first", flush=True)
