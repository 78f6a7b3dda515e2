#!/usr/bin/env python3
"""Library kernels repeated on fixed inputs while a register-only bf16-MFMA kernel runs on another stream: which ones change?"""
import ctypes, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_unet_pytorch_amd import _lib, synth
from svs_unet_pytorch_amd.model import UNet
from svs_unet_pytorch_amd.data import istft_from_tiles, stft_to_tiles
here = os.path.dirname(os.path.abspath(__file__))
BG = ctypes.CDLL(os.path.join(here, "bin", "libbg_kernels.so"))
BG.bg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L = _lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bg_out = torch.empty(2048 * 256, device="cuda")
s_v, s_bg = torch.cuda.Stream(), torch.cuda.Stream()
model = UNet(); model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state().items()}); model.to("cuda").eval()
audio = torch.from_numpy(np.stack([synth.audio(44100 * 30, 3), synth.audio(44100 * 30, 4)])).cuda()
xt = torch.rand((16, 1, 512, 128), device="cuda")
B, n = 8, 97536
wy = torch.from_numpy((synth.uniform(60, B * n).reshape(B, n) - 0.5) * 0.5).cuda()
wx = (wy * 0.8 + 0.05).contiguous()
mrws = torch.empty(int(L.svs_mrstft_workspace_bytes(B, n)) + 4096, dtype=torch.uint8, device="cuda")
def v_stft():
    t, p, pk, T = stft_to_tiles(audio); return [t, torch.view_as_real(p), pk]
tiles0, phase0, _, T0 = stft_to_tiles(audio)
def v_istft(): return [istft_from_tiles(tiles0, None, phase0, T0)]
def v_eval():
    with torch.no_grad(): return [model(xt)]
def v_mr_value():
    loss = torch.zeros(1, device="cuda")
    _lib.check(L.svs_mrstft_loss_fwd_bwd(wx.data_ptr(), wy.data_ptr(), B, n, 1.0, loss.data_ptr(), None, mrws.data_ptr(), mrws.numel(), _lib.stream_ptr()))
    return [loss]
def v_adam_like():
    return [torch.sqrt(xt.abs() + 1.0).sum(dim=(1, 2, 3))]
tmodel = UNet(); tmodel.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=False).items()}); tmodel.to("cuda").train()
tmix = torch.rand((16, 1, 512, 128), device="cuda"); tvoc = tmix * 0.5
tmasks = [torch.from_numpy(m) for m in synth.dropout_masks(16, seed=5, step=0)]
def v_train():
    tmodel.set_dropout_masks(tmasks)
    tmodel.optim.zero_grad()
    loss = tmodel.fwd_bwd(tmix, tvoc, loss_scale=166.66)
    return [tmodel._gflat, loss]
def v_train_split():
    _lib.tuning("MFMA_SPLIT", 1)
    out = [t.clone() for t in v_train()]
    _lib.tuning("MFMA_SPLIT", -1)
    return out
def v_bf16():
    model.eval_precision = "bf16"
    with torch.no_grad(): out = [model(xt)]
    model.eval_precision = "fp32"
    return out
victims = [("train step fp32 (grads)", v_train), ("train step split mode", v_train_split), ("bf16 network forward", v_bf16), ("stft_tiles", v_stft), ("istft_tiles", v_istft), ("eval forward fp32", v_eval), ("MR-STFT value", v_mr_value), ("torch sqrt+sum", v_adam_like)]
for bg_kind, bg_name in ((None, "nothing"), (1, "fp32 MFMA"), (0, "bf16 MFMA")):
    stop = False
    def background():
        while not stop and bg_kind is not None:
            for _ in range(4): BG.bg_launch(bg_kind, bg_out.data_ptr(), 2048, 400, s_bg.cuda_stream)
            s_bg.synchronize()
    th = threading.Thread(target=background); th.start()
    with torch.cuda.stream(s_v):
        for name, fn in victims:
            ref = [t.clone() for t in fn()]; s_v.synchronize()
            bad = 0
            for _ in range(reps):
                got = fn(); s_v.synchronize()
                bad += int(not all(torch.equal(a, b) for a, b in zip(got, ref)))
            print(f"background {bg_name:10s} victim {name:20s}: {bad} of {reps} runs differ", flush=True)
    stop = True; th.join(); torch.cuda.synchronize()
