import ctypes, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from svs_unet_pytorch_amd import _lib, synth
from svs_unet_pytorch_amd.data import stft_to_tiles
here = os.path.dirname(os.path.abspath(__file__))
BG = ctypes.CDLL(os.path.join(here, "bin", "libbg_kernels.so"))
BG.bg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
bg_out = torch.empty(2048 * 256, device="cuda")
s_v, s_bg = torch.cuda.Stream(), torch.cuda.Stream()
audio = torch.from_numpy(np.stack([synth.audio(44100 * 30, 3), synth.audio(44100 * 30, 4)])).cuda()
with torch.cuda.stream(s_v):
    t0, p0, pk0, T = stft_to_tiles(audio); s_v.synchronize()
stop = False
def background():
    while not stop:
        for _ in range(4): BG.bg_launch(0, bg_out.data_ptr(), 2048, 400, s_bg.cuda_stream)
        s_bg.synchronize()
th = threading.Thread(target=background); th.start()
with torch.cuda.stream(s_v):
    for rep in range(4):
        t, p, pk, _ = stft_to_tiles(audio); s_v.synchronize()
        d = (t != t0)
        nt = t.shape[1]
        full = lambda a: a[:, :, 0].permute(0, 2, 1, 3).reshape(2, 512, nt * 128)
        D = full(d); A = full(t); A0 = full(t0)
        frames = D.any(dim=1)            # (2, frames)
        bins = D.any(dim=2)
        idx = frames.nonzero()
        print(f"rep {rep}: {int(d.sum())} of {d.numel()} magnitudes differ; frames affected {int(frames.sum())} of {frames.numel()}; bins affected {int(bins.sum())} of {bins.numel()}; "
              f"max |d| {float((A - A0).abs().max()):.3e} (peak {float(A0.max()):.1f}); first frames {idx[:12].tolist()}", flush=True)
        dp = (torch.view_as_real(p) - torch.view_as_real(p0)).abs()
        print(f"        phasors: {int((dp > 0).sum())} differ, max {float(dp.max()):.3e}")
stop = True; th.join()
