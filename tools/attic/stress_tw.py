import ctypes, os, sys, threading
import torch
here = os.path.dirname(os.path.abspath(__file__))
BG = ctypes.CDLL(os.path.join(here, "bin", "libbg_kernels.so"))
BG.bg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
BG.fft_victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
bg_out = torch.empty(2048 * 256, device="cuda")
s_v, s_bg = torch.cuda.Stream(), torch.cuda.Stream()
TW = 16 * 16 + 1024
o0 = torch.zeros(2048 * TW * 2, device="cuda"); o = torch.zeros_like(o0)
BG.fft_victim_launch(0, o0.data_ptr(), 2048, s_v.cuda_stream); s_v.synchronize()
ref = o0.view(2048, TW, 2)
print("all blocks agree with block 0 when alone:", bool((ref == ref[0:1]).all()))
for kind in (1, 0):
    stop = False
    def background():
        while not stop:
            for _ in range(4): BG.bg_launch(kind, bg_out.data_ptr(), 2048, 400, s_bg.cuda_stream)
            s_bg.synchronize()
    th = threading.Thread(target=background); th.start()
    for rep in range(3):
        o.zero_()
        BG.fft_victim_launch(0, o.data_ptr(), 2048, s_v.cuda_stream); s_v.synchronize()
        got = o.view(2048, TW, 2)
        d = (got != ref).any(dim=2)
        blocks = d.any(dim=1).nonzero().flatten()
        print(f"bg kind {kind} rep {rep}: {int(d.sum())} entries differ in {blocks.numel()} blocks; first blocks {blocks[:8].tolist()}")
        if blocks.numel():
            b = int(blocks[0]); es = d[b].nonzero().flatten()
            print("    block", b, "entries", es[:10].tolist(), "... count", es.numel(), "got", got[b, es[:4]].tolist(), "want", ref[b, es[:4]].tolist())
    stop = True; th.join(); torch.cuda.synchronize()
