#!/usr/bin/env python3
"""Does a workgroup's LDS keep its contents while other kernels run on the GPU?  (tools/bg_kernels.hip: lds_guard_kernel)"""
import ctypes, os, sys, threading
import torch
here = os.path.dirname(os.path.abspath(__file__))
BG = ctypes.CDLL(os.path.join(here, "bin", "libbg_kernels.so"))
BG.bg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
BG.lds_guard_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
bg_out = torch.empty(2048 * 256, device="cuda")
s_v, s_bg = torch.cuda.Stream(), torch.cuda.Stream()
for bg_kind, bg_name in ((None, "nothing"), (1, "fp32 MFMA"), (0, "bf16 MFMA"), (2, "LDS 48 KB kernel")):
    stop = False
    def background():
        while not stop and bg_kind is not None:
            for _ in range(4): BG.bg_launch(bg_kind, bg_out.data_ptr(), 2048, 400, s_bg.cuda_stream)
            s_bg.synchronize()
    th = threading.Thread(target=background); th.start()
    for kb in (16, 48, 64, 72, 80, 120, 158):
        rep = torch.zeros(4 + 64 * 4, dtype=torch.int32, device="cuda")
        for _ in range(10):
            BG.lds_guard_launch(kb * 1024, rep.data_ptr(), 2048, 20000, s_v.cuda_stream)
        s_v.synchronize()
        r = rep.cpu().numpy().astype("uint32")
        msg = f"background {bg_name:16s} dynamic LDS {kb:3d} KB: {r[0]} blocks of 20480 saw foreign data ({r[1]} words)"
        if r[0]:
            ex = [(int(r[4 + 4 * i]), int(r[5 + 4 * i]) * 4, hex(int(r[6 + 4 * i])), int(r[7 + 4 * i])) for i in range(min(3, int(r[0])))]
            msg += f"; e.g. (block, byte offset, value, words) {ex}"
        print(msg, flush=True)
    stop = True; th.join(); torch.cuda.synchronize()
