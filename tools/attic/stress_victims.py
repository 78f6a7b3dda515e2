#!/usr/bin/env python3
"""Which instruction class gives different results while a bf16-MFMA kernel runs on another stream?  (tools/bg_kernels.hip)"""
import ctypes, os, sys, threading
import torch
here = os.path.dirname(os.path.abspath(__file__))
BG = ctypes.CDLL(os.path.join(here, "bin", "libbg_kernels.so"))
BG.bg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
BG.victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
BG.barrier_victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
names = ["v_fma_f32", "packed fp32", "sqrt/rcp/log/cos", "wave-level LDS exchange", "shuffles", "fp64 fma", "sincospif", "global loads"]
inp = torch.rand(65536, device="cuda") - 0.5
bg_out = torch.empty(2048 * 256, device="cuda")
s_v, s_bg = torch.cuda.Stream(), torch.cuda.Stream()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for bg_kind, bg_name in ((None, "nothing"), (1, "fp32 MFMA"), (0, "bf16 MFMA")):
    stop = False
    def background():
        while not stop and bg_kind is not None:
            for _ in range(4): BG.bg_launch(bg_kind, bg_out.data_ptr(), 2048, 400, s_bg.cuda_stream)
            s_bg.synchronize()
    th = threading.Thread(target=background); th.start()
    for kind, name in enumerate(names):
        out0 = torch.empty(1024 * 512, device="cuda"); out = torch.empty_like(out0)
        torch.cuda.current_stream().synchronize()
        BG.victim_launch(kind, inp.data_ptr(), out0.data_ptr(), 1024, 2000, s_v.cuda_stream); s_v.synchronize()
        bad = 0
        for _ in range(reps):
            BG.victim_launch(kind, inp.data_ptr(), out.data_ptr(), 1024, 2000, s_v.cuda_stream); s_v.synchronize()
            bad += int(not torch.equal(out, out0))
        print(f"background {bg_name:10s} victim {name:26s}: {bad} of {reps} runs differ", flush=True)
    for dyn in (0, 32768, 81920, 158 * 1024):
        out0 = torch.empty(1024 * 512, device="cuda"); out = torch.empty_like(out0)
        BG.barrier_victim_launch(dyn, inp.data_ptr(), out0.data_ptr(), 1024, 500, s_v.cuda_stream); s_v.synchronize()
        bad = 0
        for _ in range(reps):
            BG.barrier_victim_launch(dyn, inp.data_ptr(), out.data_ptr(), 1024, 500, s_v.cuda_stream); s_v.synchronize()
            bad += int(not torch.equal(out, out0))
        print(f"background {bg_name:10s} victim barrier exchange, {'static 32 KB' if dyn == 0 else 'dynamic %d KB' % (dyn // 1024):16s}: {bad} of {reps} runs differ", flush=True)
    BG.pk_victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    pk_names = {0: "v_pk_add_f32", 1: "v_pk_add_f32 neg", 2: "v_pk_mul_f32", 3: "v_pk_mul_f32 op_sel swap", 4: "v_pk_fma_f32", 5: "v_pk_fma_f32 op_sel+neg",
                6: "v_pk_mov_b32", 7: "scalar control", 8: "v_pk_mul op_sel_hi:[1,0] (bcast)", 9: "v_pk_mul sgpr op_sel_hi:[0,1]", 10: "v_pk_add op_sel_hi:[1,0]",
                11: "v_pk_mul op_sel:[0,1]", 12: "v_pk_fma neg only"}
    for kind, nm in pk_names.items():
        if bg_kind != 0: continue
        o0 = torch.empty(1024 * 512 * 2, device="cuda"); o = torch.empty_like(o0)
        torch.cuda.current_stream().synchronize()
        BG.pk_victim_launch(kind, inp.data_ptr(), o0.data_ptr(), 1024, 4000, s_v.cuda_stream); s_v.synchronize()
        bad = 0; worst = 0.0
        for _ in range(reps):
            BG.pk_victim_launch(kind, inp.data_ptr(), o.data_ptr(), 1024, 4000, s_v.cuda_stream); s_v.synchronize()
            if not torch.equal(o, o0):
                bad += 1; worst = max(worst, float((o - o0).abs().max()))
        print(f"background {bg_name:10s} victim {nm:34s}: {bad} of {reps} runs differ (max |d| {worst:.3e})", flush=True)
    BG.fft_victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    for mode, nm in ((0, "twiddle tables (sincospif)"), (1, "wave FFT of a pattern")):
        o0 = torch.zeros(2048 * 8 * 1024 * 2, device="cuda"); o = torch.zeros_like(o0)
        torch.cuda.current_stream().synchronize()                  # (the fills ran on the default stream; s_v does not wait for it)
        BG.fft_victim_launch(mode, o0.data_ptr(), 2048, s_v.cuda_stream); s_v.synchronize()
        bad = 0; worst = 0.0
        for _ in range(reps):
            BG.fft_victim_launch(mode, o.data_ptr(), 2048, s_v.cuda_stream); s_v.synchronize()
            if not torch.equal(o, o0):
                bad += 1; worst = max(worst, float((o - o0).abs().max()))
        print(f"background {bg_name:10s} victim {nm:28s}: {bad} of {reps} runs differ (max |d| {worst:.3e})", flush=True)
    stop = True; th.join(); torch.cuda.synchronize()
