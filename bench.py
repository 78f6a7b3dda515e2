#!/usr/bin/env python3
"""Headline benchmark: spectrogram tiles/sec (512x128) of the U-Net training step (fwd + L1 loss + bwd +
gradient all-reduce + Adam) on N GPUs of one node, one process per GPU (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|eval] [--batch B_per_gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload: BASELINE.json configs[2] at N=1 (train.py L1-loss step, batch 64 on one MI355X) and configs[3] at
N=8 (64 tiles per GPU = global batch 512, RCCL gradient all-reduce): weak scaling.  `--mode eval --batch 16`
is configs[1] (eval forward).  Inputs are synthetic tiles (svs_fill_tiles) already resident in HBM.

One JSON line on rank 0 with `roofline` (the dominant kernel timed live with HIP events on the stream it is
launched on, against the fp32 MFMA peak) and `cpu_baseline` (the CPU oracle -- the plain-torch restatement of
the reference -- timed on this box's host cores on a bounded sample).  The oracle is used ONLY there.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from svs_unet_pytorch_amd import _lib, synth  # noqa: E402
from svs_unet_pytorch_amd.model import ALPHA_L1, DEC_IO, ENC_CHANNELS, UNet  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
FWD_GFLOP_PER_TILE = 1.507328
TRAIN_GFLOP_PER_TILE = 4.5088768


def gemm_calls(B, mode, H=512, W=128):
    """Every MFMA GEMM call of one step at batch B: (name, kind, args, GFLOP).  kind 0 = gather GEMM, 1 = parity
    GEMM, 2 = weight gradient (include/svs_hip.h: svs_describe_plan).  Eval has the 10 forward calls only."""
    hw = [(H, W)]
    for _ in range(6):
        hw.append(((hw[-1][0] + 1) // 2, (hw[-1][1] + 1) // 2))
    calls = []
    for k in range(2, 7):
        (h, w), (ho, wo) = hw[k - 1], hw[k]
        c, n = ENC_CHANNELS[k - 1], ENC_CHANNELS[k]
        gf = 2.0 * B * ho * wo * n * c * 25 / 1e9
        calls.append((f"conv{k}.fwd", 0, (h, w, c, ho, wo, n), gf))
        if mode == "train":
            calls.append((f"conv{k}.bwd_data", 1, (ho, wo, n, h, w, c), gf))
            calls.append((f"conv{k}.bwd_weight", 2, (ho, wo, n, h, w, c), gf))
    for j in range(5):
        (h, w), (ho, wo) = hw[6 - j], hw[5 - j]
        c, n = DEC_IO[j]
        gf = 2.0 * B * h * w * n * c * 25 / 1e9
        calls.append((f"deconv{j + 1}.fwd", 1, (h, w, c, ho, wo, n), gf))
        if mode == "train":
            calls.append((f"deconv{j + 1}.bwd_data", 0, (ho, wo, n, h, w, c), gf))
            calls.append((f"deconv{j + 1}.bwd_weight", 2, (h, w, c, ho, wo, n), gf))
    return calls


def useful_fraction(kind, dims):
    """Share of a layer's nominal 2*M*N*K multiply-adds whose input tap lies inside the image (the rest multiply the
    zero padding).  Nominal FLOPs are what SURVEY.md 8(d) counts; the `..., true>` kernels skip K-tiles that are all
    padding, so their MFMA rate is quoted on both bases.  kind: 0 gather, 1 parity, 2 weight gradient."""
    h, w, c, ho, wo, n = dims
    def gather(hs, ws, hl, wl):            # 5x5 window, stride 2, pad 2: small grid (hs, ws) over large grid (hl, wl)
        vh = sum(sum(0 <= 2 * i - 2 + k < hl for k in range(5)) for i in range(hs))
        vw = sum(sum(0 <= 2 * j - 2 + k < wl for k in range(5)) for j in range(ws))
        return vh * vw / (25.0 * hs * ws)
    if kind == 0:
        return gather(ho, wo, h, w)
    if kind == 2:
        return gather(h, w, ho, wo)
    def axis(n_in, n_out):                 # transposed conv: output o takes taps k with (o + 2 - k) even and in range
        nominal = valid = 0
        for o in range(n_out):
            ks = [k for k in range(5) if (o + 2 - k) % 2 == 0]
            nominal += len(ks)
            valid += sum(0 <= (o + 2 - k) // 2 < n_in for k in ks)
        return valid, nominal
    (vh, nh), (vw, nw) = axis(h, ho), axis(w, wo)
    return vh * vw / float(nh * nw)


def time_gemm_calls(B, mode, reps=10, only_kernel=None):
    """Each GEMM call of a step, timed with HIP events on torch's current stream (the stream every launch of the
    library is given), attributed to the kernel the planner picks (svs_describe_plan -> the name rocprofv3 shows).
    A call = the GEMM kernel plus, where the planner splits K, its fixed-order slab reduction.
    Returns [(name, kernel, ksplit, ms, GFLOP, share of the GFLOP that is not zero padding)]."""
    import ctypes
    L = _lib.lib()
    dev = "cuda"
    ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    buf = ctypes.create_string_buffer(128)
    out = []
    for name, kind, (h, w, c, ho, wo, n), gflop in gemm_calls(B, mode):
        x = torch.rand((B, h, w, c), device=dev) - 0.5
        if kind == 2:
            other = torch.rand((B, ho, wo, n), device=dev) - 0.5
            dw = torch.empty(c * n * 25, device=dev)
            ks = L.svs_describe_plan(2, B, h, w, c, 0, 0, n, buf, 128)
            run = lambda: L.svs_enc_block_bwd_weight(x.data_ptr(), c, B, h, w, c, other.data_ptr(), n, ho, wo, n, dw.data_ptr(), None,
                                                     ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        else:
            wp = (torch.rand(n * c * 25, device=dev) - 0.5) * 0.05
            y = torch.empty((B, ho, wo, n), device=dev)
            ks = L.svs_describe_plan(kind, B, h, w, c, ho, wo, n, buf, 128)
            if kind == 0:
                run = lambda: L.svs_enc_block_fwd(x.data_ptr(), c, B, h, w, c, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), n, n, 0,
                                                  ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            else:
                run = lambda: L.svs_dec_block_fwd(x.data_ptr(), c, B, h, w, c, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), n, ho, wo,
                                                  n, 0, ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        if only_kernel is not None and buf.value.decode() != only_kernel:
            continue
        for _ in range(3):
            _lib.check(run(), name)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        out.append((name, buf.value.decode(), ks, e0.elapsed_time(e1) / reps, gflop, useful_fraction(kind, (h, w, c, ho, wo, n))))
    return out


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source) from the newest rocprofv3 --pmc summary kept under profiles/ (two separate
    passes, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as is -- tools/pmc_traffic.py).
    The counters cannot be collected from inside a timed run, so this is a RECORDED number: `source` names the file (which
    carries the commit it was taken at); (None, None) when no summary covers the kernel."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if kernel in d:
                return d[kernel].get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT) + (" @ " + d["_commit"] if "_commit" in d else "")
        except Exception:
            pass
    return None, None


def usable_cores():
    """CPU threads this process can really run: the affinity mask, cut to the cgroup CPU quota when there is one (a GPU box
    hands one GPU's job a 16-CPU share of a much larger host; oversubscribing it makes the oracle many times slower)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(q / int(g.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, 16)              # the share of one GPU's job on the pool's hosts


def cpu_baseline(mode, seconds_budget=24.0):
    """The CPU oracle (oracle/unet_oracle.py, the plain-torch restatement of the reference) on this box's host cores, fp32,
    same synthetic tiles: the three cases of SURVEY.md 8(d) -- config 1 (eval forward B=1, the inference.py tile loop),
    eval forward B=16, one L1 train step (fwd + bwd + Adam) at B=64 -- 2 warm-up + >= 5 timed iterations each within a
    bounded budget.  `value` is the case that corresponds to the GPU line of this run."""
    from oracle import unet_oracle as uo
    cores = usable_cores()
    torch.set_num_threads(cores)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    st = uo.to_torch_state(synth.closed_form_state(trained_stats=False))

    def run(B, train, budget):
        mix_np, voc_np = synth.tiles(B)
        mix, voc = torch.from_numpy(mix_np), torch.from_numpy(voc_np)
        if train:
            opt = uo.new_adam_state(st)
            masks = [torch.from_numpy(m) for m in synth.dropout_masks(B, seed=1)]
            step = lambda: uo.train_step(st, opt, mix, voc, dropout_masks=masks, loss_scale=ALPHA_L1)
        else:
            def step():
                with torch.no_grad():
                    uo.forward(st, mix, training=False)
        for _ in range(2):
            step()
        t0 = time.perf_counter()
        it = 0
        while True:
            step()
            it += 1
            el = time.perf_counter() - t0
            if (it >= 5 and el > budget) or it >= 50:
                break
        return {"tiles_per_s": round(B * it / el, 2), "batch": B, "iterations": it, "seconds": round(el, 2)}

    cases = {"eval_b1": run(1, False, seconds_budget / 8), "eval_b16": run(16, False, seconds_budget / 4),
             "train_b64": run(64, True, seconds_budget / 2)}
    key = "train_b64" if mode == "train" else "eval_b16"
    return {"value": cases[key]["tiles_per_s"], "unit": "tiles/s", "cores": cores, "kind": "port", "cpu": cpu_model,
            "sample": f"{cases[key]['iterations']} {'L1 train steps (fwd+bwd+Adam)' if mode == 'train' else 'eval forwards'} of batch "
                      f"{cases[key]['batch']}, fp32 torch CPU oracle, {cores} threads, {cases[key]['seconds']} s",
            "cases": cases}


def split_mode_record(model, dev, B=64, steps=20, warmup=5):
    """The train step again with the OPTIONAL split-bf16 product mode of the GEMM kernels (csrc/mfma_split.h, DESIGN.md section 5;
    off by default and not what `value` measures): fp32 operands, products formed from exact three-limb bf16 splits on the
    bf16 MFMA, fp32 accumulation."""
    H, W = 512, 128
    mix = torch.empty((B, 1, H, W), device=dev)
    voc = torch.empty_like(mix)
    _lib.check(_lib.lib().svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), B, H, W, 30_000, _lib.stream_ptr()), "svs_fill_tiles")
    model.train()
    out = {}
    for name, val in (("default", -1), ("mfma_split", 1)):
        _lib.tuning("MFMA_SPLIT", val)
        for _ in range(warmup):
            model.train_step(mix, voc, loss_scale=ALPHA_L1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.train_step(mix, voc, loss_scale=ALPHA_L1)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        out[name] = {"ms_per_step": round(ms, 4), "tiles_per_s": round(B / ms * 1e3, 1)}
    _lib.tuning("MFMA_SPLIT", -1)
    out["note"] = "same process, same device, back to back; `value` above is the default mode"
    return out


def split_mode_accuracy(dev, B=4, H=32, W=16, C=128, N=256):
    """Both product modes of the conv GEMM on the same operands (four decades of dynamic range) against torch's float64
    convolution on the CPU: error relative to sum |x w| per output, the scale of a dot product's rounding error
    (tests/test_gpu_ops.py::test_mfma_split_mode_accuracy is the gated version of this, on four kinds of call)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(7)
    spread = lambda shape: (torch.rand(shape, generator=g) - 0.5) * torch.exp(4.0 * (torch.rand(shape, generator=g) - 0.5))
    x, w = spread((B, C, H, W)), spread((N, C, 5, 5)) * 0.1
    want = F.conv2d(x.double(), w.double(), None, stride=2, padding=2)
    mag = F.conv2d(x.double().abs(), w.double().abs(), None, stride=2, padding=2)
    L, S = _lib.lib(), _lib.stream_ptr
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, wp = w.to(dev).contiguous(), torch.empty(N * C * 25, device=dev)
    _lib.check(L.svs_pack_weight_gather(wd.data_ptr(), wp.data_ptr(), N, C, S()), "svs_pack_weight_gather")
    ws = torch.empty(max(int(L.svs_enc_block_workspace_bytes(B, H, W, C, N)), 16), dtype=torch.uint8, device=dev)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    out = {"call": f"svs_enc_block_fwd B{B} {H}x{W} C{C} N{N}, error / sum|x w|"}
    for name, val in (("fp32_mfma", -1), ("mfma_split", 1)):
        _lib.tuning("MFMA_SPLIT", val)
        y = torch.empty((B, Ho, Wo, N), device=dev)
        _lib.check(L.svs_enc_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), N, N, 0,
                                       ws.data_ptr(), ws.numel(), S()), "svs_enc_block_fwd")
        err = (y.permute(0, 3, 1, 2).cpu().double() - want).abs() / mag
        out[name] = {"max": float(f"{err.max().item():.3e}"), "mean": float(f"{err.mean().item():.3e}")}
    _lib.tuning("MFMA_SPLIT", -1)
    return out


def eval_record(model, dev, B=16, steps=30, warmup=5):
    """BASELINE configs[1]: eval forward at B = 16, eager and replayed from a hipGraph (the forward is one graph of 12
    launches; the replay removes the host launch cost that an eager B=16 forward is bound by)."""
    H, W = 512, 128
    mix = torch.empty((B, 1, H, W), device=dev)
    voc = torch.empty_like(mix)
    _lib.check(_lib.lib().svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), B, H, W, 10_000, _lib.stream_ptr()), "svs_fill_tiles")
    was = model.training
    model.eval()
    out = {}
    with torch.no_grad():
        for _ in range(warmup):
            model(mix)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(mix)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        out["eager"] = {"ms": round(ms, 4), "tiles_per_s": round(B / ms * 1e3, 1),
                        "frac": round(B / ms * 1e3 * FWD_GFLOP_PER_TILE / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}
        try:
            replay = model.graphed_forward(mix)
            for _ in range(warmup):
                replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                replay()
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            out["graph"] = {"ms": round(ms, 4), "tiles_per_s": round(B / ms * 1e3, 1),
                            "frac": round(B / ms * 1e3 * FWD_GFLOP_PER_TILE / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4)}
        except Exception as e:                      # the graph path is an optimisation, never a correctness dependency
            out["graph_error"] = str(e)[:200]
        # the same forward on the bf16 MFMA (BASELINE configs[4]; reported separately: bf16 activations, fp32 accumulation)
        try:
            model.eval_precision = "fp32"
            ref = model(mix)
            model.eval_precision = "bf16"
            got = model(mix)
            for _ in range(warmup):
                model(mix)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                model(mix)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / steps
            out["bf16"] = {"ms": round(ms, 4), "tiles_per_s": round(B / ms * 1e3, 1),
                           "mask_mean_abs_diff_vs_fp32": float((got - ref).abs().mean()), "mask_max_abs_diff_vs_fp32": float((got - ref).abs().max())}
        except Exception as e:
            out["bf16_error"] = str(e)[:200]
        finally:
            model.eval_precision = "fp32"
    model.train(was)
    out["batch"] = B
    out["roofline_tiles_per_s"] = round(FP32_MFMA_PEAK_TFLOPS * 1e3 / FWD_GFLOP_PER_TILE, 1)
    return out


def full_objective_record(model, dev, B=64, steps=15, warmup=4):
    """The reference's FULL training objective (train.py:287-296: alpha_L1 * L1 + alpha_MR * MR-STFT of the re-synthesised
    waveforms, its default) as one train step at batch B: two `specific_istft`, the three-resolution STFT loss with gradient, the
    transposed iSTFT into the mask logit, then the same backward + Adam as the L1 step.  Not what `value` measures (BASELINE's
    workload is the L1-loss step); reported beside it."""
    from svs_unet_pytorch_amd.model import ALPHA_MR
    H, W = 512, 128
    mix = torch.empty((B, 1, H, W), device=dev)
    voc = torch.empty_like(mix)
    _lib.check(_lib.lib().svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), B, H, W, 40_000, _lib.stream_ptr()), "svs_fill_tiles")
    mph = (torch.rand((B, 1, H, W), device=dev) - 0.5) * 6.2831853
    vph = (torch.rand((B, 1, H, W), device=dev) - 0.5) * 6.2831853
    model.train()
    out = {}
    for name, kw in (("l1_only", {}), ("l1_plus_mrstft", dict(mix_phase=mph, voc_phase=vph, alpha_mr=ALPHA_MR))):
        for _ in range(warmup):
            model.train_step(mix, voc, loss_scale=ALPHA_L1, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model.train_step(mix, voc, loss_scale=ALPHA_L1, **kw)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        out[name] = {"ms_per_step": round(ms, 4), "tiles_per_s": round(B / ms * 1e3, 1)}
    out["batch"] = B
    out["mrstft_part_ms"] = round(out["l1_plus_mrstft"]["ms_per_step"] - out["l1_only"]["ms_per_step"], 4)
    model._ws.pop(("mr", B, W, 768), None)
    return out


def strong_record(model, dev, world, rank, grad_sync, global_batch=512, steps=6, warmup=2):
    """Train step at a FIXED global batch of 512 tiles split over the ranks (north_star's strong-scaling target is quoted on
    this): per-GPU batch 512 / world.  Rank-local timing (the caller reduces with MAX over ranks when world > 1)."""
    B = global_batch // world
    H, W = 512, 128
    mix = torch.empty((B, 1, H, W), device=dev)
    voc = torch.empty_like(mix)
    _lib.check(_lib.lib().svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), B, H, W, 20_000 + rank * B, _lib.stream_ptr()), "svs_fill_tiles")
    model.train()
    for _ in range(warmup):
        model.train_step(mix, voc, loss_scale=ALPHA_L1, grad_sync=grad_sync)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.train_step(mix, voc, loss_scale=ALPHA_L1, grad_sync=grad_sync)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    model._ws.clear()                                   # the B=512 workspace is 9.7 GB: give it back
    return {"global_batch": global_batch, "per_gpu_batch": B, "n_gpus": world, "ms_per_step": round(1e3 * el / steps, 4),
            "tiles_per_s": round(global_batch * steps / el, 1), "scaling": "strong"}


def launch_children(n_gpus, argv, worker=None, timeout=None):
    """`python bench.py --gpus N` without a launcher: start N fresh child ranks (one process per GPU; RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, exactly what `torch.distributed.run` would set)
    BEFORE this process has made any GPU call -- it never makes one -- forward rank 0's single JSON line to stdout and
    everything else to stderr, and return non-zero if any child failed (the others are then terminated by PID).  Nothing is
    re-exec'd: the children are ordinary subprocesses.  `worker` is the script the ranks run (this file; tests pass a stub)."""
    import socket
    import subprocess
    worker = worker or os.environ.get("SVS_BENCH_WORKER") or os.path.abspath(__file__)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this pool
        env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n_gpus)))
        procs.append(subprocess.Popen([sys.executable, worker, *argv], env=env, stdout=subprocess.PIPE if r == 0 else 2,
                                      text=True if r == 0 else None))
    deadline = None if timeout is None else time.monotonic() + timeout
    failed = None
    out0 = None
    pending = set(range(n_gpus))
    while pending and failed is None:
        for r in sorted(pending):
            if r == 0 and out0 is None:
                try:                                             # drain rank 0's pipe while waiting for it
                    out0, _ = procs[0].communicate(timeout=0.2)
                except subprocess.TimeoutExpired:
                    pass
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0:
                    failed = (r, rc)
        if deadline is not None and time.monotonic() > deadline:
            failed = (-1, 124)
        if pending and failed is None:
            time.sleep(0.05)
    if failed is not None:
        for r in pending:                                        # exact PIDs of the children started above
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    if out0 is None:
        try:
            out0, _ = procs[0].communicate(timeout=10)
        except subprocess.TimeoutExpired:
            procs[0].kill()
            out0, _ = procs[0].communicate()
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    records = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in records[-1:]:
            sys.stderr.write(ln + "\n")
    if failed is not None or not records:
        who = "timed out" if failed and failed[0] < 0 else (f"rank {failed[0]} exited with code {failed[1]}" if failed else "no result line from rank 0")
        sys.stderr.write(f"bench.py: {n_gpus}-rank run failed ({who})\n")
        return (failed[1] if failed else 1) or 1
    print(records[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=("train", "eval"), default="train")
    ap.add_argument("--batch", type=int, default=0, help="tiles per GPU (default 64 train / 16 eval)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layers", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the eval-B16 / strong-scaling / signal-kernel sub-records")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_children(args.gpus, sys.argv[1:]))       # no GPU call has been made in this process
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} of {world} sees no GPU (cuda:{local_rank} of {torch.cuda.device_count()} devices): "
                         "the benchmark runs the hand-written gfx950 kernels only, there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or (args.force_dist and "RANK" in os.environ)     # --force-dist: rehearse the RCCL path with 1 rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from svs_unet_pytorch_amd.parallel import init_process_group
        init_process_group(rank, world, dev)

    B = args.batch or (64 if args.mode == "train" else 16)
    H, W = 512, 128
    model = UNet()
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=False).items()})
    model.to(dev)
    model.rank = rank
    mix = torch.empty((B, 1, H, W), device=dev)
    voc = torch.empty_like(mix)
    _lib.check(_lib.lib().svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), B, H, W, rank * B, _lib.stream_ptr()), "svs_fill_tiles")

    grad_sync = None
    if args.mode == "train":
        model.train()
        if use_dist:
            from svs_unet_pytorch_amd.parallel import GradAllReduce
            grad_sync = GradAllReduce(model, dist.group.WORLD)
        step = lambda: model.train_step(mix, voc, loss_scale=ALPHA_L1, grad_sync=grad_sync)
    else:
        model.eval()

        def step():
            with torch.no_grad():
                return model(mix)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # strong-scaling point of north_star (outside the timed region above): global batch 512 split over the ranks, every rank
    # takes part, MAX over ranks
    strong = None
    if args.mode == "train" and not args.no_extras:
        fence()
        strong = strong_record(model, dev, world, rank, grad_sync)
        if dist is not None:
            t = torch.tensor([strong["ms_per_step"]], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            strong["ms_per_step"] = round(float(t.item()), 4)
            strong["tiles_per_s"] = round(strong["global_batch"] / strong["ms_per_step"] * 1e3, 1)

    if rank == 0:
        tiles_per_s = world * B * args.steps / elapsed
        gflop = TRAIN_GFLOP_PER_TILE if args.mode == "train" else FWD_GFLOP_PER_TILE
        res = {
            "metric": "spectrogram-tiles/sec (512x128) U-Net " + ("fwd+bwd (L1 train step incl. Adam" + (", RCCL grad all-reduce)" if world > 1 else ")") if args.mode == "train" else "eval forward"),
            "value": round(tiles_per_s, 1), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("train.py L1-loss step, 512x128 tiles, batch %d per GPU" % B) if args.mode == "train"
                       else ("U-Net eval forward, 512x128 tiles, batch %d per GPU" % B),
                       "global_batch": B * world, "parallelism": f"dp{world}", "mode": args.mode},
            "conv_roofline_frac": round(tiles_per_s / world * gflop / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
        }
        if strong is not None:
            res["strong_b512"] = strong
        if not args.no_layers:         # rank 0's GPU, any world size: the dominant kernel's roofline record
            calls = time_gemm_calls(B, args.mode)
            fam = {}
            for name, kernel, ks, ms, gf, _ in calls:
                e = fam.setdefault(kernel, [0.0, 0.0, 0])
                e[0] += ms; e[1] += gf; e[2] += 1
            dom = max(fam, key=lambda k: fam[k][0])
            # the dominant kernel ALONE (its fixed-order slab reduction is a separate kernel in the rocprofv3
            # summary): same calls again with the reduction skipped, so that avg_launch_ms is comparable with
            # the summary's average duration of that kernel
            _lib.tuning("SKIP_REDUCE", 1)
            alone = [c for c in time_gemm_calls(B, args.mode, only_kernel=dom)]
            _lib.tuning("SKIP_REDUCE", -1)
            ms, gf, cnt = sum(c[3] for c in alone), sum(c[4] for c in alone), len(alone)
            gf_in_image = sum(c[4] * c[5] for c in alone)
            ach = gf / ms            # GFLOP / ms = TFLOP/s
            res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(dom)[0], "traffic_source": pmc_traffic(dom)[1],
                               "launches_per_step": cnt, "avg_launch_ms": round(ms / cnt, 4),
                               "algorithmic_gflop_per_launch": round(gf / cnt, 3),
                               "achieved_in_image": round(gf_in_image / ms, 2),
                               "frac_in_image": round(gf_in_image / ms / FP32_MFMA_PEAK_TFLOPS, 4),
                               "note": "HIP events on the launch stream around each launch of this kernel in one step "
                                       "(slab reductions excluded); algorithmic FLOPs = 2*M*N*K of the layer (SURVEY.md 8d); "
                                       "*_in_image counts only products whose tap is inside the image -- the share the "
                                       "padding-skipping kernels cannot avoid executing is between the two"}
            res["kernels"] = {k: {"calls": v[2], "ms": round(v[0], 4), "tflops": round(v[1] / v[0], 2)} for k, v in fam.items()}
            res["layers"] = {name: {"kernel": kernel, "ksplit": ks, "ms": round(ms, 4), "tflops": round(gf / ms, 2),
                                    "in_image": round(fr, 3)} for name, kernel, ks, ms, gf, fr in calls}
        if world == 1:
            if not args.no_extras:
                if args.mode == "train":
                    res["optional_mfma_split"] = split_mode_record(model, dev, B)
                    res["eval_b16"] = eval_record(model, dev)                  # BASELINE configs[1]
                    try:
                        res["full_objective"] = full_objective_record(model, dev, B)     # train.py's default objective (L1 + MR-STFT)
                    except Exception as e:
                        res["full_objective_error"] = str(e)[:200]
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    from signal_bench import signal_record
                    res["signal"] = signal_record(240.0, 44100, 64)            # STFT / iSTFT GB/s, MR-STFT loss ms (configs[4] pieces)
                except Exception as e:
                    res["signal_error"] = str(e)[:200]
                if args.mode == "train":
                    # last of the GPU records: its float64 CPU convolution leaves the host's worker threads spinning for a
                    # while, which slows the launch-bound sub-records (eval forward at batch 16) if they come after it
                    try:
                        res["optional_mfma_split"]["accuracy_vs_float64"] = split_mode_accuracy(dev)
                    except Exception as e:               # a diagnostic, never a reason to lose the bench line
                        res["optional_mfma_split"]["accuracy_error"] = str(e)[:200]
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.mode)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
