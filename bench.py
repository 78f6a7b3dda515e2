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
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes kept under profiles/ (FETCH_SIZE doubled per
    the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as is); None when no summary covers the kernel."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(mode, seconds_budget=20.0):
    """The CPU oracle (oracle/unet_oracle.py) on this box's host cores: same synthetic tiles, fp32."""
    from oracle import unet_oracle as uo
    # the box gives one GPU's job a 16-CPU share, whatever os.cpu_count() says
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    B = 16
    mix_np, voc_np = synth.tiles(B)
    mix, voc = torch.from_numpy(mix_np), torch.from_numpy(voc_np)
    st = uo.to_torch_state(synth.closed_form_state(trained_stats=False))
    if mode == "train":
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
        if el > seconds_budget or it >= 40:
            break
    return {"value": round(B * it / el, 2), "unit": "tiles/s", "cores": cores, "kind": "port",
            "sample": f"{it} {'L1 train steps (fwd+bwd+Adam)' if mode == 'train' else 'eval forwards'} of batch {B}, "
                      f"fp32 torch CPU oracle, {cores} threads, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=("train", "eval"), default="train")
    ap.add_argument("--batch", type=int, default=0, help="tiles per GPU (default 64 train / 16 eval)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-layers", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
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
        if world == 1:
            if not args.no_layers:
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
                                   "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(dom),
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
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(args.mode)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
