#!/usr/bin/env python3
"""Where does the data-parallel step lose time against the single-process step?  Times train_step at batch 64 (a) fused,
(b) split into parts with a no-op exchange hook, (c) split with a real 1-rank RCCL all-reduce (run under torchrun).

    python tools/split_overhead.py            # (a), (b)
    torchrun --nproc-per-node 1 tools/split_overhead.py --dist   # (a), (b), (c)
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svs_unet_pytorch_amd import synth  # noqa: E402
from svs_unet_pytorch_amd.model import UNet  # noqa: E402


class NoSync:
    overlap = True

    def reduce_async(self, sl):
        class H:
            def wait(self):
                return None
        return H()


def timeit(fn, steps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, t_enq / steps * 1e3


def main():
    dist_mode = "--dist" in sys.argv
    dev = "cuda"
    if "--dist-first" in sys.argv:          # initialise RCCL before anything else touches the GPU, as bench.py does
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    B = 64
    model = UNet().to(dev).train()
    mix = torch.rand((B, 1, 512, 128), device=dev)
    voc = mix * 0.5
    print("fused            ms/step %.3f (host enqueue %.3f)" % timeit(lambda: model.train_step(mix, voc, 166.66)))
    ns = NoSync()
    print("split, no-op hook ms/step %.3f (host enqueue %.3f)" % timeit(lambda: model.train_step(mix, voc, 166.66, grad_sync=ns)))
    if dist_mode:
        import torch.distributed as dist
        from svs_unet_pytorch_amd.parallel import GradAllReduce
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
        gs = GradAllReduce(model)
        print("split, RCCL 1 rank ms/step %.3f (host enqueue %.3f)" % timeit(lambda: model.train_step(mix, voc, 166.66, grad_sync=gs)))
        gs.overlap = False
        print("fused + RCCL after ms/step %.3f (host enqueue %.3f)" % timeit(lambda: model.train_step(mix, voc, 166.66, grad_sync=gs)))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
