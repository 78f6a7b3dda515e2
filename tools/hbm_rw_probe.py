#!/usr/bin/env python3
"""Calibrates what a write-heavy / read-heavy / copy stream reaches on this device (torch's own fill, copy and sum kernels),
to put the write-dominated kernels (conv1 forward, deconv6 backward-data: 4 bytes in, 64-128 out per pixel) in context."""
import torch

def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us

for mb in (64, 128, 256, 1024):
    n = mb * (1 << 20) // 4
    x = torch.rand(n, device="cuda")
    y = torch.empty_like(x)
    t_fill = timeit(lambda: y.fill_(1.5))
    t_copy = timeit(lambda: y.copy_(x))
    t_sum = timeit(lambda: x.sum())
    t_axpy = timeit(lambda: y.add_(x))
    print(f"{mb:5d} MiB: fill {t_fill:7.1f} us = {mb * 1.048576 / t_fill:5.2f} TB/s written | sum {t_sum:7.1f} us = {mb * 1.048576 / t_sum:5.2f} TB/s read | "
          f"copy {t_copy:7.1f} us = {2 * mb * 1.048576 / t_copy:5.2f} TB/s r+w | y+=x {t_axpy:7.1f} us = {3 * mb * 1.048576 / t_axpy:5.2f} TB/s 2r+w")
