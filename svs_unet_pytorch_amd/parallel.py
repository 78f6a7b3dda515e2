"""Data parallelism over the GPUs of one node: one process per GPU, tiles sharded over the batch
(tiles are independent; the training step has exactly one exchange, the gradient sum).

The reference is single-process (it has no distributed code at all), so the semantics are chosen here:
  * gradients: all-reduce (SUM, fp32) of the model's flat 9,823,313-element gradient buffer over RCCL/xGMI
    (`backend="nccl"` is RCCL on ROCm) in FOUR pieces that follow the backward pass (`UNet.fwd_bwd_overlapped`):
    the decoder blocks (tensors 24..45, 21.8 MB, produced first), the conv6 block (13.1 MB), conv5 + conv4 (4.1 MB)
    and conv3..conv1 (0.26 MB); each piece is reduced on RCCL's stream while the next one is still being computed,
    so only the last, latency-sized piece is exchanged after the backward has ended.  xGMI is point-to-point, so few
    large messages beat many small buckets; four is what the dependency structure of the backward pass offers.  The
    1/world mean is folded into the fused Adam kernel (`FusedAdam.grad_scale`): no extra pass over the gradients;
  * BatchNorm: per-GPU batch statistics (what DistributedDataParallel over the reference would do); every rank
    accumulates its own running statistics during an epoch and `average_bn_buffers` takes their mean over the ranks
    (one all-reduce of 2,016 floats) before every validation pass / checkpoint, so what is saved does not depend on
    which rank writes it;
  * Dropout2d: the rank is folded into the mask counter, so shards draw independent masks;
  * parameters: identical on every rank after `broadcast_parameters`, and they stay identical because
    every rank applies the same Adam update to the same summed gradient.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """[begin, end) of the contiguous share of `n_items` units owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class GradAllReduce:
    """`grad_sync` hook for `UNet.train_step`.  `overlap=True` (default): `reduce_async(slice)` is called
    right after each piece of the backward is enqueued; `overlap=False`: `__call__(flat)` after the backward."""

    def __init__(self, model, group=None, overlap: bool = True):
        self.group = group
        self.world = dist.get_world_size(group)
        self.overlap = overlap
        model.optim.grad_scale = 1.0 / self.world
        model.rank = dist.get_rank(group)

    def reduce_async(self, grad_slice: torch.Tensor):
        """Starts the SUM all-reduce of a contiguous slice of the flat gradient buffer; returns the work handle.
        RCCL waits (on its own stream) for the kernels already enqueued on the current stream, and `wait()` makes
        the current stream wait for the reduction -- neither blocks the host."""
        return dist.all_reduce(grad_slice, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def __call__(self, flat_grad: torch.Tensor):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)


def broadcast_parameters(model, src: int = 0, group=None):
    """Make every rank start from rank `src`'s parameters and BatchNorm buffers."""
    dist.broadcast(model._flat, src=src, group=group)
    dist.broadcast(model._bn_flat, src=src, group=group)
    dist.broadcast(model._nbt_flat, src=src, group=group)
    model._param_epoch += 1


def average_bn_buffers(model, group=None):
    """BatchNorm running statistics under data parallelism: every rank normalises with its OWN batch statistics (what
    DistributedDataParallel over the reference would do) and so accumulates its own running mean / variance; before they are
    used or saved (validation, checkpoint) the ranks take the mean of the per-rank buffers -- one all-reduce of 2 x 1,008
    floats.  `num_batches_tracked` is the same on every rank already."""
    world = dist.get_world_size(group)
    if world == 1:
        return
    dist.all_reduce(model._bn_flat, op=dist.ReduceOp.SUM, group=group)
    model._bn_flat.mul_(1.0 / world)
    model._param_epoch += 1


def init_process_group(rank: int, world: int, device: torch.device):
    """RCCL process group with its collectives on a HIGH-priority stream: the gradient exchange runs beside two saturated
    compute streams (the backward's own and the library's weight-gradient stream, itself high priority) and must not wait
    for free CUs behind them."""
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, pg_options=opts)

