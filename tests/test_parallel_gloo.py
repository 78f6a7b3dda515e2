"""CPU suite (-m "not gpu"): the N>1 path with world_size 2 over gloo.  The exchange logic of parallel.py
(one SUM all-reduce of the flat gradient buffer, 1/world folded into the optimiser's grad_scale, parameter
broadcast, rank folded into the dropout counter, shard ranges) is exercised on CPU tensors; the gradients
that get exchanged come from the oracle (the HIP kernels need a GPU), sharded exactly like bench.py /
train.py shard tiles: rank r owns tiles [r*B, (r+1)*B)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import unet_oracle as uo
from svs_unet_pytorch_amd import synth
from svs_unet_pytorch_amd.model import UNet
from svs_unet_pytorch_amd.parallel import GradAllReduce, average_bn_buffers, broadcast_parameters, shard_range

H, W, B_PER_RANK = 64, 16, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_grads(rank):
    """Oracle gradients of rank's shard (per-shard BatchNorm statistics, like DistributedDataParallel)."""
    mix, voc = synth.tiles(B_PER_RANK, H, W, first_tile=rank * B_PER_RANK)
    st = uo.to_torch_state(synth.closed_form_state(trained_stats=False))
    masks = [torch.from_numpy(m) for m in synth.dropout_masks(B_PER_RANK, seed=4242, step=0, rank=rank)]
    loss, grads = uo.train_step(st, uo.new_adam_state(st), torch.from_numpy(mix), torch.from_numpy(voc), dropout_masks=masks,
                                apply_update=False)
    return loss, torch.cat([grads[k].reshape(-1) for k in uo.param_keys(st)])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = UNet()
        if rank == 0:
            model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=False).items()})
        broadcast_parameters(model, 0)
        sync = GradAllReduce(model)
        assert model.rank == rank and model.optim.grad_scale == 1.0 / world
        _, g = _shard_grads(rank)
        out[f"g{rank}"] = g.clone()
        model._gflat.copy_(g)
        # the overlapped form: two asynchronous pieces (decoder half first), as UNet.fwd_bwd_overlapped issues them
        split = sum(n for _, n in model._param_spans[:24])
        assert sync.overlap
        handles = [sync.reduce_async(model._gflat[split:]), sync.reduce_async(model._gflat[:split])]
        for h in handles:
            h.wait()
        # running statistics: every rank accumulates its own; before use they are averaged over the ranks
        model._bn_flat.fill_(float(rank + 1))
        epoch = model._param_epoch
        average_bn_buffers(model)
        assert model._param_epoch == epoch + 1 and torch.all(model._bn_flat == 1.5)
        if rank == 0:
            out["flat"] = model._flat.clone()
            out["summed"] = model._gflat.clone()
            out["scale"] = model.optim.grad_scale
        else:
            out["flat1"] = model._flat.clone()
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_gradient_exchange():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    want_params = torch.cat([torch.from_numpy(np.array(v)).reshape(-1) for k, v in synth.closed_form_state(trained_stats=False).items()
                             if not ("running_" in k or "num_batches" in k)])
    assert torch.equal(out["flat"], want_params) and torch.equal(out["flat1"], want_params)     # broadcast reached rank 1
    g0, g1 = out["g0"], out["g1"]
    assert not torch.equal(g0, g1)                                                             # shards see different tiles/masks
    assert torch.equal(out["summed"], g0 + g1)                                                 # one SUM all-reduce, nothing else
    mean = out["summed"] * out["scale"]
    assert torch.allclose(mean, (g0 + g1) / 2, rtol=1e-6, atol=1e-12)
    # and the shard gradients are the oracle's (recomputed here; fp32 reduction order differs with the thread
    # count, so only to the noise level of this network's fp32 gradients)
    ref0 = _shard_grads(0)[1]
    assert (ref0 - g0).norm() <= 2e-2 * ref0.norm()


def test_shards_cover_the_global_batch():
    world, global_batch = 8, 512
    covered = []
    for r in range(world):
        b, e = shard_range(global_batch, r, world)
        assert e - b == 64                                  # BASELINE config 4: 64 tiles per GPU
        covered += list(range(b, e))
    assert covered == list(range(global_batch))
