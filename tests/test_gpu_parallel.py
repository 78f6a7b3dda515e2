"""-m gpu: the data-parallel train step end to end with TWO ranks on the one GPU of the box (gloo carries the exchange,
since RCCL refuses two ranks on one device; the driver's 8-GPU run uses RCCL through the same code).  Each rank runs the
real `UNet.train_step(grad_sync=GradAllReduce(...))`: forward + loss, the three backward parts with an asynchronous
all-reduce after each (issued from the auxiliary stream behind `svs_unet_train_bwd_sync`), Adam with 1/world folded in.
The parent process emulates the same two steps in one process -- both shards' `fwd_bwd`, gradients summed, Adam with
grad_scale 1/2 -- and the parameters must agree bit for bit; so must the two ranks with each other."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from svs_unet_pytorch_amd import synth
from svs_unet_pytorch_amd.model import UNet

pytestmark = pytest.mark.gpu
B, STEPS, SCALE = 8, 2, 166.66


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fresh_model():
    m = UNet()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=False).items()})
    return m.to("cuda").train()


def _shard(rank):
    mix, voc = synth.tiles(B, first_tile=4000 + rank * B)
    return torch.from_numpy(mix).to("cuda"), torch.from_numpy(voc).to("cuda")


def _extras(rank, full):
    """keyword arguments of the full objective (train.py:287-296): phase tiles as angles + the MR-STFT weight"""
    if not full:
        return {}
    n = B * 512 * 128
    ang = lambda seed: torch.from_numpy((synth.uniform(seed + rank, n) * 2 * np.pi - np.pi).astype(np.float32).reshape(B, 1, 512, 128)).to("cuda")
    return dict(mix_phase=ang(70), voc_phase=ang(80), alpha_mr=0.66)


def _worker(rank, world, port, out, full):
    import torch.distributed as dist
    from svs_unet_pytorch_amd.parallel import GradAllReduce, broadcast_parameters
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _fresh_model()
        broadcast_parameters(model, 0)
        sync = GradAllReduce(model, overlap=not os.environ.get("SVS_TEST_NO_OVERLAP"))     # (tools/stress_2rank.py: exchange after the backward)
        assert model.rank == rank and model.optim.grad_scale == 0.5
        mix, voc = _shard(rank)
        extra = _extras(rank, full)
        losses = [model.train_step(mix, voc, loss_scale=SCALE, grad_sync=sync, **extra).item() for _ in range(STEPS)]
        torch.cuda.synchronize()
        out.put((rank, losses, model._flat.cpu().numpy(), model._bn_flat.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("full", [False, True], ids=["l1", "l1+mrstft"])
def test_two_rank_overlapped_step_matches_single_process(full, report):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, full)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, losses, flat, bn = out.get(timeout=600)
        got[rank] = (losses, flat, bn)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(got[0][1], got[1][1]), "ranks diverged"

    # the same two steps in one process: rank r's model sees shard r (its own BatchNorm statistics and dropout stream)
    models = [_fresh_model() for _ in range(2)]
    shards = [_shard(r) for r in range(2)]
    extras = [_extras(r, full) for r in range(2)]
    for r, m in enumerate(models):
        m.rank = r
        m.optim.grad_scale = 0.5
    losses = [[], []]
    for _ in range(STEPS):
        for r, m in enumerate(models):
            m.optim.zero_grad()
            losses[r].append(m.fwd_bwd(*shards[r], loss_scale=SCALE, **extras[r]).item())
        total = models[0]._gflat + models[1]._gflat
        for m in models:
            m._gflat.copy_(total)
            m.optim.step()
    torch.cuda.synchronize()
    for r in range(2):
        assert got[r][0] == losses[r], (got[r][0], losses[r])
        mine = models[r]._flat.cpu().numpy()
        if not np.array_equal(got[r][1], mine):                       # which tensors, how many elements, how far apart
            from svs_unet_pytorch_amd import _lib
            offs = [int(_lib.lib().svs_unet_param_offset(i)) for i in range(47)]
            bad = np.nonzero(got[r][1] != mine)[0]
            tens = sorted(set(int(np.searchsorted(offs, b, side="right") - 1) for b in bad))
            raise AssertionError(f"rank {r} parameters: {bad.size} elements differ, tensors {tens}, max |d| {np.abs(got[r][1] - mine).max():.3e}")
        assert np.array_equal(got[r][2], models[r]._bn_flat.cpu().numpy()), f"rank {r} BatchNorm buffers"
    report(f"two ranks (gloo) vs single-process emulation ({'full objective' if full else 'L1'}): parameters bit-identical", 0.0, 0.0)
