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
        sync = GradAllReduce(model, overlap=not os.environ.get("SVS_TEST_NO_OVERLAP"))     # (tools/attic/stress_2rank.py: exchange after the backward)
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


class _EmulatedAllReduce:
    """`grad_sync` hook that plays the part of the other ranks in ONE process: `reduce_async(slice)` -- called by
    `UNet.fwd_bwd_overlapped` on its exchange stream right after each backward part -- adds the other shards' gradients of that
    slice in rank order (what a SUM all-reduce leaves behind) and returns a handle whose wait() orders the caller's stream
    behind it, like a torch.distributed work handle."""
    overlap = True

    def __init__(self, model, others):
        self.model, self.others = model, others
        self.world = len(others) + 1
        self.slices = []
        model.optim.grad_scale = 1.0 / self.world

    def reduce_async(self, sl):
        off = (sl.data_ptr() - self.model._gflat.data_ptr()) // 4
        self.slices.append((off, sl.numel()))
        for g in self.others:
            sl.add_(g[off:off + sl.numel()])
        ev = torch.cuda.Event()
        ev.record()

        class Handle:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(ev)
        return Handle()


def test_eight_shard_step_emulated_on_one_gpu(report):
    """BASELINE configs[3] (global batch 512 = 8 ranks x 64 tiles) on the one GPU of the box: ranks 0..7 one after the other
    (rank folded into the dropout counter, per-shard BatchNorm statistics), rank 0's step through the real overlapped path
    (`fwd_bwd_overlapped`: four backward parts, an exchange after each on the auxiliary stream) with the other seven shards'
    gradients standing in for the all-reduce, Adam with grad_scale 1/8 -- bit-identical to summing the eight `fwd_bwd`
    gradients and taking one Adam step; the four exchanged slices tile the flat gradient buffer exactly."""
    world, Bs = 8, 64
    def shard(r):
        mix, voc = synth.tiles(Bs, first_tile=r * Bs)
        return torch.from_numpy(mix).to("cuda"), torch.from_numpy(voc).to("cuda")
    helper = _fresh_model()
    grads, losses = [], []
    for r in range(world):
        helper.rank, helper.dropout_step = r, 0
        helper.optim.zero_grad()
        losses.append(helper.fwd_bwd(*shard(r), loss_scale=SCALE).item())
        grads.append(helper._gflat.clone())
    assert len(set(losses)) == world, "shards must see different tiles / dropout masks"
    # the plain emulation: sum in rank order, one Adam step with 1/8
    plain = _fresh_model()
    plain.rank = 0
    plain.optim.grad_scale = 1.0 / world
    plain.optim.zero_grad()
    l_plain = plain.fwd_bwd(*shard(0), loss_scale=SCALE).item()
    for g in grads[1:]:
        plain._gflat.add_(g)
    plain.optim.step()
    # rank 0 through the overlapped path
    m = _fresh_model()
    m.rank = 0
    sync = _EmulatedAllReduce(m, grads[1:])
    l_over = m.train_step(*shard(0), loss_scale=SCALE, grad_sync=sync).item()
    torch.cuda.synchronize()
    assert l_over == l_plain == losses[0]
    spans = sorted(sync.slices)
    assert spans[0][0] == 0 and all(a + n == b for (a, n), (b, _) in zip(spans, spans[1:])) and sum(n for _, n in spans) == m._gflat.numel()
    assert torch.equal(m._gflat, plain._gflat), "summed gradients differ"
    assert torch.equal(m._flat, plain._flat) and torch.equal(m._bn_flat, plain._bn_flat)
    # the mean gradient has the scale of one shard's (sanity of the 1/8 fold)
    ratio = (m._gflat.norm() / world / grads[0].norm()).item()
    assert 0.2 < ratio < 1.5, ratio
    report("8-shard emulation (64 tiles each) vs overlapped step: parameters bit-identical", 0.0, 0.0)


def _rccl_worker(port, out, steps, batch):
    """One rank, REAL RCCL process group (`parallel.init_process_group`: nccl backend, high-priority stream): the path
    bench.py --force-dist and every multi-GPU rank runs."""
    import torch.distributed as dist
    from svs_unet_pytorch_amd.parallel import GradAllReduce, average_bn_buffers, broadcast_parameters, init_process_group
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    init_process_group(0, 1, dev)
    try:
        assert dist.get_backend() == "nccl"
        model = _fresh_model()
        broadcast_parameters(model, 0)
        sync = GradAllReduce(model, dist.group.WORLD)
        assert sync.overlap and model.optim.grad_scale == 1.0
        mix, voc = synth.tiles(batch, first_tile=9000)
        mix, voc = torch.from_numpy(mix).to(dev), torch.from_numpy(voc).to(dev)
        losses = [model.train_step(mix, voc, loss_scale=SCALE, grad_sync=sync).item() for _ in range(steps)]
        average_bn_buffers(model)
        torch.cuda.synchronize()
        out.put((losses, model._flat.cpu().numpy(), model._bn_flat.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_overlapped_step_is_the_plain_step(report):
    """The RCCL code path under test (world 1, child process): process group on a high-priority stream, parameter broadcast,
    the overlapped train step with four asynchronous RCCL all-reduces issued from the auxiliary stream behind
    `svs_unet_train_bwd_sync`, work-handle waits, Adam -- bit-identical to the plain single-process step."""
    steps, batch = 3, 16
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), out, steps, batch))
    p.start()
    losses, flat, bn = out.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    m = _fresh_model()
    mix, voc = synth.tiles(batch, first_tile=9000)
    mix, voc = torch.from_numpy(mix).to("cuda"), torch.from_numpy(voc).to("cuda")
    mine = [m.train_step(mix, voc, loss_scale=SCALE).item() for _ in range(steps)]
    torch.cuda.synchronize()
    assert mine == losses, (mine, losses)
    assert np.array_equal(flat, m._flat.cpu().numpy()), f"max |d| {np.abs(flat - m._flat.cpu().numpy()).max():.3e}"
    assert np.array_equal(bn, m._bn_flat.cpu().numpy())
    report("1-rank RCCL overlapped step vs plain step: parameters bit-identical", 0.0, 0.0)
