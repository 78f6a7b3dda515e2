#!/usr/bin/env python3
"""Two processes on one GPU, each repeating the SAME train step (no exchange) and comparing every gradient with its first one.
    python tools/stress_local.py [reps=40] [full=1] [procs=2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.multiprocessing as mp

def worker(rank, reps, full, out):
    from svs_unet_pytorch_amd import _lib, synth
    from svs_unet_pytorch_amd.model import UNet
    torch.cuda.set_device(0)
    B = 8
    m = UNet()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state(trained_stats=False).items()})
    m.to("cuda").train()
    mix, voc = synth.tiles(B, first_tile=4000 + rank * B)
    mix, voc = torch.from_numpy(mix).cuda(), torch.from_numpy(voc).cuda()
    n = B * 512 * 128
    ang = lambda seed: torch.from_numpy((synth.uniform(seed + rank, n) * 2 * np.pi - np.pi).astype(np.float32).reshape(B, 1, 512, 128)).cuda()
    ex = dict(mix_phase=ang(70), voc_phase=ang(80), alpha_mr=0.66) if full else {}
    masks = [torch.from_numpy(x) for x in synth.dropout_masks(B, seed=5, step=0)]
    offs = [int(_lib.lib().svs_unet_param_offset(i)) for i in range(47)]
    ref = None; bad_runs = []
    for i in range(reps):
        m.set_dropout_masks(masks)
        m.optim.zero_grad()
        loss = m.fwd_bwd(mix, voc, loss_scale=166.66, **ex)
        torch.cuda.synchronize()
        g = m._gflat.clone()
        rec = (loss.item(), None if m.last_mr_loss is None else m.last_mr_loss.item())
        snaps = {k: w.clone() for k, w in m._ws.items()}
        if ref is None: ref = (g, rec); ref_snaps = snaps
        elif not torch.equal(g, ref[0]) or rec != ref[1]:
            for k, w in snaps.items():                      # where do the workspaces differ?  (4-byte words; first / last / count)
                a, b = w.view(torch.int32), ref_snaps[k].view(torch.int32)
                d = (a != b).nonzero().flatten()
                if d.numel():
                    segs = d.cpu().numpy()
                    cuts = np.nonzero(np.diff(segs) > 65536)[0]
                    starts = [int(segs[0])] + [int(segs[c + 1]) for c in cuts]
                    ends = [int(segs[c]) for c in cuts] + [int(segs[-1])]
                    print(f"rank {rank} rep {i} workspace {k}: {d.numel()} words differ of {a.numel()}; ranges {list(zip(starts, ends))[:12]}", flush=True)
            bad = (g != ref[0]).nonzero().flatten().cpu().numpy()
            tens = sorted(set(int(np.searchsorted(offs, b, side="right") - 1) for b in bad[::max(1, bad.size // 5000)]))
            bad_runs.append((i, int(bad.size), tens, rec, ref[1], float((g - ref[0]).abs().max())))
    out.put((rank, bad_runs))

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    full = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    procs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, reps, full, out)) for r in range(procs)]
    for p in ps: p.start()
    for _ in ps:
        rank, bad = out.get(timeout=900)
        print(f"rank {rank}: {len(bad)} of {reps - 1} repetitions differ from the first")
        for b in bad[:6]: print("   ", b)
    for p in ps: p.join(timeout=60)
