"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden-vector generator, part 2; runs ONLY in the build container.

Pins the pieces of the reference's train.py that gen_golden.py did not reach (SURVEY.md 8c golden (6),
VERDICT round 1 "Next" #3):

  * `specific_istft` (train.py:33-60)            -> tests/golden/specific_istft.npz
  * `SpectrogramDataset.__getitem__` (65-143)    -> tests/golden/dataset_items.npz
  * the L1 train step of train.py:274-283 at the BASELINE batch size B = 64 (configs[2]), run on the reference's
    own model in float64 and float32               -> tests/golden/train_b64.npz

train.py is executed as the script it is (`runpy.run_path`, `--label x --epoch 0 --train_folder <tmp>`), with
`auraloss.freq.MultiResolutionSTFTLoss` satisfied by a stub class (the real package is not installable here; the
stub is only constructed at train.py:26, never called because the epoch loop is empty).  The globals the run
returns hold the reference's own `SpectrogramDataset` and `specific_istft`.  Nothing of the reference's source
is stored: fixtures are seeds, integer starts, hashes and output arrays.

Usage (build container):  python oracle/gen_golden_train.py [--skip-b64] [--batches=128,256,512] [--only-batches]
"""
from __future__ import annotations

import hashlib
import os
import random
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from svs_unet_pytorch_amd import synth  # noqa: E402
from oracle import stft_oracle as so  # noqa: E402
from oracle import tiling_oracle as to  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from oracle.gen_golden import InjectedDropout, import_reference_model, ref_model_with, sample, stats  # noqa: E402

DATASET_LENGTHS = (300, 128, 50, 129)      # T > L (random start), T == L (pad 0), T < L (pad), T == L + 1 (start in {0, 1})
DATASET_SEEDS = (0, 1, 7)


synthetic_song = synth.song


def digest(a: np.ndarray) -> np.ndarray:
    """sha256 of the array's bytes as 32 uint8 (bit-exact pin of a float32 tile in 32 bytes)."""
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8).copy()


def run_reference_train_script(train_folder: str, work: str):
    """Executes /root/reference/train.py with an empty epoch loop and returns its globals."""
    aura = types.ModuleType("auraloss")
    freq = types.ModuleType("auraloss.freq")

    class MultiResolutionSTFTLoss:          # constructed at train.py:26, never called here
        def __init__(self, *a, **k):
            pass

    freq.MultiResolutionSTFTLoss = MultiResolutionSTFTLoss
    aura.freq = freq
    sys.modules["auraloss"] = aura
    sys.modules["auraloss.freq"] = freq
    if REF not in sys.path:
        sys.path.insert(0, REF)
    argv, cwd = sys.argv, os.getcwd()
    try:
        os.chdir(work)
        sys.argv = ["train.py", "--label", "x", "--epoch", "0", "--train_folder", train_folder,
                    "--valid_folder", os.path.join(work, "no_such_folder"), "--load_path", os.path.join(work, "none.pth")]
        return runpy.run_path(os.path.join(REF, "train.py"), run_name="__main__")
    finally:
        sys.argv = argv
        os.chdir(cwd)


def train_step_golden(B, mask_seed=64):
    """One L1 train step (train.py:274-283) of the reference's own model at batch B in float64 and float32, default
    planner inputs (synth.tiles(B), closed-form fresh state, injected dropout masks) -> tests/golden/train_b{B}.npz."""
    import resource
    import time
    model_mod = import_reference_model()
    mix_np, voc_np = synth.tiles(B)
    fresh_np = synth.closed_form_state(trained_stats=False)
    masks_np = synth.dropout_masks(B, seed=mask_seed, step=0)
    g = {}
    for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        t0 = time.time()
        mix_t, voc_t = torch.from_numpy(mix_np).to(dt), torch.from_numpy(voc_np).to(dt)
        refm = ref_model_with(fresh_np, model_mod).to(dt).train()
        refm.crit = torch.nn.L1Loss()
        opt = torch.optim.Adam(refm.parameters(), lr=1e-3)
        masks_t = [torch.from_numpy(m).to(dt) for m in masks_np]
        for i in range(5):
            getattr(refm, f"deconv{i + 1}_BAD")[2] = InjectedDropout(masks_t[i])
        opt.zero_grad()
        mask = refm(mix_t)                                                     # train.py:274-283, L1 terms
        loss = refm.crit(mask * mix_t, voc_t) + refm.crit((1 - mask) * mix_t, torch.clamp(mix_t - voc_t, min=0.0))
        loss.backward()
        names = [n for n, _ in refm.named_parameters()]
        grads = {n: p.grad.detach().clone() for n, p in refm.named_parameters()}
        opt.step()
        sd = refm.state_dict()
        if dt is torch.float64 and B <= 64:
            st_o = uo.to_torch_state(fresh_np, dt)
            lo, grads_o = uo.train_step(st_o, uo.new_adam_state(st_o), mix_t, voc_t, dropout_masks=masks_t)
            assert abs(lo - loss.item()) <= 1e-12, (lo, loss.item())
            for n in names:
                e = (grads_o[n] - grads[n]).norm().item()
                assert e <= 1e-9 * max(grads[n].norm().item(), 1e-3), f"oracle grad {n} drifted at B={B}: {e}"
        p = dt_name + "."
        g[p + "loss"] = np.array(loss.item(), np.float64)
        g[p + "mask_stats"] = stats(mask)
        g[p + "grad_norm"] = np.array([grads[n].double().norm().item() for n in names], np.float64)
        g[p + "grad_sum"] = np.array([grads[n].double().sum().item() for n in names], np.float64)
        for n in names:
            g[p + "grad_sample." + n] = sample(grads[n], 64)
        if dt is torch.float64:
            for k in sd:
                if "running_" in k:
                    g[p + "buf." + k] = sd[k].to(torch.float32).numpy().copy()
            for n in ("conv1.0.weight", "conv6.0.weight", "deconv1.weight", "deconv6.weight"):
                g[p + "param_after." + n] = sample(sd[n], 64)
        print(f"B={B} {dt_name}: loss {loss.item():.9f}  ({time.time() - t0:.0f} s, peak RSS "
              f"{resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB)", flush=True)
        del refm, opt, mask, loss, grads, sd, mix_t, voc_t
    g["param_names"] = np.array(names)
    g["mask_seed"] = np.array(mask_seed, np.int64)
    np.savez_compressed(os.path.join(OUT, f"train_b{B}.npz"), **g)


def batches_from_argv():
    for arg in sys.argv:
        if arg.startswith("--batches="):
            return [int(b) for b in arg.split("=", 1)[1].split(",")]
    return []


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--only-batches" in sys.argv:                 # leave the other fixtures as they are
        for b in batches_from_argv():
            train_step_golden(b, mask_seed=b)
        return
    work = tempfile.mkdtemp(prefix="svs_golden_train_")
    try:
        folder = os.path.join(work, "train")
        os.makedirs(os.path.join(folder, "mixture"))
        os.makedirs(os.path.join(folder, "vocal"))
        songs = {}
        for n, T in enumerate(DATASET_LENGTHS):
            mix, voc, pm, pv = synthetic_song(n, T)
            songs[n] = (mix, voc, pm, pv)
            base = f"{n:04d}_len{T}"
            np.save(os.path.join(folder, "mixture", base + "_spec.npy"), mix)
            np.save(os.path.join(folder, "vocal", base + "_spec.npy"), voc)
            np.save(os.path.join(folder, "mixture", base + "_phase.npy"), pm)
            np.save(os.path.join(folder, "vocal", base + "_phase.npy"), pv)
        G = run_reference_train_script(folder, work)
        RefDataset, ref_specific_istft = G["SpectrogramDataset"], G["specific_istft"]
        assert G["INPUT_LEN"] == 128 and G["HOP_SIZE"] == 768 and G["WINDOW_SIZE"] == 1024

        # ------------------------------------------------------------ SpectrogramDataset.__getitem__ (train.py:86-143)
        ds = RefDataset(folder)
        assert len(ds) == len(DATASET_LENGTHS) * 64 and len(ds.file_names) == len(DATASET_LENGTHS)
        g = {"lengths": np.array(DATASET_LENGTHS, np.int64), "seeds": np.array(DATASET_SEEDS, np.int64),
             "len": np.array(len(ds), np.int64)}
        for seed in DATASET_SEEDS:
            for idx in (0, 1, 2, 3, 5, 6):                       # idx % 4 -> song; 5, 6 exercise the wrap-around
                n = idx % len(DATASET_LENGTHS)
                T = DATASET_LENGTHS[n]
                random.seed(seed * 1000 + idx)
                mix_t, voc_t, mph_t, vph_t = ds[idx]
                assert mix_t.shape == voc_t.shape == mph_t.shape == vph_t.shape == (1, 512, 128) and mix_t.dtype == torch.float32
                # what start did the reference draw?  (one random.randint(0, T - 128) when T > 128, train.py:121)
                random.seed(seed * 1000 + idx)
                start = random.randint(0, T - 128) if T > 128 else 0
                mix, voc, pm, pv = songs[n]
                want_mix, want_voc = to.crop_item(mix, voc, start)
                assert np.array_equal(want_mix, mix_t.numpy()) and np.array_equal(want_voc, voc_t.numpy()), \
                    "oracle crop_item drifted from the reference's SpectrogramDataset"
                want_ph = to.crop_phase(pm, start)
                assert np.array_equal(want_ph, mph_t.numpy()), "oracle crop_phase drifted from the reference"
                assert np.array_equal(to.crop_phase(pv, start), vph_t.numpy())
                p = f"s{seed}.i{idx}."
                g[p + "start"] = np.array(start, np.int64)
                g[p + "mix_sha"] = digest(mix_t.numpy())
                g[p + "voc_sha"] = digest(voc_t.numpy())
                g[p + "mix_phase_sha"] = digest(mph_t.numpy())
                g[p + "voc_phase_sha"] = digest(vph_t.numpy())
                g[p + "mix_sum"] = np.array(mix_t.double().sum().item())
                g[p + "mix_phase_sample"] = sample(mph_t, 64)
        np.savez_compressed(os.path.join(OUT, "dataset_items.npz"), **g)

        # ------------------------------------------------------------ specific_istft (train.py:33-60)
        T = 128
        mag = synth.uniform(3, 2 * 512 * T).reshape(2, 1, 512, T)
        ang = (synth.uniform(4, 2 * 512 * T) * 2 * np.pi - np.pi).astype(np.float32).reshape(2, 1, 512, T)
        with torch.no_grad():
            wav = ref_specific_istft(torch.from_numpy(mag), torch.from_numpy(ang))
        assert wav.shape == (2, 1, 97536) and wav.dtype == torch.float32
        mine = so.specific_istft(mag, ang)
        e = np.abs(mine - wav.numpy())[..., 1024:-1024].max() / np.abs(wav.numpy()).max()
        assert e <= 2e-6, f"oracle specific_istft drifted from the reference: {e}"
        # the differentiable form: d(sum(w * wav))/d(mag) for a fixed random w, in float64 on the reference function
        G["stft_window"] = G["stft_window"].double()
        ref_fn = types.FunctionType(G["specific_istft"].__code__, G)      # same code, float64 window global
        mag64 = torch.from_numpy(mag).double().requires_grad_(True)
        wgt = torch.from_numpy(synth.uniform(8, 2 * 97536).reshape(2, 1, 97536)).double() - 0.5
        (ref_fn(mag64, torch.from_numpy(ang).double()) * wgt).sum().backward()
        np.savez_compressed(os.path.join(OUT, "specific_istft.npz"), wav=wav.numpy(),
                            dmag_sample=sample(mag64.grad, 4096), dmag_stats=stats(mag64.grad),
                            dmag_tile0_rows=mag64.grad[0, 0, :4, :].float().numpy().copy())
    finally:
        shutil.rmtree(work, ignore_errors=True)

    # ---------------------------------------------------------------- train step at B = 64 (BASELINE configs[2])
    if "--skip-b64" not in sys.argv:
        train_step_golden(64)
    # ---------------------------------------------------------------- the single-GPU legs of configs[3]: global batch 512 on
    # 1 / 2 / 4 GPUs is B = 512 / 256 / 128 per GPU (bench.py's strong_b512 record and the driver's strong-scaling runs)
    for b in batches_from_argv():
        train_step_golden(b, mask_seed=b)

    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))}
    print("golden fixtures written:", sizes)


if __name__ == "__main__":
    main()
