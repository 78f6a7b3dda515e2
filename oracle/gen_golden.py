"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Golden-vector generator; runs ONLY in the build container.

Imports the reference itself (/root/reference/model.py; its unused `import auraloss` at model.py:6
is satisfied with an empty stub module) and runs the reference's own inference.py as a script, on
inputs/weights from the counter-based generator (svs_unet_pytorch_amd/synth.py), and writes small
numeric fixtures to tests/golden/.  Nothing from the reference's sources is stored: fixtures are
inputs' seeds and expected outputs only.  While it is at it, it asserts that the CPU restatement in
oracle/ reproduces the reference (so a drifted restatement fails here, before any fixture is used).

Usage (build container):  python oracle/gen_golden.py
"""
from __future__ import annotations

import os
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
from oracle import unet_oracle as uo  # noqa: E402
from oracle import tiling_oracle as to  # noqa: E402


def import_reference_model():
    sys.modules.setdefault("auraloss", types.ModuleType("auraloss"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import model  # the reference's model.py
    return model


def sample(t: torch.Tensor, n: int = 256) -> np.ndarray:
    """Deterministic strided sample of a tensor (first element, then every numel//n-th)."""
    f = t.detach().reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].to(torch.float32).numpy().copy()


def stats(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.sum().item(), d.abs().sum().item(), d.min().item(), d.max().item()], np.float64)


def corner(t: torch.Tensor) -> np.ndarray:
    return t.detach()[..., :8, :8].to(torch.float32).numpy().copy()


def ref_model_with(state_np, model_mod):
    m = model_mod.UNet()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in state_np.items()}
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


class InjectedDropout(torch.nn.Module):
    """Stands in for Dropout2d(0.5) in the reference instance so both sides use the SAME masks."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask[:, :, None, None] if self.training else x


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model_mod = import_reference_model()
    state_np = synth.closed_form_state()
    assert list(state_np.keys()) == list(model_mod.UNet().state_dict().keys()), "state_dict order drifted"

    # ---------------------------------------------------------------- (1)+(2) eval forward, per-layer taps
    ref = ref_model_with(state_np, model_mod).eval()
    taps_ref = {}

    def hook(name):
        def f(_m, _i, o):
            taps_ref[name] = o.detach()
        return f

    for i in range(1, 7):
        getattr(ref, f"conv{i}")[0].register_forward_hook(hook(f"conv{i}.raw"))
        getattr(ref, f"conv{i}").register_forward_hook(hook(f"conv{i}.out"))
        getattr(ref, f"deconv{i}").register_forward_hook(hook(f"deconv{i}.raw"))
        if i < 6:
            getattr(ref, f"deconv{i}_BAD").register_forward_hook(hook(f"deconv{i}.out"))

    mix16, voc16 = synth.tiles(16)
    with torch.no_grad():
        mask1 = ref(torch.from_numpy(mix16[:1]))
        taps1 = dict(taps_ref)
        mask16 = ref(torch.from_numpy(mix16))
    # batch independence: tile 0 alone == tile 0 inside the batch of 16
    assert torch.equal(mask1[0], mask16[0]) or (mask1[0] - mask16[0]).abs().max() < 1e-6

    st = uo.to_torch_state(state_np)
    taps_or = {}
    with torch.no_grad():
        o1 = uo.forward(st, torch.from_numpy(mix16[:1]), training=False, taps=taps_or)
    err = (o1 - mask1).abs().max().item()
    assert err <= 1e-6, f"oracle eval forward drifted from the reference: {err}"
    for k, v in taps1.items():
        e = (taps_or[k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30)
        assert e <= 1e-5, f"oracle tap {k} drifted: {e}"

    g = {"mask_tile0": mask1[0, 0].numpy().copy()}
    g["mask16_sum"] = mask16.double().sum((1, 2, 3)).numpy()
    g["mask16_corner"] = corner(mask16[:, 0])
    g["mask16_stats"] = stats(mask16)
    for k, v in taps1.items():
        g[f"tap.{k}.stats"] = stats(v)
        g[f"tap.{k}.sample"] = sample(v)
        g[f"tap.{k}.shape"] = np.array(v.shape, np.int64)
    np.savez_compressed(os.path.join(OUT, "eval_forward.npz"), **g)

    # ---------------------------------------------------------------- (3) odd sizes (aaa.py:33,62 feeds 513 rows)
    g = {}
    for (h, w) in ((513, 128), (512, 100), (512, 32), (64, 16)):
        x = torch.from_numpy(synth.uniform(synth.SEED_MIX, h * w, 7 << 32).reshape(1, 1, h, w))
        with torch.no_grad():
            y = ref(x)
            yo = uo.forward(st, x, training=False)
        assert y.shape == x.shape
        assert (y - yo).abs().max().item() <= 1e-6
        g[f"mask_{h}x{w}"] = y[0, 0].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "eval_odd_sizes.npz"), **g)

    # ---------------------------------------------------------------- (4) train steps, B=4, injected dropout
    # fp32 gradients of this net carry ~1e-3 relative rounding noise (measured: the reference in
    # fp32 vs itself in fp64), and the conv biases that feed a BatchNorm have a true gradient of
    # exactly 0.  The fixtures therefore hold the reference run in float64 ("truth") AND in
    # float32 (the noise scale a correct fp32 implementation is allowed).
    chk = torch.nn.functional.dropout2d(torch.ones(8, 16, 3, 3), 0.5, True)
    assert set(chk.unique().tolist()) <= {0.0, 2.0} and (chk.amax((2, 3)) == chk.amin((2, 3))).all()
    B = 4
    mix4, voc4 = synth.tiles(B, first_tile=100)
    fresh_np = synth.closed_form_state(trained_stats=False)
    g = {}
    for tag, use_dropout in (("nodrop", False), ("drop", True)):
        for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
            mix4_t, voc4_t = torch.from_numpy(mix4).to(dt), torch.from_numpy(voc4).to(dt)
            refm = ref_model_with(fresh_np, model_mod).to(dt).train()
            refm.crit = torch.nn.L1Loss()                       # config.py:33,44; train.py:281-282 call shape
            opt = torch.optim.Adam(refm.parameters(), lr=1e-3)  # as model.py:116, rebuilt after the dtype cast
            st_o = uo.to_torch_state(fresh_np, dt)
            opt_o = uo.new_adam_state(st_o)
            for step in range(2):
                masks_np = synth.dropout_masks(B, seed=99, step=step) if use_dropout else None
                masks_t = [torch.from_numpy(m).to(dt) for m in masks_np] if use_dropout else None
                for i in range(5):
                    seq = getattr(refm, f"deconv{i + 1}_BAD")
                    seq[2] = InjectedDropout(masks_t[i]) if use_dropout else torch.nn.Identity()
                opt.zero_grad()
                # train.py:274-283, L1 terms
                mask = refm(mix4_t)
                pred_vocal = mask * mix4_t
                pred_accomp = (1 - mask) * mix4_t
                target_accomp = torch.clamp(mix4_t - voc4_t, min=0.0)
                loss = refm.crit(pred_vocal, voc4_t) + refm.crit(pred_accomp, target_accomp)
                loss.backward()
                names = [n for n, _ in refm.named_parameters()]
                grads = {n: p.grad.detach().clone() for n, p in refm.named_parameters()}
                opt.step()
                sd = refm.state_dict()

                if dt is torch.float64:   # the restatement must BE the reference's algorithm
                    lo, grads_o = uo.train_step(st_o, opt_o, mix4_t, voc4_t, dropout_masks=masks_t)
                    assert abs(lo - loss.item()) <= 1e-12, (lo, loss.item())
                    for n in names:
                        e = (grads_o[n] - grads[n]).norm().item()
                        assert e <= 1e-9 * max(grads[n].norm().item(), 1e-3), f"oracle grad {n} drifted: {e}"
                    for k in sd:
                        if sd[k].is_floating_point():
                            e = (st_o[k] - sd[k]).abs().max().item()
                            assert e <= 1e-9, f"oracle state {k} after step {step}: {e}"
                        else:
                            assert int(st_o[k]) == int(sd[k])

                p = f"{tag}.{dt_name}.step{step}."
                g[p + "loss"] = np.array(loss.item(), np.float64)
                g[p + "mask_stats"] = stats(mask)
                g[p + "grad_norm"] = np.array([grads[n].double().norm().item() for n in names], np.float64)
                g[p + "grad_sum"] = np.array([grads[n].double().sum().item() for n in names], np.float64)
                for n in ("conv1.0.weight", "conv4.0.weight", "deconv6.weight", "deconv3.weight",
                          "conv2.1.weight", "deconv2_BAD.0.bias", "conv6.1.bias", "deconv6.bias"):
                    g[p + "grad_sample." + n] = sample(grads[n], 128)
                if dt is torch.float64:
                    for k in sd:
                        if "running_" in k:
                            g[p + "buf." + k] = sd[k].to(torch.float32).numpy().copy()
        g[tag + ".param_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "train_steps.npz"), **g)

    # ---------------------------------------------------------------- (5) inference.py run as a script
    work = tempfile.mkdtemp(prefix="svs_golden_")
    try:
        mixdir, tar = os.path.join(work, "mixture"), os.path.join(work, "pred")
        os.makedirs(mixdir)
        ckpt = os.path.join(work, "svs_closed_form.pth")
        torch.save({"model_state_dict": {k: torch.from_numpy(np.array(v)) for k, v in state_np.items()}}, ckpt)
        lengths = (1, 127, 128, 129, 256, 300)
        specs = {}
        for n, T in enumerate(lengths):
            spec = synth.uniform(synth.SEED_MIX, 513 * T, (200 + n) << 32).reshape(513, T)
            specs[T] = spec
            np.save(os.path.join(mixdir, f"{n:04d}_len{T}_spec.npy"), spec)
        g = {}
        for solo in (1, 0):
            shutil.rmtree(tar, ignore_errors=True)
            argv = sys.argv
            cwd = os.getcwd()
            try:
                os.chdir(work)
                sys.argv = ["inference.py", "--model_path", ckpt, "--tar", tar,
                            "--mixture_folder", mixdir, "--vocal_solo", str(solo)]
                try:
                    runpy.run_path(os.path.join(REF, "inference.py"), run_name="__main__")
                except SystemExit as e:  # the script never calls exit on success
                    assert not e.code, e
            finally:
                sys.argv = argv
                os.chdir(cwd)
            for n, T in enumerate(lengths):
                got = np.load(os.path.join(tar, f"{n:04d}_len{T}_spec.npy"))
                assert got.shape == (513, T) and got.dtype == np.float32
                with torch.no_grad():
                    want = to.separate(specs[T], lambda t: uo.forward(st, torch.from_numpy(t)).numpy(),
                                       vocal_solo=bool(solo))
                assert np.abs(got - want).max() <= 1e-6, (T, np.abs(got - want).max())
                if solo == 1 or T in (129, 300):
                    g[f"solo{solo}.T{T}"] = got
        g["lengths"] = np.array(lengths, np.int64)
        g["plan_T"] = np.array([1, 127, 128, 129, 256, 300], np.int64)
        np.savez_compressed(os.path.join(OUT, "inference_tiling.npz"), **g)
    finally:
        shutil.rmtree(work, ignore_errors=True)

    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))}
    print("golden fixtures written:", sizes)


if __name__ == "__main__":
    main()
