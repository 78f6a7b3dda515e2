"""CPU suite (-m "not gpu"): pins the oracle (oracle/*.py, the CPU restatement used as the checker) against
the golden fixtures that oracle/gen_golden.py captured from the imported reference, and against torch's own
stft/istft (the reference's in-repo inverse, train.py:51-58)."""
import numpy as np
import pytest
import torch

from oracle import stft_oracle as so
from oracle import tiling_oracle as to
from oracle import unet_oracle as uo
from svs_unet_pytorch_amd import synth


@pytest.fixture(scope="module")
def state():
    return uo.to_torch_state(synth.closed_form_state())


def _sample(t, n=256):
    f = t.detach().reshape(-1)
    step = max(f.numel() // n, 1)
    return f[::step][:n].to(torch.float32).numpy()


def test_eval_forward_matches_reference(golden, state):
    g = golden("eval_forward.npz")
    mix16, _ = synth.tiles(16)
    taps = {}
    with torch.no_grad():
        m1 = uo.forward(state, torch.from_numpy(mix16[:1]), taps=taps)
    assert np.abs(m1[0, 0].numpy() - g["mask_tile0"]).max() <= 1e-6
    for k, v in taps.items():
        if k == "mask":
            continue
        key = f"tap.{k}"
        assert tuple(g[key + ".shape"]) == tuple(v.shape), k
        scale = max(abs(g[key + ".stats"][2]), abs(g[key + ".stats"][3]))
        assert np.abs(_sample(v) - g[key + ".sample"]).max() <= 1e-5 * scale, k
    with torch.no_grad():
        m4 = uo.forward(state, torch.from_numpy(mix16[:4]))
    assert np.abs(m4.double().sum((1, 2, 3)).numpy() - g["mask16_sum"][:4]).max() / 65536 <= 1e-6
    assert np.abs(m4[:, 0, :8, :8].numpy() - g["mask16_corner"][:4]).max() <= 1e-6


def test_eval_odd_sizes_match_reference(golden, state):
    g = golden("eval_odd_sizes.npz")
    for (h, w) in ((513, 128), (512, 100), (64, 16)):
        x = torch.from_numpy(synth.uniform(synth.SEED_MIX, h * w, 7 << 32).reshape(1, 1, h, w))
        with torch.no_grad():
            y = uo.forward(state, x)
        assert y.shape == x.shape
        assert np.abs(y[0, 0].numpy() - g[f"mask_{h}x{w}"]).max() <= 1e-6


def test_output_size_rule():
    # ConvTranspose2d(k5,s2,p2,output_size=...) (model.py:183): natural size 2n-1, output_padding 0 or 1
    assert uo._deconv_output_padding((8, 2), (16, 4)) == (1, 1)
    assert uo._deconv_output_padding((9, 2), (17, 4)) == (0, 1)
    with pytest.raises(ValueError):
        uo._deconv_output_padding((8, 2), (18, 4))


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_train_step_matches_reference_fp64(golden, tag):
    g = golden("train_steps.npz")
    names = list(g[tag + ".param_names"])
    B = 4
    mix4, voc4 = synth.tiles(B, first_tile=100)
    mix, voc = torch.from_numpy(mix4).double(), torch.from_numpy(voc4).double()
    st = uo.to_torch_state(synth.closed_form_state(trained_stats=False), torch.float64)
    assert uo.param_keys(st) == names
    opt = uo.new_adam_state(st)
    for step in range(2):
        masks = [torch.from_numpy(m).double() for m in synth.dropout_masks(B, seed=99, step=step)] if tag == "drop" else None
        loss, grads = uo.train_step(st, opt, mix, voc, dropout_masks=masks)
        p = f"{tag}.f64.step{step}."
        assert abs(loss - float(g[p + "loss"])) <= 1e-12
        gn = np.array([grads[n].norm().item() for n in names])
        assert np.abs(gn - g[p + "grad_norm"]).max() <= 1e-9 * max(g[p + "grad_norm"].max(), 1.0)
        for n in ("conv1.0.weight", "deconv6.weight", "conv2.1.weight", "deconv6.bias"):
            assert np.abs(_sample(grads[n], 128) - g[p + "grad_sample." + n]).max() <= 1e-6 * max(np.abs(g[p + "grad_sample." + n]).max(), 1e-12) + 1e-12
        for k in st:
            if "running_" in k:
                assert np.abs(st[k].float().numpy() - g[p + "buf." + k]).max() <= 1e-6, k
            if "num_batches_tracked" in k:
                assert int(st[k]) == step + 1


def test_inference_tiling_matches_reference_script(golden, state):
    g = golden("inference_tiling.npz")
    plans = {1: [(0, 1, 127)], 127: [(0, 127, 1)], 128: [(0, 128, 0)], 129: [(0, 128, 0), (128, 129, 127)],
             256: [(0, 128, 0), (128, 256, 0)], 300: [(0, 128, 0), (128, 256, 0), (256, 300, 84)]}
    for T, want in plans.items():                       # SURVEY.md 8a row A8 (restated from inference.py:75-92)
        assert to.segment_plan(T) == want
    assert to.segment_plan(0) == []

    def fn(tile):
        with torch.no_grad():
            return uo.forward(state, torch.from_numpy(tile)).numpy()

    for n, T in enumerate(g["lengths"]):
        T = int(T)
        if T > 129:
            continue                                     # keep the CPU suite short; the GPU suite covers all
        spec = synth.uniform(synth.SEED_MIX, 513 * T, (200 + n) << 32).reshape(513, T)
        got = to.separate(spec, fn, vocal_solo=True)
        assert got.shape == (513, T) and got.dtype == np.float32
        assert np.abs(got - g[f"solo1.T{T}"]).max() <= 1e-6
        if f"solo0.T{T}" in g.files:
            assert np.abs(to.separate(spec, fn, vocal_solo=False) - g[f"solo0.T{T}"]).max() <= 1e-6


def test_stft_oracle_against_torch():
    for n in (20000, 97536, 100000):
        y = synth.audio(n)
        d = so.stft(y)
        assert d.shape == (513, 1 + n // 768) and d.dtype == np.complex64            # frames = 1 + len//hop
        ref = torch.stft(torch.from_numpy(y).double(), 1024, 768, 1024, torch.hann_window(1024, dtype=torch.float64), center=True,
                         pad_mode="constant", return_complex=True).numpy()
        assert np.abs(d - ref).max() <= 1e-5 * np.abs(ref).max()
        mag, ph = so.magphase(d)
        assert mag.dtype == np.float32 and ph.dtype == np.complex64
        assert np.abs(np.abs(ph) - 1).max() <= 1e-5
        back = so.istft(mag * ph)
        assert back.shape == (768 * (d.shape[1] - 1),)
        assert np.abs(back - y[: back.size])[1024:-1024].max() <= 2e-5             # hop 768 envelope min 0.043: interior only
    z = np.zeros((513, 4), np.complex64)
    mag, ph = so.magphase(z)
    assert np.all(mag == 0) and np.all(ph == 1 + 0j)                                 # zero bins -> unit phasor 1+0j


def test_istft_oracle_against_torch_istft():
    """train.py:51-58: torch.istft(n_fft=1024, hop=768, win=1024, hann) is the reference's own inverse."""
    T = 128
    mag = synth.uniform(3, 2 * 512 * T).reshape(2, 1, 512, T)
    ang = (synth.uniform(4, 2 * 512 * T) * 2 * np.pi - np.pi).astype(np.float32).reshape(2, 1, 512, T)
    got = so.specific_istft(mag, ang)
    assert got.shape == (2, 1, 97536)
    m = torch.nn.functional.pad(torch.from_numpy(mag).double(), (0, 0, 1, 0))
    a = torch.nn.functional.pad(torch.from_numpy(ang).double(), (0, 0, 1, 0))
    want = torch.istft(torch.polar(m, a).squeeze(1), n_fft=1024, hop_length=768, win_length=1024,
                       window=torch.hann_window(1024, dtype=torch.float64), return_complex=False).unsqueeze(1).numpy()
    assert np.abs(got - want)[..., 1024:-1024].max() <= 1e-5 * np.abs(want).max()


def test_to_spec_normalises_by_mixture_max():
    y_mix, y_voc = synth.audio(30000, 0), synth.audio(25000, 1) * 0.5
    spec_mix, _ = so.to_spec(y_mix, y_mix)
    spec_voc, ph = so.to_spec(y_mix, y_voc)
    assert abs(spec_mix.max() - 1.0) <= 1e-6                                         # data.py:84-85,105
    assert spec_voc.shape == spec_mix.shape == ph.shape                              # data.py:97-98 length alignment
    w = so.to_wave(spec_voc, ph)
    assert abs(np.abs(w).max() - 0.9) <= 1e-6                                        # data.py:162-164


# ------------------------------------------------------------------------------------------------
# fixtures captured from the reference's own train.py run as a script (oracle/gen_golden_train.py)
# ------------------------------------------------------------------------------------------------
def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_dataset_items_match_reference_dataset(golden):
    """SpectrogramDataset.__getitem__ (train.py:86-143): the oracle's crop_item / crop_phase reproduce the reference's
    four tensors bit for bit (sha256 of the float32 bytes) for random-start, exact-length, short and L+1 songs."""
    import random
    g = golden("dataset_items.npz")
    lengths = [int(t) for t in g["lengths"]]
    assert int(g["len"]) == len(lengths) * 64                                        # train.py:83-84
    songs = {n: synth.song(n, T) for n, T in enumerate(lengths)}
    for seed in g["seeds"]:
        for idx in (0, 1, 2, 3, 5, 6):
            n = idx % len(lengths)
            T = lengths[n]
            random.seed(int(seed) * 1000 + idx)
            start = random.randint(0, T - 128) if T > 128 else 0                     # train.py:121: ONE draw per item
            p = f"s{int(seed)}.i{idx}."
            assert start == int(g[p + "start"])
            mix, voc, pm, pv = songs[n]
            m, v = to.crop_item(mix, voc, start)
            assert m.shape == v.shape == (1, 512, 128) and m.dtype == np.float32
            assert np.array_equal(_sha(m), g[p + "mix_sha"]) and np.array_equal(_sha(v), g[p + "voc_sha"])
            assert np.array_equal(_sha(to.crop_phase(pm, start)), g[p + "mix_phase_sha"])
            assert np.array_equal(_sha(to.crop_phase(pv, start)), g[p + "voc_phase_sha"])


def test_specific_istft_oracle_matches_reference_function(golden):
    """train.py:33-60 called on the reference's own function object: waveform (fp32) and d(sum w*wav)/d(mag) (fp64)."""
    g = golden("specific_istft.npz")
    T = 128
    mag = synth.uniform(3, 2 * 512 * T).reshape(2, 1, 512, T)
    ang = (synth.uniform(4, 2 * 512 * T) * 2 * np.pi - np.pi).astype(np.float32).reshape(2, 1, 512, T)
    got = so.specific_istft(mag, ang)
    want = g["wav"]
    assert got.shape == want.shape == (2, 1, 97536)
    assert np.abs(got - want)[..., 1024:-1024].max() <= 2e-6 * np.abs(want).max()
    # adjoint of the oracle's linear map mag -> wav against the reference's autograd gradient
    wgt = synth.uniform(8, 2 * 97536).reshape(2, 1, 97536).astype(np.float64) - 0.5
    dmag = so.specific_istft_adjoint(wgt, ang)
    f = dmag.reshape(-1)
    step = max(f.size // 4096, 1)
    scale = np.abs(g["dmag_sample"]).max()
    assert np.abs(f[::step][:4096] - g["dmag_sample"]).max() <= 1e-6 * scale
    assert np.abs(dmag[0, 0, :4, :] - g["dmag_tile0_rows"]).max() <= 1e-6 * scale


def test_train_step_b64_oracle_fp32_within_reference_noise(golden):
    """BASELINE configs[2] batch size: the fp32 oracle against the reference's own fp64 / fp32 runs at B = 64."""
    g = golden("train_b64.npz")
    names = list(g["param_names"])
    B = 64
    mix_np, voc_np = synth.tiles(B)
    st = uo.to_torch_state(synth.closed_form_state(trained_stats=False))
    masks = [torch.from_numpy(m) for m in synth.dropout_masks(B, seed=64, step=0)]
    loss, grads = uo.train_step(st, uo.new_adam_state(st), torch.from_numpy(mix_np), torch.from_numpy(voc_np), dropout_masks=masks)
    assert abs(loss - float(g["f64.loss"])) <= 1e-5 * float(g["f64.loss"])
    gn64, gn32 = g["f64.grad_norm"], g["f32.grad_norm"]
    for i, n in enumerate(names):
        noise = max(abs(gn32[i] - gn64[i]), 1e-4 * gn64[i], 2e-6)
        assert abs(grads[n].double().norm().item() - gn64[i]) <= 20 * noise, n


def test_mrstft_oracle_against_direct_numpy_frames():
    """oracle/mrstft_oracle.py goes through torch.stft; this restates the same published definition with explicit frames in
    numpy (reflect padding by n_fft/2, periodic Hann window of win_length centred in the n_fft frame, rfft, clamp, the two
    terms) so that a slip in the torch.stft arguments cannot hide.  auraloss itself: parity unpinned."""
    import numpy as np
    from oracle import mrstft_oracle as mo
    rng = np.random.default_rng(3)
    B, L = 2, 6000
    x = (rng.random((B, L)) - 0.5) * 0.6
    y = 0.7 * x + (rng.random((B, L)) - 0.5) * 0.2
    total = 0.0
    for n_fft, hop, win in zip(mo.FFT_SIZES, mo.HOP_SIZES, mo.WIN_LENGTHS):
        w = np.zeros(n_fft)
        w[(n_fft - win) // 2:(n_fft - win) // 2 + win] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(win) / win)
        mags = []
        for sig in (x, y):
            pad = np.pad(sig, ((0, 0), (n_fft // 2, n_fft // 2)), mode="reflect")
            frames = 1 + L // hop
            fr = np.stack([pad[:, t * hop:t * hop + n_fft] * w for t in range(frames)], axis=1)       # (B, frames, n_fft)
            spec = np.fft.rfft(fr, axis=-1)
            mags.append(np.sqrt(np.maximum(spec.real ** 2 + spec.imag ** 2, mo.EPS)))
        xm, ym = mags
        sc = (np.sqrt(((ym - xm) ** 2).sum(axis=(1, 2))) / np.sqrt((ym ** 2).sum(axis=(1, 2)))).mean()     # per waveform, then the mean
        total += sc + np.abs(np.log(xm) - np.log(ym)).mean()
    want = total / 3
    got = float(mo.mrstft_loss(torch.from_numpy(x), torch.from_numpy(y)))
    assert abs(got - want) <= 1e-10 * abs(want), (got, want)
