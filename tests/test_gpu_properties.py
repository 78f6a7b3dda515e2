"""Size-independent properties at BASELINE.json's full sizes, edge cases and error behaviour (-m gpu).

The oracle finishes the small cases in seconds; at the full sizes (240 s of stereo audio = 216 tiles, batch 64 waveforms for the
multi-resolution loss) the checks are properties the domain offers: exact linearity in the mask, the tile path against the
general path, invariance under permutation of the batch, batch-size independence, loss(y, y) = 0, scale invariance."""
import numpy as np
import pytest
import torch

from oracle import stft_oracle as so
from svs_unet_pytorch_amd import _lib, synth

pytestmark = pytest.mark.gpu

DEV = "cuda"


def L():
    return _lib.lib()


def S():
    return _lib.stream_ptr()


def make_model():
    from svs_unet_pytorch_amd.model import UNet
    m = UNet()
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synth.closed_form_state().items()})
    return m.to(DEV).eval()


# ------------------------------------------------------------------------------------------------
# STFT / iSTFT at 240 s of 44.1 kHz stereo (BASELINE configs[4])
# ------------------------------------------------------------------------------------------------
def test_signal_path_full_size_properties(report):
    from svs_unet_pytorch_amd.data import istft_from_tiles, stft_to_tiles
    n = 240 * 44100
    y = torch.from_numpy(np.stack([synth.audio(n, 30), synth.audio(n, 31) * 0.25])).to(DEV)
    tiles, phase, peak, T = stft_to_tiles(y)
    assert T == 1 + n // 768 == 13782 and tiles.shape == (2, 108, 1, 512, 128) and phase.shape == (2, T, 513)
    assert torch.all(tiles.view(2, 108, 512, 128)[:, -1, :, T - 107 * 128:] == 0)      # padding of the last tile (inference.py:90-92)
    norm = phase.abs()
    assert report("phasors are unit vectors (or 0 for an empty bin)", torch.where(norm > 0, (norm - 1).abs(), torch.zeros_like(norm)).max().item(), 2e-6)
    # (1) homogeneity: channel 1 is 0.25 x another signal, so scaling the input by 4 scales magnitudes by exactly 4, phasors equal
    t4, p4, pk4, _ = stft_to_tiles(y * 4.0)
    assert torch.equal(t4, tiles * 4.0) and torch.equal(pk4, peak * 4.0)
    # (2) peak is the maximum over bins 0..512, so never below the maximum of the tile rows (bins 1..512)
    assert torch.all(peak >= tiles.amax(dim=(1, 2, 3, 4)))
    # (3) the inverse is exactly linear in the mask for power-of-two masks, and invert is its complement
    plain = istft_from_tiles(tiles, None, phase, T)
    half = istft_from_tiles(tiles, torch.full_like(tiles, 0.5), phase, T)
    half_inv = istft_from_tiles(tiles, torch.full_like(tiles, 0.5), phase, T, invert=True)
    assert plain.shape == (2, 768 * (T - 1))
    assert torch.equal(half, plain * 0.5) and torch.equal(half_inv, half)
    zero = istft_from_tiles(tiles, torch.ones_like(tiles), phase, T, invert=True)
    assert torch.all(zero == 0)
    # (4) tile path == general path (svs_stft_fwd / svs_istft on a (513, T) spectrogram with the DC row zeroed), channel 0
    mag = torch.empty((513, T), device=DEV)
    phs = torch.empty((513, T, 2), device=DEV)
    _lib.check(L().svs_stft_fwd(y[0].data_ptr(), n, 1024, 768, mag.data_ptr(), phs.data_ptr(), S()))
    full = tiles[0, :, 0].permute(1, 0, 2).reshape(512, 108 * 128)[:, :T]
    assert torch.equal(full, mag[1:])
    assert report("frame-major phasors vs f-major phasors", (phase[0].T - torch.view_as_complex(phs)).abs().max().item(), 1e-6)
    out = torch.empty(768 * (T - 1), device=DEV)
    ws = torch.empty(int(L().svs_istft_workspace_bytes(1024, 768, T)), dtype=torch.uint8, device=DEV)
    # (5) round trip with all 513 bins: analysis + synthesis is the identity away from the two ends
    _lib.check(L().svs_istft(mag.data_ptr(), phs.data_ptr(), 0, 1024, 768, T, out.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    inner = slice(1024, 768 * (T - 1) - 1024)
    assert report("round trip of 240 s (all bins), relative to the signal's peak", (out[inner] - y[0, inner]).abs().max().item() / y[0].abs().max().item(), 2e-6)
    mag[0] = 0
    _lib.check(L().svs_istft(mag.data_ptr(), phs.data_ptr(), 0, 1024, 768, T, out.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    assert report("istft_tiles == svs_istft on the same spectrogram", (out - plain[0]).abs().max().item() / plain[0].abs().max().item(), 3e-6)
    # (6) a window of the long signal against the numpy oracle: frames 6000..6127 depend on samples around 6000*768 only
    f0 = 6000
    seg2 = y[0, f0 * 768 - 768: (f0 + 128) * 768].cpu().numpy()                  # frame j+1 of seg2 is frame f0+j of y
    mag_o = np.abs(so.stft(seg2))[1:, 1:129]
    got = full[:, f0:f0 + 128].cpu().numpy()
    assert report("frames 6000..6127 of the 240 s signal vs numpy oracle", np.abs(got - mag_o).max() / mag_o.max(), 2e-6)


def test_signal_path_edge_cases(report):
    """Mono, very short and ragged inputs (data.py:78-109 accepts any length; frames = 1 + n // hop), and argument errors."""
    from svs_unet_pytorch_amd.data import istft_from_tiles, stft_to_tiles
    for n in (1, 700, 768, 1536, 5000):
        y_np = synth.audio(max(n, 2), 40)[:n].astype(np.float32)
        tiles, phase, peak, T = stft_to_tiles(torch.from_numpy(y_np[None]).to(DEV))
        assert T == 1 + n // 768 and tiles.shape == (1, 1, 1, 512, 128)
        mag_o = np.abs(so.stft(y_np))
        assert report(f"stft_tiles n={n} (T={T})", np.abs(tiles[0, 0, 0, :, :T].cpu().numpy() - mag_o[1:]).max() / max(mag_o.max(), 1e-30), 3e-6)
        assert torch.all(tiles[0, 0, 0, :, T:] == 0)
        assert report(f"stft_tiles n={n} peak", abs(peak.item() - mag_o.max()) / max(mag_o.max(), 1e-30), 3e-6)
        if T >= 2:
            got = istft_from_tiles(tiles, None, phase, T).cpu().numpy()
            assert got.shape == (1, 768 * (T - 1))
            spec = mag_o * so.magphase(so.stft(y_np))[1]
            spec[0] = 0
            want = so.istft(spec)
            assert report(f"istft_tiles n={n}", np.abs(got[0] - want).max() / max(np.abs(want).max(), 1e-30), 5e-5)
        else:
            with pytest.raises(RuntimeError, match="frames"):                   # librosa.istft of one frame has no samples to return
                istft_from_tiles(tiles, None, phase, T)
    y = torch.zeros((1, 4096), device=DEV)
    mag = torch.empty((513, 6), device=DEV)
    with pytest.raises(RuntimeError, match="n_fft"):
        _lib.check(L().svs_stft_fwd(y.data_ptr(), 4096, 2048, 768, mag.data_ptr(), None, S()))
    with pytest.raises(RuntimeError, match="bad arguments"):
        _lib.check(L().svs_stft_fwd(None, 4096, 1024, 768, mag.data_ptr(), None, S()))
    with pytest.raises(RuntimeError, match="bad arguments"):
        _lib.check(L().svs_stft_fwd(y.data_ptr(), 0, 1024, 768, mag.data_ptr(), None, S()))
    tiles, phase, _, T = stft_to_tiles(y)
    out = torch.empty(4096, device=DEV)
    with pytest.raises(RuntimeError, match="hop"):                              # hop > n_fft would leave samples that no frame covers
        _lib.check(L().svs_istft_tiles(tiles.data_ptr(), 512 * 128, 128, 512, 1, None, 0, torch.view_as_real(phase).data_ptr(), 1, 1, 1024, 1025, T,
                                       out.data_ptr(), None, S()))
    # all-zero input: magnitudes 0, phasors (1, 0) like numpy's angle(0) = 0 (data.py:81 librosa.magphase), inverse 0
    tiles, phase, peak, T = stft_to_tiles(y)
    assert torch.all(tiles == 0) and peak.item() == 0
    assert torch.all(istft_from_tiles(tiles, None, phase, T) == 0)


# ------------------------------------------------------------------------------------------------
# network forward at 216 tiles
# ------------------------------------------------------------------------------------------------
def test_eval_forward_full_batch_properties(report):
    """Tiles are independent in eval mode (BatchNorm uses running statistics): the mask of a tile does not depend on its
    neighbours in the batch nor on its position -- bitwise under a permutation of the same batch (every output element's
    summation order is fixed by the plan, which depends on the batch SIZE only), to fp32 rounding across batch sizes."""
    model = make_model()
    B = 216
    x = torch.from_numpy(synth.uniform(synth.SEED_MIX, B * 512 * 128, 91 << 32).reshape(B, 1, 512, 128)).to(DEV)
    perm = torch.from_numpy(np.random.RandomState(5).permutation(B)).to(DEV)
    with torch.no_grad():
        full = model(x)
        permuted = model(x[perm].contiguous())
        assert torch.equal(permuted, full[perm])
        assert torch.equal(model(x), full)                                       # and reproducible run to run
        for lo, hi in ((0, 1), (7, 23), (200, 216)):
            part = model(x[lo:hi].contiguous())
            assert report(f"eval forward of tiles {lo}..{hi} alone vs inside the batch of 216", (part - full[lo:hi]).abs().max().item(), 2e-6)
        assert float(full.min()) >= 0.0 and float(full.max()) <= 1.0             # sigmoid range (model.py:108)
        model.eval_precision = "bf16"
        b_full = model(x)
        assert torch.equal(model(x[perm].contiguous()), b_full[perm])
        assert report("bf16 network, 216 tiles: mean |mask - fp32 mask|", (b_full - full).abs().mean().item(), 1e-2)
        model.eval_precision = "fp32"


def test_streaming_bf16_against_fp32(report):
    """End to end with bf16 convolutions (BASELINE configs[4]) against the fp32 chain on the same audio: a bounded deviation
    of the separated waveform (both peak-normalised to 0.9), not a parity gate."""
    from svs_unet_pytorch_amd.streaming import separate_waveform
    model = make_model()
    n = 44100 * 20
    y = torch.from_numpy(np.stack([synth.audio(n, 50), synth.audio(n, 51)])).to(DEV)
    a = separate_waveform(model, y, precision="fp32")
    b = separate_waveform(model, y, precision="bf16")
    assert a.shape == b.shape == (2, 768 * (n // 768))
    assert torch.isfinite(b).all() and abs(b.abs().max().item() - 0.9) < 1e-5
    assert report("separated waveform, bf16 vs fp32 network: max |d| (peak 0.9)", (a - b).abs().max().item(), 2e-2)
    assert report("separated waveform, bf16 vs fp32 network: rms d / rms", ((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item(), 2e-2)
    assert model.eval_precision == "fp32"                                         # the per-call override does not stick


# ------------------------------------------------------------------------------------------------
# multi-resolution STFT loss at the training batch (train.py:293: 64 waveforms of 97,536 samples)
# ------------------------------------------------------------------------------------------------
def test_mrstft_full_batch_properties(report):
    B, n = 64, 97536
    y = torch.from_numpy((synth.uniform(60, B * n).reshape(B, n) - 0.5) * 0.5).to(DEV)
    x = (y * 0.8 + torch.from_numpy((synth.uniform(61, B * n).reshape(B, n) - 0.5) * 0.1).to(DEV)).contiguous()
    ws = torch.empty(int(L().svs_mrstft_workspace_bytes(B, n)) + 4096, dtype=torch.uint8, device=DEV)

    def run(a, b, want_grad=True, scale=1.0):
        loss = torch.zeros(1, device=DEV)
        d = torch.empty_like(a) if want_grad else None
        _lib.check(L().svs_mrstft_loss_fwd_bwd(a.data_ptr(), b.data_ptr(), B, n, scale, loss.data_ptr(), None if d is None else d.data_ptr(),
                                               ws.data_ptr(), ws.numel(), S()))
        return loss.item(), d

    l_xy, g = run(x, y)
    assert np.isfinite(l_xy) and l_xy > 0 and torch.isfinite(g).all()
    l_yy, _ = run(y, y, want_grad=False)        # (the gradient at x = y is a subgradient of the L1 log term: not tested)
    assert report("MR-STFT loss(y, y) (both frames share one complex FFT, so X and Y differ by its rounding)", l_yy, 2e-5)
    l_val, _ = run(x, y, want_grad=False)
    assert l_val == l_xy
    # scale invariance: both terms depend on |X| / |Y| only (spectral convergence is a ratio, the other is a log ratio);
    # a factor of 2 is exact in every step except the eps clamp of the magnitude (1e-8 under the square root)
    l2, g2 = run(x * 2, y * 2)
    assert report("loss(2x, 2y) vs loss(x, y)", abs(l2 - l_xy) / l_xy, 1e-5)
    # ... for the gradient the clamp matters: a bin with |X| just under 1e-4 has gradient 0, and ~1e4 x the typical size once
    # doubling lifts it over the clamp (the fp64 oracle shows the same: 3,337 of 6.2 M samples differ by more than 1e-3 of the maximum), so
    # the property is checked per waveform, on the median
    per = (g2 * 2 - g).norm(dim=1) / g.norm(dim=1)
    assert report("gradient(2x, 2y) vs gradient(x, y) / 2, median over the 64 waveforms", per.median().item(), 1e-4)
    assert int((per > 1e-4).sum()) <= 16
    # grad_scale is linear
    _, g3 = run(x, y, scale=4.0)
    assert torch.equal(g3, g * 4.0)
    # the full batch against the fp64 restatement (about 10 s of host time; tolerance as in test_mrstft_loss_and_gradient:
    # the gradient is as accurate as the small bins of an fp32 FFT -- measured 9.4e-4)
    from oracle import mrstft_oracle as mo
    want_loss, want_grad = mo.mrstft_loss_and_grad(x.cpu().double(), y.cpu().double())
    assert report("MR-STFT loss, batch 64, vs fp64 oracle", abs(l_xy - want_loss) / want_loss, 1e-5)
    assert report("MR-STFT gradient, batch 64, vs fp64 oracle (rel-L2)", ((g.cpu().double() - want_grad).norm() / want_grad.norm()).item(), 3e-3)
    # too-small workspace: an error code and a message, nothing launched
    with pytest.raises(RuntimeError, match="workspace"):
        loss = torch.zeros(1, device=DEV)
        _lib.check(L().svs_mrstft_loss_fwd_bwd(x.data_ptr(), y.data_ptr(), B, n, 1.0, loss.data_ptr(), g.data_ptr(), ws.data_ptr(), 1024, S()))


def test_argument_errors(report):
    """Error behaviour of the boundary: negative return code + svs_last_error_string(), never a launch on bad geometry."""
    with pytest.raises(RuntimeError, match="unknown"):
        _lib.tuning("NO_SUCH_SWITCH", 1)
    model = make_model()
    with pytest.raises((RuntimeError, ValueError)):
        model(torch.zeros((1, 1, 512, 128)))                                     # host tensor: the product path has no CPU fallback
    with pytest.raises((RuntimeError, ValueError)):
        model(torch.zeros((1, 2, 512, 128), device=DEV))                         # two channels
    model.eval_precision = "fp16"
    with pytest.raises(ValueError):
        model(torch.zeros((1, 1, 512, 128), device=DEV))
    model.eval_precision = "fp32"
    x = torch.zeros((2, 1, 512, 128), device=DEV)
    need = int(L().svs_unet_eval_workspace_bytes(2, 512, 128))
    ws = torch.empty(need, dtype=torch.uint8, device=DEV)
    out = torch.empty_like(x)
    model(x)                                                                      # builds the prepared blob
    rc = L().svs_unet_forward_eval(model._prepared.data_ptr(), x.data_ptr(), out.data_ptr(), 2, 512, 128, ws.data_ptr(), need // 2, S())
    assert rc != 0 and b"workspace" in L().svs_last_error_string()
    rc = L().svs_unet_forward_eval(model._prepared.data_ptr(), x.data_ptr(), out.data_ptr(), 0, 512, 128, ws.data_ptr(), need, S())
    assert rc < 0
