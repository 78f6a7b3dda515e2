"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (torch, autograd for the gradient) of the multi-resolution STFT loss that the reference adds to its
L1 terms:  train.py:24-26  `mrstft_loss_fn = auraloss.freq.MultiResolutionSTFTLoss(sample_rate=SAMPLE_RATE, device=device)`,
train.py:287-296  `total_loss = alpha_L1 * l1_loss + alpha_MR * mrstft_loss_fn(pred_voc_wav, target_voc_wav)`.

PARITY UNPINNED: the arithmetic lives in auraloss 0.4.0 (/root/reference/uv.lock:90-91), which is neither vendored
nor installable here, and the reference holds no fixture for it.  This file restates that package's published
definition for exactly the call the reference makes -- every constructor argument at its default except
`sample_rate` / `device`, which do not enter the arithmetic (they matter only with perceptual weighting / a mel
scale, both off by default):

  MultiResolutionSTFTLoss: fft_sizes (1024, 2048, 512), hop_sizes (120, 240, 50), win_lengths (600, 1200, 240),
      window "hann_window", w_sc = 1, w_log_mag = 1, w_lin_mag = 0, w_phs = 0; result = mean over the resolutions of
  STFTLoss: x_stft = torch.stft(x, n_fft, hop, win_length, hann_window(win_length), return_complex=True)
            (torch defaults: center=True, pad_mode="reflect", onesided);
            mag = sqrt(clamp(re^2 + im^2, min=1e-8));
            loss = SpectralConvergence + LogSTFTMagnitude
               SpectralConvergence = mean over the batch of ||y_mag_b - x_mag_b||_F / ||y_mag_b||_F
                                     (0.4.0: `torch.norm(..., p="fro", dim=[-1, -2])` per item, then `.mean()`; the 0.2.x
                                      form took ONE ratio of norms over the whole batch tensor -- recalled from the package's
                                      published source, like everything here: parity unpinned)
               LogSTFTMagnitude    = mean |log(x_mag) - log(y_mag)|            (L1Loss, reduction "mean")
  inputs (B, 1, L) are flattened to (B, L).

The pieces built on top of it (`specific_istft`, train.py:33-60, and the total of train.py:296) ARE pinned
(oracle/stft_oracle.py, tests/golden/specific_istft.npz).
"""
from __future__ import annotations

import torch

FFT_SIZES = (1024, 2048, 512)
HOP_SIZES = (120, 240, 50)
WIN_LENGTHS = (600, 1200, 240)
EPS = 1e-8
ALPHA_L1 = 166.66      # train.py:24
ALPHA_MR = 0.66        # train.py:25


def stft_mag(x: torch.Tensor, n_fft: int, hop: int, win: int) -> torch.Tensor:
    window = torch.hann_window(win, dtype=x.dtype, device=x.device)
    s = torch.stft(x, n_fft, hop, win, window, return_complex=True)
    return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=EPS))


def stft_loss(x: torch.Tensor, y: torch.Tensor, n_fft: int, hop: int, win: int) -> torch.Tensor:
    xm, ym = stft_mag(x, n_fft, hop, win), stft_mag(y, n_fft, hop, win)
    sc = (torch.norm(ym - xm, p="fro", dim=[-1, -2]) / torch.norm(ym, p="fro", dim=[-1, -2])).mean()
    log_mag = torch.nn.functional.l1_loss(torch.log(xm), torch.log(ym))
    return sc + log_mag


def mrstft_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """x = predicted, y = target waveform, (B, 1, L) or (B, L)."""
    x, y = x.reshape(-1, x.shape[-1]), y.reshape(-1, y.shape[-1])
    total = 0.0
    for n_fft, hop, win in zip(FFT_SIZES, HOP_SIZES, WIN_LENGTHS):
        total = total + stft_loss(x, y, n_fft, hop, win)
    return total / len(FFT_SIZES)


def mrstft_loss_and_grad(x: torch.Tensor, y: torch.Tensor):
    """(loss value, d loss / d x) by autograd."""
    x = x.detach().clone().requires_grad_(True)
    loss = mrstft_loss(x, y.detach())
    loss.backward()
    return float(loss), x.grad.detach()


def specific_istft_torch(magnitude: torch.Tensor, phase: torch.Tensor, hop: int = 768) -> torch.Tensor:
    """train.py:33-60 restated on torch ops (differentiable): (B,1,512,T) magnitude / angle -> (B,1,hop*(T-1))."""
    m = torch.nn.functional.pad(magnitude, (0, 0, 1, 0))
    a = torch.nn.functional.pad(phase, (0, 0, 1, 0))
    s = torch.polar(m, a).squeeze(1)
    w = torch.hann_window(1024, dtype=magnitude.dtype, device=magnitude.device)
    return torch.istft(s, n_fft=1024, hop_length=hop, win_length=1024, window=w, return_complex=False).unsqueeze(1)


def total_loss(mask, mix, voc, mix_phase, voc_phase, alpha_l1: float = ALPHA_L1, alpha_mr: float = ALPHA_MR):
    """train.py:274-296: (total, l1, mr) for a given mask (all (B,1,512,T))."""
    pred_vocal = mask * mix
    pred_accomp = (1 - mask) * mix
    target_accomp = torch.clamp(mix - voc, min=0.0)
    l1 = torch.nn.functional.l1_loss(pred_vocal, voc) + torch.nn.functional.l1_loss(pred_accomp, target_accomp)
    mr = mrstft_loss(specific_istft_torch(pred_vocal, mix_phase), specific_istft_torch(voc, voc_phase))
    return alpha_l1 * l1 + alpha_mr * mr, l1, mr


def train_grads_full(state, mix, voc, mix_phase, voc_phase, dropout_masks=None, alpha_l1: float = ALPHA_L1, alpha_mr: float = ALPHA_MR):
    """Gradients of the reference's full objective (train.py:274-299) for the oracle U-Net in training mode.
    Returns (l1, mr, grads dict); `state`'s BatchNorm statistics are updated as a training forward does."""
    from collections import OrderedDict

    from oracle import unet_oracle as uo
    keys = uo.param_keys(state)
    leaves = {k: state[k].detach().clone().requires_grad_(True) for k in keys}
    work = OrderedDict((k, leaves.get(k, v)) for k, v in state.items())
    mask = uo.forward(work, mix, training=True, dropout_masks=dropout_masks, update_stats=True)
    total, l1, mr = total_loss(mask, mix, voc, mix_phase, voc_phase, alpha_l1, alpha_mr)
    total.backward()
    for k, v in work.items():
        if k not in leaves:
            state[k] = v
    return float(l1.detach()), float(mr.detach()), {k: leaves[k].grad.detach() for k in keys}
