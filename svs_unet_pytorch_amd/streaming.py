"""End-to-end on-GPU separation of a waveform: STFT -> normalise -> 128-frame tiles -> U-Net mask -> masked
magnitude x mixture phase -> inverse STFT -> peak-normalise, with every intermediate resident in HBM
(BASELINE.json configs[4], in fp32; the bf16 MFMA variant is not built yet).

This is the composition of the reference's three CLI stages without the .npy round trips:
  data.py to_spec   (data.py:78-109)     svs_stft_fwd + svs_absmax + svs_scale_by_inv
  inference.py      (inference.py:65-127) inference.separate's bookkeeping, kept on the device
  data.py to_wave   (data.py:151-166)     svs_istft + peak normalisation
Channels of a stereo signal are separated independently (the reference itself downmixes to mono with
librosa.load(mono=True), data.py:78; a stereo caller gets per-channel masks).
"""
from __future__ import annotations

import torch

from . import _lib
from .config import HOP_SIZE, INPUT_LEN, WINDOW_SIZE
from .data import istft, stft_magphase


@torch.no_grad()
def separate_spectrogram_device(model, mag: torch.Tensor, seg_len: int = INPUT_LEN, vocal_solo: bool = True, max_batch: int = 256):
    """(F+1, T) magnitude on the GPU -> (F+1, T) separated magnitude on the GPU (inference.py:65-127 semantics)."""
    crop = mag[1:, :]
    F_, T = crop.shape
    n = T // seg_len + 1
    if T % seg_len == 0:
        n -= 1                                     # the empty last segment is skipped (inference.py:88)
    if n == 0:
        return torch.zeros_like(mag)
    padded = torch.zeros((F_, n * seg_len), dtype=torch.float32, device=mag.device)
    padded[:, :T] = crop
    tiles = padded.view(F_, n, seg_len).permute(1, 0, 2).contiguous().unsqueeze(1)
    out = torch.empty_like(tiles)
    was_training = model.training
    model.eval()
    for s in range(0, n, max_batch):
        t = tiles[s:s + max_batch]
        mask = model(t)
        _lib.check(_lib.lib().svs_apply_mask(t.data_ptr(), mask.data_ptr(), out[s:s + max_batch].data_ptr(), t.numel(),
                                             0 if vocal_solo else 1, _lib.stream_ptr()), "svs_apply_mask")
    model.train(was_training)
    full = out[:, 0].permute(1, 0, 2).reshape(F_, n * seg_len)[:, :T]
    return torch.cat([torch.zeros((1, T), dtype=torch.float32, device=mag.device), full], dim=0)


@torch.no_grad()
def separate_waveform(model, y: torch.Tensor, vocal_solo: bool = True, n_fft: int = WINDOW_SIZE, hop: int = HOP_SIZE,
                      peak: float | None = 0.9):
    """float32 samples (n,) or (channels, n) on the GPU -> separated samples (hop*(T-1),) or (channels, hop*(T-1))."""
    if y.dim() == 2:
        return torch.stack([separate_waveform(model, y[c], vocal_solo, n_fft, hop, peak) for c in range(y.shape[0])])
    mag, phase = stft_magphase(y, n_fft, hop)
    L = _lib.lib()
    ws = torch.empty(4096, dtype=torch.uint8, device=y.device)
    norm = torch.empty(1, dtype=torch.float32, device=y.device)
    _lib.check(L.svs_absmax(mag.data_ptr(), mag.numel(), norm.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "svs_absmax")
    _lib.check(L.svs_scale_by_inv(mag.data_ptr(), mag.numel(), norm.data_ptr(), 1.0, _lib.stream_ptr()), "svs_scale_by_inv")
    pred = separate_spectrogram_device(model, mag, INPUT_LEN, vocal_solo)
    return istft(pred, phase, n_fft, hop, peak=peak)
