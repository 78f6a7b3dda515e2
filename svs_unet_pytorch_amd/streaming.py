"""End-to-end on-GPU separation of a waveform: STFT -> normalise -> 128-frame tiles -> U-Net mask -> masked
magnitude x mixture phase -> inverse STFT -> peak-normalise, with every intermediate resident in HBM
(BASELINE.json configs[4]).

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
from .data import istft_from_tiles, stft_to_tiles


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
                      peak: float | None = 0.9, max_batch: int = 256, precision: str | None = None):
    """float32 samples (n,) or (channels, n) on the GPU -> separated samples (hop*(T-1),) or (channels, hop*(T-1)).
    All channels go through ONE forward transform (which writes network tiles and frame-major phasors directly), one
    batched network forward per `max_batch` tiles and ONE inverse transform (which applies the mask on load and
    overlap-adds in LDS); the only other passes are the two per-channel normalisations."""
    squeeze = y.dim() == 1
    if squeeze:
        y = y[None]
    tiles, phase, norm, T = stft_to_tiles(y, n_fft, hop, INPUT_LEN)
    L = _lib.lib()
    C, n_tiles = tiles.shape[:2]
    for c in range(C):                                           # divide by the channel's maximum magnitude (data.py:84-85,105)
        _lib.check(L.svs_scale_by_inv(tiles[c].data_ptr(), tiles[c].numel(), norm[c:].data_ptr(), 1.0, _lib.stream_ptr()), "svs_scale_by_inv")
    flat = tiles.view(C * n_tiles, 1, tiles.shape[3], tiles.shape[4])
    mask = torch.empty_like(flat)
    was_training, was_precision = model.training, model.eval_precision
    try:
        model.eval()
        if precision is not None:                                # "bf16": the convolutions run on the bf16 MFMA (configs[4])
            model.eval_precision = precision
        for s in range(0, flat.shape[0], max_batch):
            mask[s:s + max_batch] = model(flat[s:s + max_batch])
    finally:                                                     # a failing forward must not leave the model in another mode
        model.eval_precision = was_precision
        model.train(was_training)
    out = istft_from_tiles(tiles, mask, phase, T, invert=not vocal_solo, n_fft=n_fft, hop=hop, peak=peak)
    return out[0] if squeeze else out
