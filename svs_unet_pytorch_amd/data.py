"""wav <-> spectrogram conversion -- same command line, folder layout and file names as the reference's
data.py (/root/reference/data.py:20-28, 46-169), with the STFT / magnitude-phase split / inverse STFT /
normalisations running in the gfx950 kernels of csrc/stft.hip instead of librosa.

    python -m svs_unet_pytorch_amd.data --src MUSDB/train --tar spec/train --direction to_spec
    python -m svs_unet_pytorch_amd.data --src spec/pred --phase spec/test/mixture --tar wav --direction to_wave

to_spec (data.py:46-112): per song folder, mixture.wav fixes the normalisation (its maximum magnitude,
data.py:84-85); mixture.wav and vocals.wav are STFT'd (n_fft=--win_size, hop=--hop_size, periodic Hann,
centred), divided by that maximum (data.py:105) and saved as NNNN_<song>_spec.npy (float32 (513,T)) and
NNNN_<song>_phase.npy (complex64 unit phasors) under <tar>/mixture and <tar>/vocal (data.py:107-109).
to_wave (data.py:117-169): <name>_spec.npy times its phase -> inverse STFT -> peak-normalise to 0.9 ->
<name>.wav at --sr.

Out of the accelerated path: wav decoding + resampling to --sr (the
reference uses librosa.load / soxr; here scipy.io.wavfile + scipy.signal.resample_poly on the CPU) and
wav encoding (soundfile there, scipy.io.wavfile here).  Resampled audio therefore matches the reference
only up to the resampler's filter; everything after the resampler follows the reference's arithmetic.
"""
from __future__ import annotations

import argparse
import os
import sys
from fractions import Fraction

import numpy as np
import torch

from . import _lib
from .config import HOP_SIZE, SAMPLE_RATE, WINDOW_SIZE, num2str

TRACK_MAP = {"mixture.wav": "mixture", "vocals.wav": "vocal"}      # data.py:40-43


# ------------------------------------------------------------------------------------------------
# GPU signal path
# ------------------------------------------------------------------------------------------------
def stft_magphase(y: torch.Tensor, n_fft: int = WINDOW_SIZE, hop: int = HOP_SIZE):
    """float32 (n,) on the GPU -> (mag float32 (513,T), phase complex64 (513,T)), both on the GPU."""
    L = _lib.lib()
    y = y.contiguous().float()
    T = int(L.svs_stft_frames(y.numel(), hop))
    mag = torch.empty((n_fft // 2 + 1, T), dtype=torch.float32, device=y.device)
    ph = torch.empty((n_fft // 2 + 1, T, 2), dtype=torch.float32, device=y.device)
    _lib.check(L.svs_stft_fwd(y.data_ptr(), y.numel(), n_fft, hop, mag.data_ptr(), ph.data_ptr(), _lib.stream_ptr()), "svs_stft_fwd")
    return mag, torch.view_as_complex(ph)


def istft(mag: torch.Tensor, phase: torch.Tensor, n_fft: int = WINDOW_SIZE, hop: int = HOP_SIZE, peak: float | None = None):
    """mag float32 (513,T) and phase (complex64 unit phasors, or float32 angles) -> float32 (hop*(T-1),).
    `peak`: scale so that max|y| == peak (data.py:162-164); None leaves the amplitude alone."""
    L = _lib.lib()
    mag = mag.contiguous().float()
    T = mag.shape[1]
    is_angle = not torch.is_complex(phase)
    ph = phase.contiguous().float() if is_angle else torch.view_as_real(phase.contiguous().to(torch.complex64)).contiguous()
    y = torch.empty(hop * (T - 1), dtype=torch.float32, device=mag.device)
    ws = torch.empty(int(L.svs_istft_workspace_bytes(n_fft, hop, T)) + 4096, dtype=torch.uint8, device=mag.device)
    _lib.check(L.svs_istft(mag.data_ptr(), ph.data_ptr(), 1 if is_angle else 0, n_fft, hop, T, y.data_ptr(), ws.data_ptr(), ws.numel(),
                           _lib.stream_ptr()), "svs_istft")
    if peak is not None:
        pk = torch.empty(1, dtype=torch.float32, device=mag.device)
        _lib.check(L.svs_absmax(y.data_ptr(), y.numel(), pk.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "svs_absmax")
        _lib.check(L.svs_scale_by_inv(y.data_ptr(), y.numel(), pk.data_ptr(), float(peak), _lib.stream_ptr()), "svs_scale_by_inv")
    return y


def specific_istft(magnitude: torch.Tensor, phase: torch.Tensor, n_fft: int = WINDOW_SIZE, hop: int = HOP_SIZE):
    """train.py:33-60: (B,1,512,T) magnitude and angle (DC row dropped) -> (B,1,hop*(T-1)) waveforms, ONE launch for the
    whole batch (the DC row that train.py:41-42 pads back is the absent first row of the tile layout)."""
    B, _, F_, T = magnitude.shape
    m = magnitude.contiguous().float()
    a = phase.contiguous().float()
    out = torch.empty((B, 1, hop * (T - 1)), dtype=torch.float32, device=m.device)
    _lib.check(_lib.lib().svs_istft_tiles(m.data_ptr(), F_ * T, T, F_, 1, None, 0, a.data_ptr(), 3, B, n_fft, hop, T, out.data_ptr(), None,
                                          _lib.stream_ptr()), "svs_istft_tiles")
    return out


def stft_to_tiles(y: torch.Tensor, n_fft: int = WINDOW_SIZE, hop: int = HOP_SIZE, seg: int = 128):
    """float32 (channels, n) on the GPU -> (tiles (channels, n_tiles, 1, n_fft/2, seg) magnitude with the DC row dropped and the
    last tile zero-padded (inference.py:68,84-92), frame-major unit phasors (channels, T, n_fft/2+1) complex64, the maximum
    magnitude per channel (channels,) incl. the DC row (data.py:84), T).  One launch; nothing is repacked afterwards."""
    L = _lib.lib()
    y = y.contiguous().float()
    C, n = y.shape
    T = int(L.svs_stft_frames(n, hop))
    n_tiles = T // seg + (1 if T % seg else 0)                      # the empty last segment is skipped (inference.py:88)
    rows = n_fft // 2
    tiles = torch.empty((C, n_tiles, 1, rows, seg), dtype=torch.float32, device=y.device)
    phase = torch.empty((C, T, rows + 1, 2), dtype=torch.float32, device=y.device)
    groups = int(L.svs_stft_groups(n_tiles * seg))
    part = torch.empty((C, groups), dtype=torch.float32, device=y.device)
    _lib.check(L.svs_stft_tiles(y.data_ptr(), n, C, n_fft, hop, tiles.data_ptr(), n_tiles * rows * seg, seg, rows, 1, n_tiles * seg,
                                phase.data_ptr(), 1, part.data_ptr(), _lib.stream_ptr()), "svs_stft_tiles")
    peak = torch.empty(C, dtype=torch.float32, device=y.device)
    for c in range(C):
        _lib.check(L.svs_max(part[c].data_ptr(), groups, peak[c:].data_ptr(), _lib.stream_ptr()), "svs_max")
    return tiles, torch.view_as_complex(phase), peak, T


def istft_from_tiles(tiles: torch.Tensor, mask, phase_fm: torch.Tensor, frames: int, invert: bool = False,
                     n_fft: int = WINDOW_SIZE, hop: int = HOP_SIZE, peak: float | None = None):
    """(channels, n_tiles, 1, 512, seg) magnitude tiles [times mask or 1 - mask, fused: inference.py:100-107] and frame-major
    phasors (channels, T, 513) -> (channels, hop*(T-1)) samples, optionally peak-normalised per channel (data.py:162-164)."""
    L = _lib.lib()
    C, n_tiles, _, rows, seg = tiles.shape
    ph = torch.view_as_real(phase_fm.contiguous()).contiguous()
    y = torch.empty((C, hop * (frames - 1)), dtype=torch.float32, device=tiles.device)
    groups = int(L.svs_istft_groups(hop, frames, C))
    part = torch.empty((C, groups), dtype=torch.float32, device=tiles.device) if peak is not None else None
    _lib.check(L.svs_istft_tiles(tiles.data_ptr(), n_tiles * rows * seg, seg, rows, 1, None if mask is None else mask.data_ptr(),
                                 1 if invert else 0, ph.data_ptr(), 1, C, n_fft, hop, frames, y.data_ptr(),
                                 None if part is None else part.data_ptr(), _lib.stream_ptr()), "svs_istft_tiles")
    if peak is not None:
        pk = torch.empty(C, dtype=torch.float32, device=tiles.device)
        for c in range(C):
            _lib.check(L.svs_max(part[c].data_ptr(), groups, pk[c:].data_ptr(), _lib.stream_ptr()), "svs_max")
            _lib.check(L.svs_scale_by_inv(y[c].data_ptr(), y.shape[1], pk[c:].data_ptr(), float(peak), _lib.stream_ptr()), "svs_scale_by_inv")
    return y


# ------------------------------------------------------------------------------------------------
# host-side file glue (not on the accelerated path)
# ------------------------------------------------------------------------------------------------
def load_wav_mono(path: str, sr: int) -> np.ndarray:
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    rate, data = wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    else:
        data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1)
    if rate != sr:
        fr = Fraction(sr, rate)
        data = resample_poly(data, fr.numerator, fr.denominator).astype(np.float32)
    return np.ascontiguousarray(data, dtype=np.float32)


def write_wav(path: str, y: np.ndarray, sr: int):
    from scipy.io import wavfile
    wavfile.write(path, sr, np.asarray(y, dtype=np.float32))


def to_spec(args, device):
    os.makedirs(args.tar, exist_ok=True)
    for folder in TRACK_MAP.values():
        os.makedirs(os.path.join(args.tar, folder), exist_ok=True)
    print(f"Scanning source folder: {args.src}")
    songs = sorted(d for d in os.listdir(args.src) if os.path.isdir(os.path.join(args.src, d)))
    print(f"Found {len(songs)} song folders.")
    if not songs:
        print("Error: no song folders found, check --src.")
        sys.exit(1)                                                     # data.py:61-63
    ws = torch.empty(4096, dtype=torch.uint8, device=device)
    for audio_idx, song in enumerate(songs):
        song_path = os.path.join(args.src, song)
        mix_path = os.path.join(song_path, "mixture.wav")
        if not os.path.exists(mix_path):
            continue
        try:
            y_mix = load_wav_mono(mix_path, args.sr)
            spec_mix, _ = stft_magphase(torch.from_numpy(y_mix).to(device), args.win_size, args.hop_size)
            norm = torch.empty(1, dtype=torch.float32, device=device)     # max magnitude, 0 -> 1 (data.py:84-85)
            _lib.check(_lib.lib().svs_absmax(spec_mix.data_ptr(), spec_mix.numel(), norm.data_ptr(), ws.data_ptr(), ws.numel(),
                                             _lib.stream_ptr()), "svs_absmax")
            for wav_file, folder in TRACK_MAP.items():
                track = os.path.join(song_path, wav_file)
                if not os.path.exists(track):
                    continue
                y = load_wav_mono(track, args.sr)
                y = y[: len(y_mix)] if len(y) > len(y_mix) else np.pad(y, (0, len(y_mix) - len(y)))   # data.py:97-98
                spec, phase = stft_magphase(torch.from_numpy(y).to(device), args.win_size, args.hop_size)
                _lib.check(_lib.lib().svs_scale_by_inv(spec.data_ptr(), spec.numel(), norm.data_ptr(), 1.0, _lib.stream_ptr()),
                           "svs_scale_by_inv")
                base = f"{num2str(audio_idx)}_{song}"
                np.save(os.path.join(args.tar, folder, f"{base}_spec.npy"), spec.cpu().numpy())
                np.save(os.path.join(args.tar, folder, f"{base}_phase.npy"), phase.cpu().numpy())
        except Exception as e:                                          # data.py:111-112
            print(f"Error processing {song}: {e}")


def to_wave(args, device):
    if args.phase == "-1":
        raise Exception("--phase is required for to_wave")            # data.py:118
    os.makedirs(args.tar, exist_ok=True)
    files = sorted(f for f in os.listdir(args.src) if f.endswith("_spec.npy"))
    print(f"Restoring {len(files)} files...")
    for spec_name in files:
        try:
            mag = np.load(os.path.join(args.src, spec_name))
            phase_name = spec_name.replace("_spec.npy", "_phase.npy")
            phase = None
            for p in (os.path.join(args.phase, phase_name), os.path.join(args.phase, "mixture", phase_name)):   # data.py:135-143
                if os.path.exists(p):
                    phase = np.load(p)
                    break
            if phase is None:
                phase = np.exp(2j * np.pi * np.random.rand(*mag.shape))                                        # data.py:148
            m = min(mag.shape[1], phase.shape[1])                                                              # data.py:151-153
            y = istft(torch.from_numpy(np.ascontiguousarray(mag[:, :m])).to(device),
                      torch.from_numpy(np.ascontiguousarray(phase[:, :m]).astype(np.complex64)).to(device),
                      args.win_size, args.hop_size, peak=0.9)
            write_wav(os.path.join(args.tar, spec_name.replace("_spec.npy", ".wav")), y.cpu().numpy(), args.sr)
        except Exception as e:                                          # data.py:168-169
            print(f"Restore failed {spec_name}: {e}")


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--src", type=str, required=True, help="source folder (song folders, or *_spec.npy for to_wave)")
    parser.add_argument("--tar", type=str, required=True, help="target folder")
    parser.add_argument("--phase", type=str, default="-1", help="phase folder (to_wave only)")
    parser.add_argument("--win_size", type=int, default=WINDOW_SIZE)
    parser.add_argument("--hop_size", type=int, default=HOP_SIZE)
    parser.add_argument("--sr", type=int, default=SAMPLE_RATE)
    parser.add_argument("--direction", default="to_spec", choices=["to_spec", "to_wave"])
    args = parser.parse_args(argv)
    if args.win_size != WINDOW_SIZE:             # data.py:24 lets it vary; the gfx950 transforms are built for the config's 1024 only
        parser.error(f"--win_size {args.win_size}: the STFT / iSTFT kernels are built for n_fft = {WINDOW_SIZE} (config.WINDOW_SIZE) only; "
                     "--hop_size may be anything in 1..win_size")
    if not 0 < args.hop_size <= args.win_size:
        parser.error(f"--hop_size {args.hop_size}: must be in 1..{args.win_size} (a larger hop leaves samples that no frame covers)")
    if not torch.cuda.is_available():
        print("data.py needs a ROCm device (the STFT/iSTFT are gfx950 kernels, no CPU path).")
        sys.exit(1)
    device = torch.device("cuda")
    if args.direction == "to_spec":
        to_spec(args, device)
    else:
        to_wave(args, device)


if __name__ == "__main__":
    main()
