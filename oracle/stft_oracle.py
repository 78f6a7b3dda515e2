"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (numpy, float64 arithmetic, results cast to the reference's dtypes) of the
signal front/back end of the reference: data.py `to_spec` (data.py:78-109) and `to_wave`
(data.py:151-166), and train.py `specific_istft` (train.py:33-60).

PARITY UNPINNED for the librosa half: the arithmetic lives in librosa 0.10.1
(/root/reference/uv.lock:713-714), which is neither vendored nor installable here, and the
reference holds no fixture for it.  This file restates librosa's published algorithm for
exactly the calls the reference makes:
  librosa.stft(y, n_fft=W, hop_length=H)        data.py:79,100   win_length=n_fft, periodic Hann,
                                                center=True, pad_mode='constant', complex64,
                                                frames = 1 + len(y)//H
  librosa.magphase(D)                           data.py:80,101   mag=|D|, phase=D/|D| (1+0j where 0)
  librosa.istft(S, win_length=W, hop_length=H)  data.py:159      n_fft=2*(rows-1), centred, divided by
                                                the window sum-of-squares where it exceeds tiny
The torch half IS pinned: tests/test_oracle_golden.py checks `istft` here against torch.istft with
the arguments of train.py:51-58 (the reference's own in-repo inverse) and `stft` against torch.stft.
"""
from __future__ import annotations

import numpy as np


def hann_periodic(n: int) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True) == torch.hann_window(n)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def n_frames(n_samples: int, hop: int) -> int:
    return 1 + n_samples // hop


def stft(y: np.ndarray, n_fft: int = 1024, hop: int = 768) -> np.ndarray:
    """(n_fft//2+1, 1+len(y)//hop) complex64, centred with zero padding."""
    y = np.asarray(y, dtype=np.float64)
    pad = n_fft // 2
    yp = np.concatenate([np.zeros(pad), y, np.zeros(pad)])
    w = hann_periodic(n_fft)
    t = n_frames(len(y), hop)
    frames = np.stack([yp[i * hop: i * hop + n_fft] * w for i in range(t)], axis=1)
    return np.fft.rfft(frames, axis=0).astype(np.complex64)


def magphase(d: np.ndarray):
    """mag float32, phase complex64 unit phasors (1+0j where the bin is exactly zero)."""
    mag = np.abs(d).astype(np.float32)
    zero = mag == 0
    safe = mag + zero
    phase = np.empty_like(d, dtype=np.complex64)
    phase.real = d.real / safe + zero
    phase.imag = d.imag / safe
    return mag, phase


def to_spec(y_mix: np.ndarray, y_track: np.ndarray, n_fft: int = 1024, hop: int = 768):
    """data.py:78-109 for one track: align the track to the mixture length (data.py:97-98),
    STFT both, divide the track magnitude by the MIXTURE's maximum (data.py:84-85,105)."""
    s_mix, _ = magphase(stft(y_mix, n_fft, hop))
    norm = s_mix.max()
    if norm == 0:
        norm = 1
    if len(y_track) > len(y_mix):
        y_track = y_track[: len(y_mix)]
    else:
        y_track = np.pad(y_track, (0, len(y_mix) - len(y_track)))
    spec, phase = magphase(stft(y_track, n_fft, hop))
    return (spec / np.float32(norm)).astype(np.float32), phase


def window_sumsquare(n_frames_: int, n_fft: int, hop: int) -> np.ndarray:
    w2 = hann_periodic(n_fft) ** 2
    out = np.zeros(n_fft + hop * (n_frames_ - 1))
    for i in range(n_frames_):
        out[i * hop: i * hop + n_fft] += w2
    return out


def istft(s: np.ndarray, n_fft: int = 1024, hop: int = 768) -> np.ndarray:
    """(n_fft//2+1, T) complex -> hop*(T-1) float32 samples (centred; both n_fft//2 edges trimmed)."""
    s = np.asarray(s, dtype=np.complex128)
    t = s.shape[1]
    w = hann_periodic(n_fft)
    y = np.zeros(n_fft + hop * (t - 1))
    frames = np.fft.irfft(s, n=n_fft, axis=0) * w[:, None]
    for i in range(t):
        y[i * hop: i * hop + n_fft] += frames[:, i]
    env = window_sumsquare(t, n_fft, hop)
    ok = env > np.finfo(np.float32).tiny
    y[ok] /= env[ok]
    return y[n_fft // 2: n_fft // 2 + hop * (t - 1)].astype(np.float32)


def to_wave(mag: np.ndarray, phase: np.ndarray, n_fft: int = 1024, hop: int = 768) -> np.ndarray:
    """data.py:151-164: common frame count, istft(mag*phase), peak-normalise to 0.9."""
    m = min(mag.shape[1], phase.shape[1])
    y = istft(mag[:, :m] * phase[:, :m], n_fft, hop)
    peak = np.max(np.abs(y)) if y.size else 0.0
    if peak > 0:
        y = (y / peak * np.float32(0.9)).astype(np.float32)
    return y


def specific_istft(magnitude: np.ndarray, phase: np.ndarray, n_fft: int = 1024, hop: int = 768) -> np.ndarray:
    """train.py:33-60: (B,1,512,T) magnitude and angle -> (B,1,hop*(T-1)) waveform.
    A zero row is put back at DC (train.py:41-42), polar -> complex (train.py:45), istft with a
    periodic Hann window (train.py:51-58)."""
    b = magnitude.shape[0]
    out = []
    for i in range(b):
        mag = np.concatenate([np.zeros((1, magnitude.shape[-1])), magnitude[i, 0].astype(np.float64)], axis=0)
        ang = np.concatenate([np.zeros((1, phase.shape[-1])), phase[i, 0].astype(np.float64)], axis=0)
        out.append(istft(mag * np.exp(1j * ang), n_fft, hop))
    return np.stack(out)[:, None, :]


def specific_istft_adjoint(d_wav: np.ndarray, phase: np.ndarray, n_fft: int = 1024, hop: int = 768) -> np.ndarray:
    """Gradient of specific_istft with respect to `magnitude`: (B,1,hop*(T-1)) d(loss)/d(wav) -> (B,1,512,T) float64.
    The map magnitude -> waveform is linear, so this is its transpose (what autograd does to train.py:33-60):
    divide by the window envelope, re-frame with the window (transpose of overlap-add), transpose of irfft
    (G[k] = c_k/N * rfft(frame)[k], c_k = 1 for DC / Nyquist whose imaginary parts irfft ignores, else 2), then
    d|S| = Re(conj(e^{i*phase}) * G); the DC row that train.py:41-42 pads in is dropped again."""
    d_wav = np.asarray(d_wav, dtype=np.float64)
    b, t = d_wav.shape[0], phase.shape[-1]
    w = hann_periodic(n_fft)
    env = window_sumsquare(t, n_fft, hop)
    ok = env > np.finfo(np.float32).tiny
    ck = np.full(n_fft // 2 + 1, 2.0)
    ck[0] = ck[-1] = 1.0
    out = np.zeros((b, 1, n_fft // 2, t))
    for i in range(b):
        g = np.zeros(n_fft + hop * (t - 1))
        g[n_fft // 2: n_fft // 2 + hop * (t - 1)] = d_wav[i, 0]
        g[ok] /= env[ok]
        frames = np.stack([g[j * hop: j * hop + n_fft] * w for j in range(t)], axis=1)
        G = np.fft.rfft(frames, axis=0) * (ck / n_fft)[:, None]
        ang = np.concatenate([np.zeros((1, t)), phase[i, 0].astype(np.float64)], axis=0)
        dm = G.real * np.cos(ang) + G.imag * np.sin(ang)
        dm[-1] = G.real[-1] * np.cos(ang[-1])                 # Nyquist: only the real part of S enters irfft
        out[i, 0] = dm[1:]
    return out
