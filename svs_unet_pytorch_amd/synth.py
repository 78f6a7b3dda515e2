"""Counter-based synthetic data: tiles, audio and closed-form checkpoints.

Every benchmark, golden vector and parity test in this repo draws its inputs from one
stateless generator, so that the container that makes the fixtures (which can import the
reference) and the GPU box (which cannot) regenerate bit-identical tensors without files
and without depending on a torch RNG version.

    u32(seed, idx)  = mix32(lo(idx) + 0x9E3779B9 * mix32(hi(idx) + 0x85EBCA6B * mix32(seed + 1)))
    uniform(seed, idx) = (u32 >> 8) * 2^-24            in [0, 1), exactly representable in f32

`mix32` is the 32-bit finaliser x ^= x>>16; x *= 0x7FEB352D; x ^= x>>15; x *= 0x846CA68B;
x ^= x>>16.  The same function is implemented on the device in csrc/synth.hip
(`svs_fill_uniform`); tests/test_synth.py pins the two against each other.

Data contract being imitated: magnitudes are normalised into [0, 1] by the
mixture maximum (reference data.py:84-85,105) and 0 <= vocal <~ mixture, so
mix = u(seed 0), voc = mix * u(seed 1); element index = flat NCHW offset + 2^32 * tile index.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

SEED_MIX = 0
SEED_VOC_RATIO = 1
SEED_AUDIO = 2
SEED_WEIGHTS = 1234

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_G1 = np.uint32(0x9E3779B9)
_G2 = np.uint32(0x85EBCA6B)


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= _M1
    x ^= x >> np.uint32(15)
    x *= _M2
    x ^= x >> np.uint32(16)
    return x


def u32(seed: int, idx: np.ndarray) -> np.ndarray:
    """32 random bits for every 64-bit counter in `idx` (uint64 array)."""
    idx = np.asarray(idx, dtype=np.uint64)
    lo = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (idx >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        s = _mix32(np.asarray([(seed + 1) & 0xFFFFFFFF], dtype=np.uint32))[0]
        h = _mix32(hi + _G2 * s)
        return _mix32(lo + _G1 * h)


def uniform(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """`n` float32 values in [0, 1) for counters offset .. offset+n-1."""
    idx = np.arange(n, dtype=np.uint64) + np.uint64(offset)
    return (u32(seed, idx) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def tiles(batch: int, height: int = 512, width: int = 128, first_tile: int = 0):
    """(mix, voc) float32 arrays of shape (batch, 1, height, width)."""
    per = height * width
    mix = np.empty((batch, 1, height, width), np.float32)
    voc = np.empty_like(mix)
    for b in range(batch):
        off = (first_tile + b) << 32
        m = uniform(SEED_MIX, per, off)
        r = uniform(SEED_VOC_RATIO, per, off)
        mix[b, 0] = m.reshape(height, width)
        voc[b, 0] = (m * r).reshape(height, width)
    return mix, voc


def song(n: int, frames: int):
    """Synthetic spectrogram files of song `n` as data.py writes them (data.py:107-109): (mix, voc) float32 (513, frames)
    magnitudes with 0 <= voc <= mix <= 1, and (mix_phase, voc_phase) complex64 unit phasors."""
    off = (300 + n) << 32
    mix = uniform(SEED_MIX, 513 * frames, off).reshape(513, frames)
    voc = (mix * uniform(SEED_VOC_RATIO, 513 * frames, off).reshape(513, frames)).astype(np.float32)
    pm = (uniform(5, 513 * frames, off).reshape(513, frames).astype(np.float64) * 2 - 1) * np.pi
    pv = (uniform(6, 513 * frames, off).reshape(513, frames).astype(np.float64) * 2 - 1) * np.pi
    return mix, voc, np.exp(1j * pm).astype(np.complex64), np.exp(1j * pv).astype(np.complex64)


def audio(n_samples: int, channel: int = 0) -> np.ndarray:
    """float32 samples in [-1, 1): the config-5 synthetic stream (seed 2)."""
    return uniform(SEED_AUDIO, n_samples, channel << 32) * np.float32(2.0) - np.float32(1.0)


# --------------------------------------------------------------------------------------
# Closed-form checkpoint: the 79 state_dict entries of reference model.py:43-109
# --------------------------------------------------------------------------------------
ENC_CHANNELS = (1, 16, 32, 64, 128, 256, 512)           # model.py:47-76
DEC_IO = ((512, 256), (512, 128), (256, 64), (128, 32), (64, 16), (32, 1))  # model.py:79-109


def state_dict_spec():
    """Ordered (key, shape, kind) list in the reference's state_dict order."""
    spec = []
    for i in range(6):
        cin, cout = ENC_CHANNELS[i], ENC_CHANNELS[i + 1]
        p = f"conv{i + 1}"
        spec.append((f"{p}.0.weight", (cout, cin, 5, 5), "w"))
        spec.append((f"{p}.0.bias", (cout,), "b"))
        spec.append((f"{p}.1.weight", (cout,), "gamma"))
        spec.append((f"{p}.1.bias", (cout,), "beta"))
        spec.append((f"{p}.1.running_mean", (cout,), "rmean"))
        spec.append((f"{p}.1.running_var", (cout,), "rvar"))
        spec.append((f"{p}.1.num_batches_tracked", (), "nbt"))
    for i, (cin, cout) in enumerate(DEC_IO):
        p = f"deconv{i + 1}"
        spec.append((f"{p}.weight", (cin, cout, 5, 5), "wt"))
        spec.append((f"{p}.bias", (cout,), "b"))
        if i < 5:
            q = f"{p}_BAD.0"
            spec.append((f"{q}.weight", (cout,), "gamma"))
            spec.append((f"{q}.bias", (cout,), "beta"))
            spec.append((f"{q}.running_mean", (cout,), "rmean"))
            spec.append((f"{q}.running_var", (cout,), "rvar"))
            spec.append((f"{q}.num_batches_tracked", (), "nbt"))
    return spec


def closed_form_state(seed: int = SEED_WEIGHTS, trained_stats: bool = True):
    """OrderedDict key -> numpy array.  Conv weights/biases ~ U(-1/sqrt(fan_in), +) like
    torch's default init; BN gamma ~ U[0.5,1.5), beta ~ U[-0.1,0.1); with `trained_stats`
    the running statistics are non-trivial (mean ~ U[-0.1,0.1), var ~ U[0.5,1.5)) so that
    eval-mode folding is exercised, otherwise they are the fresh-module values (0, 1)."""
    out = OrderedDict()
    fan_in = 1.0
    for t, (key, shape, kind) in enumerate(state_dict_spec()):
        n = int(np.prod(shape)) if shape else 1
        u = uniform(seed + t, n)
        if kind == "w":           # Conv2d (Cout, Cin, 5, 5): fan_in = Cin*25
            fan_in = shape[1] * 25
            v = (u * 2 - 1) * np.float32(1.0 / math.sqrt(fan_in))
        elif kind == "wt":        # ConvTranspose2d (Cin, Cout, 5, 5): torch's fan_in = Cout*25
            fan_in = shape[1] * 25
            v = (u * 2 - 1) * np.float32(1.0 / math.sqrt(fan_in))
        elif kind == "b":
            v = (u * 2 - 1) * np.float32(1.0 / math.sqrt(fan_in))
        elif kind == "gamma":
            v = u + np.float32(0.5)
        elif kind == "beta":
            v = (u * 2 - 1) * np.float32(0.1)
        elif kind == "rmean":
            v = (u * 2 - 1) * np.float32(0.1) if trained_stats else np.zeros(n, np.float32)
        elif kind == "rvar":
            v = u + np.float32(0.5) if trained_stats else np.ones(n, np.float32)
        else:                     # num_batches_tracked
            out[key] = np.asarray(7 if trained_stats else 0, dtype=np.int64)
            continue
        out[key] = v.astype(np.float32).reshape(shape)
    return out


def dropout_masks(batch: int, seed: int, step: int = 0, rank: int = 0):
    """Per-(sample, channel) keep masks of the five decoder Dropout2d(0.5) layers
    (model.py:80-108): values in {0, 2} (survivors are scaled by 1/(1-p)).  Counter =
    (layer << 48) | (rank << 40) | (step << 20 ... ) is folded into the stream offset so
    that shards and steps draw independent masks."""
    masks = []
    for layer, (_, cout) in enumerate(DEC_IO[:5]):
        off = (layer << 56) | ((rank & 0xFF) << 48) | ((step & 0xFFFFFF) << 24)
        u = u32(seed, np.arange(batch * cout, dtype=np.uint64) + np.uint64(off))
        keep = (u >> np.uint32(31)).astype(np.float32) * np.float32(2.0)
        masks.append(keep.reshape(batch, cout))
    return masks
