"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

Restatement of the reference's inference tile bookkeeping (/root/reference/inference.py:65-127)
in plain Python/numpy loops.  Integer work: the product must match it bit for bit.

  drop the DC row                      inference.py:68
  num_segments = T // seg_len + 1      inference.py:75
  segment i = columns [i*L, i*L+L)     inference.py:80-84
  skip empty segments (T % L == 0)     inference.py:88
  right zero-pad short segments to L   inference.py:90-92
  mask (or 1-mask) times mixture       inference.py:100-107
  crop the padding off again           inference.py:113-114
  concatenate along time               inference.py:120
  put a zero float32 row back on top   inference.py:123
"""
from __future__ import annotations

import numpy as np


def segment_plan(n_frames: int, seg_len: int = 128):
    """[(start, end, pad)] for every non-empty segment, in order."""
    plan = []
    num_segments = n_frames // seg_len + 1
    for i in range(num_segments):
        start = i * seg_len
        end = min(start + seg_len, n_frames)
        cur = end - start
        if cur <= 0:
            continue
        plan.append((start, end, seg_len - cur))
    return plan


def separate(mix_spec: np.ndarray, mask_fn, seg_len: int = 128, vocal_solo: bool = True) -> np.ndarray:
    """(513, T) float32 mixture magnitude -> (513, T) float32 separated magnitude.
    `mask_fn(tile)` maps a (1, 1, 512, seg_len) float32 array to the soft mask of the same shape."""
    crop = mix_spec[1:, :]
    pieces = []
    for start, end, pad in segment_plan(crop.shape[1], seg_len):
        seg = crop[:, start:end]
        if pad:
            seg = np.concatenate([seg, np.zeros((seg.shape[0], pad), seg.dtype)], axis=1)
        tile = np.ascontiguousarray(seg[None, None]).astype(np.float32)
        msk = np.asarray(mask_fn(tile), dtype=np.float32)
        if not vocal_solo:
            msk = np.float32(1.0) - msk
        pred = (tile * msk)[0, 0]
        if pad:
            pred = pred[:, : end - start]
        pieces.append(pred)
    if not pieces:
        return None
    full = np.concatenate(pieces, axis=1)
    return np.concatenate([np.zeros((1, full.shape[1]), np.float32), full], axis=0)


def crop_item(mix_file, voc_file, start, seg_len=128):
    """SpectrogramDataset.__getitem__ of the reference (train.py:86-143), magnitudes, with the random start passed in
    (the reference draws it with ONE random.randint(0, T - seg_len) per item, train.py:121, shared by all four
    outputs): rows 1.. of the (513, T) files, columns [start, start + seg_len) when T > seg_len, else the whole song
    right-padded with zeros (train.py:129-135).  Returns mix, voc of shape (1, 512, seg_len) float32.
    PINNED: oracle/gen_golden_train.py runs the reference's train.py as a script and asserts that its
    SpectrogramDataset returns exactly these arrays (tests/golden/dataset_items.npz holds the hashes)."""
    mix, voc = np.asarray(mix_file)[1:, :], np.asarray(voc_file)[1:, :]
    T = mix.shape[1]
    if T > seg_len:
        mix, voc = mix[:, start:start + seg_len], voc[:, start:start + seg_len]
    else:
        pad = ((0, 0), (0, seg_len - T))
        mix, voc = np.pad(mix, pad), np.pad(voc, pad)
    return mix[np.newaxis].astype(np.float32), voc[np.newaxis].astype(np.float32)


def crop_phase(phase_file, start, seg_len=128):
    """The phase half of the same item (train.py:103-112,124-135): np.angle of the complex64 unit-phasor file as
    float32, DC row dropped, the SAME start, right zero-padding.  Returns (1, 512, seg_len) float32.  Pinned as above."""
    ang = np.angle(np.asarray(phase_file)).astype(np.float32)[1:, :]
    T = ang.shape[1]
    if T > seg_len:
        ang = ang[:, start:start + seg_len]
    else:
        ang = np.pad(ang, ((0, 0), (0, seg_len - T)))
    return ang[np.newaxis].astype(np.float32)
