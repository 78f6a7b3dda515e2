"""Inference driver -- same command line and file conventions as the reference's inference.py
(/root/reference/inference.py:29-34, 56-127), with the per-tile loop replaced by ONE batched forward.

    python -m svs_unet_pytorch_amd.inference --model_path CKPT/svs_x.pth --mixture_folder spec/mixture \
        --tar spec/pred [--vocal_solo 0|1]

Per file: load (513, T) -> drop the DC row (inference.py:68) -> cut into T//INPUT_LEN + 1 segments of
INPUT_LEN frames, skipping an empty last one and right-zero-padding a short one (inference.py:75-92)
-> mask = UNet(tile) (inference.py:100) -> optionally 1-mask (inference.py:102) -> mix*mask
(inference.py:107) -> crop the padding (inference.py:113-114) -> concatenate (inference.py:120) ->
put a zero float32 row back on top (inference.py:123) -> save under the same name (inference.py:126-127).
Tiles are independent in eval mode, so all segments of a file go through the network as one batch and
stay on the GPU until the finished spectrogram is copied back once (the reference copies every tile).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

from . import _lib
from .config import INPUT_LEN
from .model import UNet


def segment_plan(n_frames: int, seg_len: int = INPUT_LEN):
    """[(start, end, pad)] of the non-empty segments, in order (inference.py:75-92)."""
    plan = []
    for i in range(n_frames // seg_len + 1):
        start = i * seg_len
        end = min(start + seg_len, n_frames)
        if end > start:
            plan.append((start, end, seg_len - (end - start)))
    return plan


@torch.no_grad()
def separate(model: UNet, mix_spec, seg_len: int = INPUT_LEN, vocal_solo: bool = True, max_batch: int = 256):
    """(F+1, T) mixture magnitude (numpy or tensor) -> (F+1, T) float32 numpy array."""
    dev = model._flat.device
    spec = torch.as_tensor(np.asarray(mix_spec)).to(torch.float32)
    crop = spec[1:, :]
    F_, T = crop.shape
    plan = segment_plan(T, seg_len)
    if not plan:
        return None
    n = len(plan)
    padded = torch.zeros((F_, n * seg_len), dtype=torch.float32)
    padded[:, :T] = crop
    tiles = padded.view(F_, n, seg_len).permute(1, 0, 2).contiguous().unsqueeze(1).to(dev)     # (n,1,F,L)
    out = torch.empty_like(tiles)
    was_training = model.training
    model.eval()
    for s in range(0, n, max_batch):
        t = tiles[s:s + max_batch]
        mask = model(t)
        _lib.check(_lib.lib().svs_apply_mask(t.data_ptr(), mask.data_ptr(), out[s:s + max_batch].data_ptr(), t.numel(),
                                             0 if vocal_solo else 1, _lib.stream_ptr()), "svs_apply_mask")
    model.train(was_training)
    full = out[:, 0].permute(1, 0, 2).reshape(F_, n * seg_len)[:, :T].cpu()
    return torch.cat([torch.zeros((1, T), dtype=torch.float32), full], dim=0).numpy()


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--model_path", type=str, required=True)
    parser.add_argument("--tar", type=str, required=True)
    parser.add_argument("--mixture_folder", type=str, required=True)
    parser.add_argument("--vocal_solo", type=int, default=1, help="1: keep the vocal, 0: remove it")
    args = parser.parse_args(argv)

    os.makedirs(args.tar, exist_ok=True)
    if not torch.cuda.is_available():
        print("Inference needs a ROCm device (hand-written gfx950 kernels, no CPU path).")
        sys.exit(1)
    device = torch.device("cuda")
    print(f"Inference using device: {device}")

    model = UNet().to(device)
    try:
        checkpoint = torch.load(args.model_path, map_location=device)
        if isinstance(checkpoint, dict) and "model_state_dict" in checkpoint:
            model.load_state_dict(checkpoint["model_state_dict"])
    except Exception as e:      # inference.py:49-51
        print(f"Failed to load the model: {e}")
        sys.exit(1)
    model.eval()

    files = sorted(f for f in os.listdir(args.mixture_folder) if f.endswith("_spec.npy"))[:20]   # inference.py:58-59
    print(f"Found {len(files)} files, separating...")
    for name in files:
        mix = np.load(os.path.join(args.mixture_folder, name))
        pred = separate(model, mix, INPUT_LEN, bool(args.vocal_solo))
        if pred is not None:
            np.save(os.path.join(args.tar, name), pred)
    print("Separation finished!")


if __name__ == "__main__":
    main()
