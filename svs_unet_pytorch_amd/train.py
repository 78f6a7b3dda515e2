"""Training driver -- same command line, checkpoint / log file names and dataset semantics as the
reference's train.py (/root/reference/train.py:157-171, 65-143, 244-387), with the optimisation step
(train.py:265-300, L1 terms) running as one fused library call per step on each GPU.

    python -m svs_unet_pytorch_amd.train --train_folder spec/train --valid_folder spec/valid --label run1 \
        --batch_size 64 --epoch 400 --val_interval 10 [--load_path CKPT/svs_run1.pth]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m svs_unet_pytorch_amd.train ...

What is kept: flags (--train_folder --load_path --label(required) --epoch --batch_size --valid_folder
--val_interval), `SpectrogramDataset` (file pairing, len = songs x SAMPLES_PER_SONG, DC-row drop, shared
random 128-frame crop or right zero-pad, phase as np.angle), LR drop to 5e-4 with a snapshot at epoch 400
(train.py:251-262), per-epoch checkpoint CKPT/svs_<label>.pth with keys {epoch, model_state_dict, optim,
scheduler, loss_list_*} (train.py:369-382), best-validation checkpoint CKPT/svs_best_<label>.pth via
model.save (train.py:353-355), LOG/log_<label>.txt with one float per epoch and `Val <float>` lines
(train.py:314,350,357-363).

What is different: the loss is alpha_L1 * (L1 vocal + L1 accompaniment) (train.py:281-283,296 with
model.crit = nn.L1Loss).  The reference adds alpha_MR * MultiResolutionSTFTLoss from the `auraloss` package
(train.py:287-296), which is outside this round's scope -- the logged totals are
therefore the L1 part only.  With WORLD_SIZE > 1 the batch is sharded over ranks and gradients are
all-reduced over RCCL (parallel.py); rank 0 writes the files.  CKPT/ and LOG/ are created if missing (the
reference assumes they exist).
"""
from __future__ import annotations

import argparse
import os
import random

import numpy as np
import torch
import torch.utils.data as Data

from .config import INPUT_LEN, SAMPLES_PER_SONG
from .model import ALPHA_L1, UNet


class SpectrogramDataset(Data.Dataset):
    """train.py:65-143.  Returns (mix, voc, mix_phase, voc_phase), each float32 (1, 512, INPUT_LEN)."""

    def __init__(self, path, samples_per_song=SAMPLES_PER_SONG):
        self.path = path
        self.mixture_path = os.path.join(path, "mixture")
        self.vocal_path = os.path.join(path, "vocal")
        self.samples_per_song = samples_per_song
        if not os.path.exists(self.mixture_path):
            raise FileNotFoundError(f"Mixture folder not found: {self.mixture_path}")     # train.py:72-73
        names = sorted(f for f in os.listdir(self.mixture_path) if f.endswith("_spec.npy"))
        self.file_names = [f for f in names if os.path.exists(os.path.join(self.vocal_path, f))]
        print(f"[{os.path.basename(path)}] {len(self.file_names)} songs x {self.samples_per_song} samples = {len(self)} items.")

    def __len__(self):
        return len(self.file_names) * self.samples_per_song

    def __getitem__(self, idx):
        name = self.file_names[idx % len(self.file_names)]
        pname = name.replace("_spec.npy", "_phase.npy")
        mix = np.load(os.path.join(self.mixture_path, name))[1:, :]                      # drop the DC row (train.py:109-112)
        voc = np.load(os.path.join(self.vocal_path, name))[1:, :]
        mix_phase = np.angle(np.load(os.path.join(self.mixture_path, pname))).astype(np.float32)[1:, :]
        voc_phase = np.angle(np.load(os.path.join(self.vocal_path, pname))).astype(np.float32)[1:, :]
        target, cur = INPUT_LEN, mix.shape[1]
        if cur > target:
            start = random.randint(0, cur - target)                                       # shared start (train.py:121)
            mix, voc = mix[:, start:start + target], voc[:, start:start + target]
            mix_phase, voc_phase = mix_phase[:, start:start + target], voc_phase[:, start:start + target]
        else:
            pad = ((0, 0), (0, target - cur))                                             # train.py:129-135
            mix, voc = np.pad(mix, pad), np.pad(voc, pad)
            mix_phase, voc_phase = np.pad(mix_phase, pad), np.pad(voc_phase, pad)
        as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a[np.newaxis], dtype=np.float32))
        return as_t(mix), as_t(voc), as_t(mix_phase), as_t(voc_phase)


def epoch_order(n, shuffle=True, rank=0, world=1, epoch=0):
    """Item indices of one epoch for one rank -- DataLoader(shuffle=True) on one process, DistributedSampler on several:
    one permutation of all n items (the same on every rank: seeded by the epoch), padded by wrapping around to a multiple
    of `world`, rank r taking positions r, r + world, ..."""
    if shuffle:
        gen = torch.Generator()
        gen.manual_seed(epoch if world > 1 else random.getrandbits(31))
        order = torch.randperm(n, generator=gen).tolist()
    else:
        order = list(range(n))
    if world > 1:
        total = (n + world - 1) // world * world
        order = (order + order[:total - n])[rank:total:world]
    return order


class ResidentSpectrograms:
    """The training set kept in HBM: every song's mixture / vocal magnitude (DC row dropped) is uploaded once, and a batch
    is cut out of them by one kernel (`svs_crop_tiles`) -- no worker processes, no per-step host-to-device copies of
    tiles.  Same items and cropping rule as SpectrogramDataset (train.py:86-143): `samples_per_song` items per song,
    item idx -> song idx % n_songs, one random start shared by mixture and vocal, right zero-padding for short songs.
    A MUSDB18-sized set is ~1 GB of the 288 GB."""

    def __init__(self, dataset: SpectrogramDataset, device):
        self.n_songs = len(dataset.file_names)
        self.samples_per_song = dataset.samples_per_song
        self.device = device
        mix_parts, voc_parts, offsets, frames, off = [], [], [], [], 0
        for name in dataset.file_names:
            mix = np.load(os.path.join(dataset.mixture_path, name))[1:, :]                   # drop the DC row (train.py:109-112)
            voc = np.load(os.path.join(dataset.vocal_path, name))[1:, :]
            if voc.shape != mix.shape:
                raise ValueError(f"{name}: mixture {mix.shape} and vocal {voc.shape} differ")
            mix_parts.append(np.ascontiguousarray(mix, dtype=np.float32).reshape(-1))
            voc_parts.append(np.ascontiguousarray(voc, dtype=np.float32).reshape(-1))
            offsets.append(off)
            frames.append(mix.shape[1])
            off += mix.size
        self.rows = 512
        empty = np.zeros(0, np.float32)
        self.mix = torch.from_numpy(np.concatenate(mix_parts) if mix_parts else empty).to(device)
        self.voc = torch.from_numpy(np.concatenate(voc_parts) if voc_parts else empty).to(device)
        self.frames_host = list(frames)
        self.offsets = torch.tensor(offsets, dtype=torch.int64, device=device)
        self.frames = torch.tensor(frames, dtype=torch.int32, device=device)

    def __len__(self):
        return self.n_songs * self.samples_per_song

    def draw(self, items, rng=random):
        """(song, start) per item: the host side of __getitem__ (train.py:115-127)."""
        songs = [int(i) % self.n_songs for i in items]
        starts = [rng.randint(0, self.frames_host[s] - INPUT_LEN) if self.frames_host[s] > INPUT_LEN else 0 for s in songs]
        return songs, starts

    def crop(self, songs, starts):
        """mix, voc (B, 1, 512, INPUT_LEN) on the device for the given songs / start frames."""
        from . import _lib
        B = len(songs)
        idx = torch.tensor([songs, starts], dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)
        mix = torch.empty((B, 1, self.rows, INPUT_LEN), dtype=torch.float32, device=self.device)
        voc = torch.empty_like(mix)
        _lib.check(_lib.lib().svs_crop_tiles(self.mix.data_ptr(), self.voc.data_ptr(), self.offsets.data_ptr(), self.frames.data_ptr(),
                                             idx[0].data_ptr(), idx[1].data_ptr(), B, self.rows, INPUT_LEN, mix.data_ptr(), voc.data_ptr(),
                                             _lib.stream_ptr()), "svs_crop_tiles")
        return mix, voc

    def batches(self, batch_size, shuffle=True, rank=0, world=1, epoch=0):
        """One epoch of (mix, voc) batches: DataLoader(shuffle) / DistributedSampler semantics -- a permutation of all items
        (the same on every rank, seeded by the epoch), padded to a multiple of `world`, rank r taking items r, r+world, ...;
        the last batch may be short."""
        order = epoch_order(len(self), shuffle, rank, world, epoch)
        for i in range(0, len(order), batch_size):
            yield self.crop(*self.draw(order[i:i + batch_size]))

    def num_batches(self, batch_size, world=1):
        per_rank = (len(self) + world - 1) // world
        return (per_rank + batch_size - 1) // batch_size


def l1_terms(model, mix, voc):
    """Eval-mode loss of train.py:329-338 (no gradient)."""
    mask = model(mix)
    return model.crit(mask * mix, voc) + model.crit((1 - mask) * mix, torch.clamp(mix - voc, min=0.0))


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--train_folder", type=str, default="./data/vocals")
    parser.add_argument("--load_path", type=str, default="result.pth")
    parser.add_argument("--label", type=str, required=True)
    parser.add_argument("--epoch", type=int, default=2)
    parser.add_argument("--batch_size", type=int, default=2)
    parser.add_argument("--valid_folder", type=str, default="unet_spectrograms/valid")
    parser.add_argument("--val_interval", type=int, default=20)
    args = parser.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs a ROCm device (hand-written gfx950 kernels, no CPU path).")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    print(f"Using device: {device}")
    grad_sync = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from .parallel import init_process_group
        init_process_group(rank, world, device)

    log_file = f"LOG/log_{args.label}.txt"
    best_weight = f"CKPT/svs_best_{args.label}.pth"
    ckpt_weight = f"CKPT/svs_{args.label}.pth"
    if rank == 0:
        os.makedirs("LOG", exist_ok=True)
        os.makedirs("CKPT", exist_ok=True)

    train_dataset = SpectrogramDataset(args.train_folder)
    sampler = train_loader = resident = None
    per_rank_batch = max(args.batch_size // world, 1)
    if os.environ.get("SVS_DATA_LOADER", "resident") == "resident":
        resident = ResidentSpectrograms(train_dataset, device)            # the whole set lives in HBM (see the class)
    else:                                                                  # the reference's loader: 8 worker processes
        sampler = Data.distributed.DistributedSampler(train_dataset, world, rank, shuffle=True) if world > 1 else None
        train_loader = Data.DataLoader(train_dataset, batch_size=per_rank_batch, num_workers=8,
                                       shuffle=sampler is None, sampler=sampler, pin_memory=True)
    valid_loader = None
    if os.path.exists(args.valid_folder):
        valid_dataset = SpectrogramDataset(args.valid_folder)
        if len(valid_dataset) > 0:
            valid_loader = Data.DataLoader(valid_dataset, batch_size=max(args.batch_size // world, 1), num_workers=2,
                                           shuffle=False, pin_memory=True)
    else:
        print(f"Warning: validation folder {args.valid_folder} not found, validation is skipped.")   # train.py:199-200

    model = UNet().to(device)
    start_epoch = 0
    scheduler = None
    if os.path.exists(args.load_path):                                    # train.py:205-237
        model.load(args.load_path)
        checkpoint = torch.load(args.load_path, map_location=device)
        model.load_state_dict(checkpoint["model_state_dict"])
        if "optim" in checkpoint:
            model.optim.load_state_dict(checkpoint["optim"])
        start_epoch = checkpoint.get("epoch", 0)
        model.dropout_step = int(checkpoint.get("dropout_step", model.optim._step))      # Dropout2d masks continue, not replay
        for key in checkpoint:
            if key.startswith("loss_list"):
                setattr(model, key, checkpoint[key])
        print(f"Loaded checkpoint from {args.load_path}")
    if world > 1:
        from .parallel import GradAllReduce, broadcast_parameters
        broadcast_parameters(model, 0)
        grad_sync = GradAllReduce(model)

    best_val_loss = 100.0
    log_buffer = []
    print(f"Start training for {args.epoch - start_epoch} epochs...")
    for ep in range(start_epoch, args.epoch):
        model.train()
        if sampler is not None:
            sampler.set_epoch(ep)
        if ep == 400:                                                      # train.py:251-262
            for group in model.optim.param_groups:
                group["lr"] = 5e-4
            if rank == 0:
                torch.save({"epoch": ep + 1, "model_state_dict": model.state_dict(), "optim": model.optim.state_dict(),
                            "scheduler": None}, f"CKPT/svs_{args.label}_400.pth")
            print(f"\n[Info] Epoch {ep}: learning rate manually changed to 5e-4!\n")
        loss_sum = torch.zeros((), device=device)
        if resident is not None:
            steps = resident.num_batches(per_rank_batch, world)
            stream = resident.batches(per_rank_batch, True, rank, world, ep)
        else:
            steps = len(train_loader)
            stream = ((m.to(device, non_blocking=True), v.to(device, non_blocking=True)) for m, v, _a, _b in train_loader)
        for mix, voc in stream:
            l1 = model.train_step(mix, voc, loss_scale=ALPHA_L1, grad_sync=grad_sync)      # train.py:271-300, L1 terms
            loss_sum += ALPHA_L1 * l1                                                       # no host sync per step
        avg_train_loss = float(loss_sum) / max(steps, 1)
        log_buffer.append(f"{avg_train_loss}\n")

        if valid_loader and (ep + 1) % args.val_interval == 0:             # train.py:317-363
            model.eval()
            val_sum = 0.0
            with torch.no_grad():
                for mix, voc, _a, _b in valid_loader:
                    mix, voc = mix.to(device), voc.to(device)
                    val_sum += ALPHA_L1 * float(l1_terms(model, mix, voc))
            avg_val_loss = val_sum / len(valid_loader)
            log_buffer.append(f"Val {avg_val_loss}\n")
            print(f"\n[Epoch {ep + 1}] Train Loss: {avg_train_loss:.4e} | Val Loss: {avg_val_loss:.4e}")
            if avg_val_loss < best_val_loss and rank == 0:
                best_val_loss = avg_val_loss
                model.save(best_weight)
            if rank == 0:
                try:
                    with open(log_file, "a") as f:
                        f.writelines(log_buffer)
                    log_buffer = []
                except Exception as e:
                    print(f"Log write failed: {e}")
        else:
            print(f"Epoch {ep + 1} Avg Loss: {avg_train_loss:.4e}")

        if rank == 0:                                                      # train.py:369-382
            checkpoint = {"epoch": ep + 1, "model_state_dict": model.state_dict(), "optim": model.optim.state_dict(),
                          "scheduler": scheduler.state_dict() if scheduler is not None else None,
                          "dropout_step": model.dropout_step}
            for key in model.__dict__:
                if key.startswith("loss_list"):
                    checkpoint[key] = getattr(model, key)
            torch.save(checkpoint, ckpt_weight)

    if log_buffer and rank == 0:
        with open(log_file, "a") as f:
            f.writelines(log_buffer)
    print("Finish training!")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
