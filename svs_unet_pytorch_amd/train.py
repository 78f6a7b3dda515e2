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

The objective is the reference's: alpha_L1 * (L1 vocal + L1 accompaniment) + alpha_MR * MultiResolutionSTFTLoss of
the re-synthesised vocal waveforms (train.py:24-26,281-296 with model.crit = nn.L1Loss), all of it on the GPU
(csrc/mrstft.hip restates the `auraloss` definition, which is not installable here: parity unpinned for that term).
SVS_OBJECTIVE=l1 drops the MR term (the BASELINE "L1-loss step"); it is also dropped, with a warning, when the
data set has no *_phase.npy files.  The logged totals are the same quantity as the reference's.  With WORLD_SIZE > 1 the batch is sharded over ranks and gradients are
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
from .model import ALPHA_L1, ALPHA_MR, UNet


class SpectrogramDataset(Data.Dataset):
    """train.py:65-143.  Returns (mix, voc, mix_phase, voc_phase), each float32 (1, 512, INPUT_LEN)."""

    def __init__(self, path, samples_per_song=SAMPLES_PER_SONG, with_phase=True):
        self.path = path
        self.with_phase = with_phase          # False: L1-only objective or a set without *_phase.npy -> empty phase tensors
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
        if self.with_phase:
            mix_phase = np.angle(np.load(os.path.join(self.mixture_path, pname))).astype(np.float32)[1:, :]
            voc_phase = np.angle(np.load(os.path.join(self.vocal_path, pname))).astype(np.float32)[1:, :]
        else:                                                                             # same crop / pad arithmetic on nothing
            mix_phase = voc_phase = np.zeros((0, mix.shape[1]), np.float32)
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

    def has_phase_files(self):
        pname = lambda n: n.replace("_spec.npy", "_phase.npy")
        return all(os.path.exists(os.path.join(d, pname(n))) for n in self.file_names for d in (self.mixture_path, self.vocal_path))


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

    def __init__(self, dataset: SpectrogramDataset, device, with_phase: bool = True):
        self.n_songs = len(dataset.file_names)
        self.samples_per_song = dataset.samples_per_song
        self.device = device
        mix_parts, voc_parts, offsets, frames, off = [], [], [], [], 0
        ph_mix, ph_voc = [], []
        pname = lambda n: n.replace("_spec.npy", "_phase.npy")
        self.has_phase = with_phase and all(os.path.exists(os.path.join(d, pname(n))) for n in dataset.file_names
                                            for d in (dataset.mixture_path, dataset.vocal_path))
        for name in dataset.file_names:
            if self.has_phase:                       # unit phasors complex64 (513, T) -> rows 1.. (train.py:103-112)
                ph_mix.append(np.ascontiguousarray(np.load(os.path.join(dataset.mixture_path, pname(name)))[1:, :], dtype=np.complex64).reshape(-1))
                ph_voc.append(np.ascontiguousarray(np.load(os.path.join(dataset.vocal_path, pname(name)))[1:, :], dtype=np.complex64).reshape(-1))
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
        self.mix_ang = self.voc_ang = None
        if self.has_phase and mix_parts:             # np.angle on the device (train.py:103-104), kept resident like the magnitudes
            from . import _lib
            for attr, parts in (("mix_ang", ph_mix), ("voc_ang", ph_voc)):
                z = torch.view_as_real(torch.from_numpy(np.concatenate(parts)).to(device)).contiguous()
                ang = torch.empty(z.shape[0], dtype=torch.float32, device=device)
                _lib.check(_lib.lib().svs_phase_angle(z.data_ptr(), ang.data_ptr(), ang.numel(), _lib.stream_ptr()), "svs_phase_angle")
                setattr(self, attr, ang)

    def __len__(self):
        return self.n_songs * self.samples_per_song

    def draw(self, items, rng=random):
        """(song, start) per item: the host side of __getitem__ (train.py:115-127)."""
        songs = [int(i) % self.n_songs for i in items]
        starts = [rng.randint(0, self.frames_host[s] - INPUT_LEN) if self.frames_host[s] > INPUT_LEN else 0 for s in songs]
        return songs, starts

    def crop(self, songs, starts, with_phase: bool = False):
        """mix, voc [, mix_phase, voc_phase] (B, 1, 512, INPUT_LEN) on the device for the given songs / start frames; the
        phase tiles are angles cut with the SAME start (train.py:121-127)."""
        from . import _lib
        B = len(songs)
        idx = torch.tensor([songs, starts], dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)
        out = []
        pairs = [(self.mix, self.voc)] + ([(self.mix_ang, self.voc_ang)] if with_phase else [])
        for a_src, b_src in pairs:
            a = torch.empty((B, 1, self.rows, INPUT_LEN), dtype=torch.float32, device=self.device)
            b = torch.empty_like(a)
            _lib.check(_lib.lib().svs_crop_tiles(a_src.data_ptr(), b_src.data_ptr(), self.offsets.data_ptr(), self.frames.data_ptr(),
                                                 idx[0].data_ptr(), idx[1].data_ptr(), B, self.rows, INPUT_LEN, a.data_ptr(), b.data_ptr(),
                                                 _lib.stream_ptr()), "svs_crop_tiles")
            out += [a, b]
        return tuple(out)

    def batches(self, batch_size, shuffle=True, rank=0, world=1, epoch=0, with_phase: bool = False):
        """One epoch of (mix, voc) batches: DataLoader(shuffle) / DistributedSampler semantics -- a permutation of all items
        (the same on every rank, seeded by the epoch), padded to a multiple of `world`, rank r taking items r, r+world, ...;
        the last batch may be short."""
        order = epoch_order(len(self), shuffle, rank, world, epoch)
        for i in range(0, len(order), batch_size):
            yield self.crop(*self.draw(order[i:i + batch_size]), with_phase=with_phase)

    def num_batches(self, batch_size, world=1):
        per_rank = (len(self) + world - 1) // world
        return (per_rank + batch_size - 1) // batch_size


def l1_terms(model, mix, voc):
    """Eval-mode loss of train.py:329-338 (no gradient)."""
    mask = model(mix)
    return model.crit(mask * mix, voc) + model.crit((1 - mask) * mix, torch.clamp(mix - voc, min=0.0))


def mr_term(model, mix, voc, mix_phase, voc_phase):
    """Eval-mode MR-STFT term of train.py:341-343 (no gradient): MR(specific_istft(mask*mix, mix_phase), specific_istft(voc, voc_phase))."""
    from . import _lib
    from .config import HOP_SIZE, WINDOW_SIZE
    from .data import specific_istft
    mask = model(mix)
    pred = specific_istft(mask * mix, mix_phase, WINDOW_SIZE, HOP_SIZE)
    tgt = specific_istft(voc, voc_phase, WINDOW_SIZE, HOP_SIZE)
    B, L_ = pred.shape[0], pred.shape[-1]
    L = _lib.lib()
    ws = torch.empty(int(L.svs_mrstft_workspace_bytes(B, L_)), dtype=torch.uint8, device=mix.device)
    out = torch.empty(1, dtype=torch.float32, device=mix.device)
    _lib.check(L.svs_mrstft_loss_fwd_bwd(pred.data_ptr(), tgt.data_ptr(), B, L_, 0.0, out.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr()), "svs_mrstft_loss_fwd_bwd")
    return out[0]


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

    # objective: the reference's alpha_L1 * L1 + alpha_MR * MR-STFT (train.py:296) unless SVS_OBJECTIVE=l1 or no phases
    full = os.environ.get("SVS_OBJECTIVE", "full") != "l1"
    if full and not (resident.has_phase if resident is not None else train_dataset.has_phase_files()):
        print("Warning: no *_phase.npy files next to the spectrograms -- training on the L1 terms only.")
        full = False
    # the CPU-side datasets load phase files only when the objective needs them (and they exist): the L1-only fallback
    # holds for the validation loop and the SVS_DATA_LOADER=cpu path too, not just for the resident loader
    train_dataset.with_phase = full
    if valid_loader is not None:
        valid_dataset.with_phase = full and valid_dataset.has_phase_files()
        if full and not valid_dataset.with_phase:
            print("Warning: the validation set has no *_phase.npy files -- its loss is the L1 part only.")
    alpha_mr = ALPHA_MR if full else 0.0
    print(f"Objective: {ALPHA_L1} * L1" + (f" + {ALPHA_MR} * MR-STFT (train.py:296)" if full else " (L1 terms only)"))

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
            stream = resident.batches(per_rank_batch, True, rank, world, ep, with_phase=full)
        else:
            steps = len(train_loader)
            to_dev = lambda t: t.to(device, non_blocking=True)
            stream = ((to_dev(m), to_dev(v)) + ((to_dev(a), to_dev(b)) if full else ()) for m, v, a, b in train_loader)
        for batch in stream:
            mix, voc = batch[0], batch[1]
            mph, vph = (batch[2], batch[3]) if full else (None, None)
            l1 = model.train_step(mix, voc, loss_scale=ALPHA_L1, grad_sync=grad_sync, mix_phase=mph, voc_phase=vph,
                                  alpha_mr=alpha_mr)                                        # train.py:271-300
            loss_sum += ALPHA_L1 * l1                                                       # no host sync per step
            if model.last_mr_loss is not None:
                loss_sum += alpha_mr * model.last_mr_loss
        avg_train_loss = float(loss_sum) / max(steps, 1)
        log_buffer.append(f"{avg_train_loss}\n")

        if world > 1:                                                      # one set of running statistics for validation / checkpoints
            from .parallel import average_bn_buffers
            average_bn_buffers(model)
        if valid_loader and (ep + 1) % args.val_interval == 0:             # train.py:317-363
            model.eval()
            val_sum = 0.0
            with torch.no_grad():
                for mix, voc, mph, vph in valid_loader:
                    mix, voc = mix.to(device), voc.to(device)
                    val_sum += ALPHA_L1 * float(l1_terms(model, mix, voc))
                    if full and valid_dataset.with_phase:                  # train.py:341-346
                        val_sum += alpha_mr * float(mr_term(model, mix, voc, mph.to(device), vph.to(device)))
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
