"""`UNet` -- drop-in for the reference's `model.UNet` (/root/reference/model.py:42-220) whose
arithmetic runs entirely in hand-written gfx950 kernels behind the C ABI of include/svs_hip.h.

What is kept from the reference: the no-argument constructor, the module tree names
(so `state_dict()` has the same 79 keys and checkpoints are interchangeable), `forward(mix) -> mask`
on float32 `(B, 1, H, W)`, `.optim` (Adam lr 1e-3, model.py:116), `.crit`, `.save/.load`,
`.backward(mix, voc)`, `.getLoss()`, `.loss_list_*`.

What is different underneath:
  * the 46 parameter tensors are views into ONE flat fp32 buffer (and their gradients into another),
    so Adam is one fused kernel and the data-parallel gradient exchange is one RCCL all-reduce;
  * the nn.Conv2d / nn.BatchNorm2d modules only hold parameters -- they are never called;
  * forward() dispatches to `svs_unet_forward_eval` (BN folded into the conv epilogues) or to the
    training path (`svs_unet_train_forward` / `svs_unet_train_backward` through one autograd
    Function), and `train_step()` is the fused forward + L1 loss + backward + Adam fast path;
  * there is NO CPU / PyTorch fallback: a tensor that is not on a ROCm device raises.

`crit` is `nn.L1Loss()`: the reference's committed `WeightedL1Loss` cannot run (model.py:16-17,35 --
see SURVEY.md section 0) and train.py:281-282 calls `model.crit(pred, target)` with L1Loss's shape.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, lib, ptr


def _on_model_device(fn):
    """Runs a UNet / FusedAdam method with the model's device current, so that every stream handed to the library
    (and the library's own side stream) belongs to the device the parameters live on -- whatever device the caller
    happens to have selected."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        dev = (self._model if hasattr(self, "_model") else self)._flat.device
        if dev.type != "cuda" or torch.cuda.current_device() == dev.index:
            return fn(self, *a, **k)
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return wrapped

ENC_CHANNELS = (1, 16, 32, 64, 128, 256, 512)                                   # model.py:47-76
DEC_IO = ((512, 256), (512, 128), (256, 64), (128, 32), (64, 16), (32, 1))      # model.py:79-109
ALPHA_L1 = 166.66                                                                # train.py:24
ALPHA_MR = 0.66                                                                  # train.py:25


class WeightedL1Loss(nn.Module):
    """Frame-weighted L1 of model.py:15-40 with the missing `self.reduction` assignment restored
    (the committed class raises AttributeError at model.py:35).  Torch ops; not on the hot path."""

    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction = reduction

    @staticmethod
    def weighted(pred_spec, target_spec):
        diff = torch.abs(pred_spec - target_spec)
        return diff * diff.sum(dim=-1, keepdim=True)

    def forward(self, target_vocal, target_mix, mask):
        pred_vocal = mask * target_mix
        pred_accomp = (1 - mask) * target_mix
        target_accomp = torch.clamp(target_mix - target_vocal, min=0.0)
        loss = self.weighted(pred_vocal, target_vocal) + self.weighted(pred_accomp, target_accomp)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr=1e-3) semantics (model.py:116) as one `svs_adam_step` launch over the
    model's flat parameter / gradient / moment buffers.  state_dict() has torch.optim.Adam's
    per-parameter layout (step, exp_avg, exp_avg_sq) so reference checkpoints load and save."""

    def __init__(self, model: "UNet", lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self._model = model
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=0,
                                                        amsgrad=False, maximize=False))
        self._step = 0
        self.grad_scale = 1.0          # 1/world_size under data parallelism
        self._m = None
        self._v = None

    def _ensure_state(self):
        flat = self._model._flat
        if self._m is None or self._m.device != flat.device:
            m_old, v_old = self._m, self._v
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            if m_old is not None:
                self._m.copy_(m_old)
                self._v.copy_(v_old)
            self._publish_state()

    def _publish_state(self):
        self.state.clear()
        for p, (off, n) in zip(self._model._param_list, self._model._param_spans):
            self.state[p] = {"step": torch.tensor(float(self._step)),
                             "exp_avg": self._m[off:off + n].view_as(p),
                             "exp_avg_sq": self._v[off:off + n].view_as(p)}

    def zero_grad(self, set_to_none: bool = True):
        self._model._grads_clean = True     # the next backward overwrites the flat gradient buffer

    @torch.no_grad()
    @_on_model_device
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        model = self._model
        self._ensure_state()
        g = self.param_groups[0]
        self._step += 1
        check(lib().svs_adam_step(ptr(model._flat), ptr(model._gflat), ptr(self._m), ptr(self._v),
                                  model._flat.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                  float(g["eps"]), int(self._step), float(self.grad_scale), _lib.stream_ptr()),
              "svs_adam_step")
        model._param_epoch += 1
        return loss

    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        loaded = [self.state.get(p) for p in self._model._param_list]
        self._m = None
        self._ensure_state()
        steps = [0]
        for st, (off, n) in zip(loaded, self._model._param_spans):
            if not st:
                continue
            self._m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self._v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.append(int(float(st["step"])))
        self._step = max(steps)
        self._publish_state()


class _TrainForward(torch.autograd.Function):
    """mask = UNet(mix) in training mode.  Intermediates live in the model's workspace; backward turns
    d(mask) into parameter gradients written straight into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, anchor, mix, model):
        mask = model._train_forward(mix)
        ctx.model = model
        ctx.generation = model._generation
        ctx.save_for_backward(mix, mask)
        return mask

    @staticmethod
    def backward(ctx, d_mask):
        model = ctx.model
        if ctx.generation != model._generation:
            raise RuntimeError("UNet: backward() after a newer training forward reused the workspace; "
                               "call backward before the next forward")
        mix, mask = ctx.saved_tensors
        model._train_backward(mix, mask, d_mask.contiguous())
        return None, None, None


class UNet(nn.Module):
    def __init__(self):
        super().__init__()
        # ---- the reference's module tree (names fix the checkpoint format, model.py:47-109)
        for k in range(6):
            cin, cout = ENC_CHANNELS[k], ENC_CHANNELS[k + 1]
            setattr(self, f"conv{k + 1}", nn.Sequential(
                nn.Conv2d(cin, cout, kernel_size=(5, 5), stride=(2, 2), padding=2),
                nn.BatchNorm2d(cout),
                nn.LeakyReLU(negative_slope=0.2, inplace=True)))
        for j, (cin, cout) in enumerate(DEC_IO):
            setattr(self, f"deconv{j + 1}", nn.ConvTranspose2d(cin, cout, kernel_size=(5, 5), stride=(2, 2), padding=2))
            if j < 5:
                setattr(self, f"deconv{j + 1}_BAD", nn.Sequential(nn.BatchNorm2d(cout), nn.ReLU(True), nn.Dropout2d(0.5)))

        self.loss_list_vocal = []
        self.loss_list_accomp = []
        self.loss_list_total = []

        # ---- flat storage + runtime state (plain attributes, invisible to state_dict)
        self._param_list = list(self.parameters())
        self._param_spans = []
        off = 0
        for p in self._param_list:
            self._param_spans.append((off, p.numel()))
            off += p.numel()
        self._n_params = off
        self._bn_list = [getattr(self, f"conv{k + 1}")[1] for k in range(6)] + \
                        [getattr(self, f"deconv{j + 1}_BAD")[0] for j in range(5)]
        self._flat = self._gflat = self._bn_flat = self._nbt_flat = None
        self._param_epoch = 0
        self._prepared = None
        self._prepared_key = None
        self._ws = {}
        self._generation = 0
        self._grads_clean = True
        self._anchor = None
        self._injected_masks = None
        self.dropout_seed = 4242
        self._xstream = None            # stream the overlapped gradient exchange is issued from
        self.dropout_step = 0
        self.last_mr_loss = None        # MR-STFT part of the last training objective (device scalar) or None
        self.eval_precision = "fp32"    # "bf16": eval forwards run the bf16-MFMA network (BASELINE configs[4]); training is fp32
        self._prepared_bf16 = None
        self._prepared_bf16_key = None
        self.rank = 0
        self._flatten()

        self.optim = FusedAdam(self, lr=1e-3)
        self.crit = nn.L1Loss()

    # ==============================================================================
    #   flat storage
    # ==============================================================================
    def _flatten(self):
        """(Re)point every parameter, gradient and BN buffer at one flat buffer on the current device."""
        dev = self._param_list[0].device
        with torch.no_grad():
            flat = torch.empty(self._n_params, dtype=torch.float32, device=dev)
            gflat = torch.zeros(self._n_params, dtype=torch.float32, device=dev)
            for p, (off, n) in zip(self._param_list, self._param_spans):
                flat[off:off + n].copy_(p.data.reshape(-1).to(torch.float32))
                p.data = flat[off:off + n].view(p.shape)
                p.grad = gflat[off:off + n].view(p.shape)
            nbn = sum(2 * bn.num_features for bn in self._bn_list)
            bn_flat = torch.empty(nbn, dtype=torch.float32, device=dev)
            nbt = torch.empty(len(self._bn_list), dtype=torch.int64, device=dev)
            off = 0
            for i, bn in enumerate(self._bn_list):
                c = bn.num_features
                bn_flat[off:off + c].copy_(bn.running_mean)
                bn_flat[off + c:off + 2 * c].copy_(bn.running_var)
                nbt[i] = bn.num_batches_tracked
                bn.running_mean = bn_flat[off:off + c]
                bn.running_var = bn_flat[off + c:off + 2 * c]
                bn.num_batches_tracked = nbt[i]
                off += 2 * c
        self._flat, self._gflat, self._bn_flat, self._nbt_flat = flat, gflat, bn_flat, nbt
        self._prepared = None
        self._prepared_key = None
        self._ws = {}
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._param_epoch += 1
        if hasattr(self, "optim"):
            self.optim._ensure_state()

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._flatten()
        return out

    def _attach_grads(self):
        for p, (off, n) in zip(self._param_list, self._param_spans):
            g = p.grad
            if g is None or g.data_ptr() != self._gflat.data_ptr() + 4 * off:
                p.grad = self._gflat[off:off + n].view(p.shape)

    # ==============================================================================
    #   IO  (model.py:122-152)
    # ==============================================================================
    def load(self, path):
        if os.path.exists(path):
            print("Load the pre-trained model from {}".format(path))
            state = torch.load(path, map_location="cpu")
            for key, obj in state.items():
                if "loss_list" in key:
                    setattr(self, key, obj)
            self.load_state_dict(state["model_state_dict"], strict=False)
            if "optim" in state:
                self.optim.load_state_dict(state["optim"])
            self.dropout_step = int(state.get("dropout_step", self.optim._step))     # masks do not replay after a resume
        else:
            print("Pre-trained model {} is not exist...".format(path))

    def save(self, path):
        state = {"model_state_dict": self.state_dict(), "optim": self.optim.state_dict(), "dropout_step": self.dropout_step}
        for key in self.__dict__:
            if "loss_list" in key:
                state[key] = getattr(self, key)
        torch.save(state, path)

    # ==============================================================================
    #   Set & Get  (model.py:157-167)
    # ==============================================================================
    def getLoss(self, normalize=False):
        loss_dict = {}
        for key in self.__dict__:
            if "loss_list" in key:
                val = getattr(self, key)
                if len(val) > 0:
                    loss_dict[key] = np.mean(val) if normalize else round(val[-1], 6)
        return loss_dict

    def set_dropout_masks(self, masks):
        """Test hook: five (B, C) tensors with values {0, 2} (C = 256,128,64,32,16) used instead of the
        generated Dropout2d masks; [] disables dropout; None restores the generator."""
        self._injected_masks = masks

    # ==============================================================================
    #   device plumbing
    # ==============================================================================
    def _check_input(self, mix):
        if not isinstance(mix, torch.Tensor) or mix.device.type != "cuda":
            raise RuntimeError("svs_unet_pytorch_amd.UNet runs on a ROCm device only (hand-written gfx950 kernels, "
                               "no CPU fallback): move the input with .to('cuda')")
        if mix.device != self._flat.device:
            raise RuntimeError(f"UNet parameters are on {self._flat.device}, input on {mix.device}")
        if mix.dim() != 4 or mix.shape[1] != 1:
            raise ValueError(f"expected (B, 1, H, W), got {tuple(mix.shape)}")
        if mix.dtype != torch.float32:
            mix = mix.float()
        return mix.contiguous()

    def _workspace(self, kind, B, H, W):
        key = (kind, B, H, W)
        ws = self._ws.get(key)
        if ws is None:
            fn = {"eval": lib().svs_unet_eval_workspace_bytes, "eval_bf16": lib().svs_unet_eval_bf16_workspace_bytes,
                  "train": lib().svs_unet_train_workspace_bytes}[kind]
            nbytes = int(fn(B, H, W))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self._flat.device)
            if len(self._ws) > 8:
                self._ws.clear()
            self._ws[key] = ws
        return ws

    def _drop_masks(self, B):
        """Flat (B*256 + B*128 + B*64 + B*32 + B*16) keep-mask buffer for this step, or None."""
        if self._injected_masks is not None:
            if len(self._injected_masks) == 0:
                return None
            return torch.cat([m.reshape(-1).to(self._flat.device, torch.float32) for m in self._injected_masks])
        out = torch.empty(B * sum(c for _, c in DEC_IO[:5]), dtype=torch.float32, device=self._flat.device)
        check(lib().svs_dropout_masks_all(out.data_ptr(), B, self.dropout_seed, self.dropout_step, self.rank, _lib.stream_ptr()),
              "svs_dropout_masks_all")
        self.dropout_step += 1
        return out

    # ==============================================================================
    #   forward / backward
    # ==============================================================================
    def _eval_forward(self, mix):
        B, _, H, W = mix.shape
        # everything that can change the folded weights: in-place edits of any parameter / BatchNorm buffer (each
        # parameter has its own version counter: p.data is re-pointed at the flat buffer), and `_param_epoch`, which
        # every path that updates them through raw pointers bumps (training forwards, Adam, broadcasts, re-flattening)
        key = self._eval_key()
        if self._prepared is None or self._prepared_key != key:
            if self._prepared is None:
                self._prepared = torch.empty(int(lib().svs_unet_prepared_bytes()), dtype=torch.uint8, device=self._flat.device)
            check(lib().svs_unet_prepare_eval(ptr(self._flat), ptr(self._bn_flat), ptr(self._prepared), _lib.stream_ptr()),
                  "svs_unet_prepare_eval")
            self._prepared_key = key
        mask = torch.empty_like(mix)
        if self.eval_precision == "bf16":
            if self._prepared_bf16 is None or self._prepared_bf16_key != key:
                if self._prepared_bf16 is None or self._prepared_bf16.device != self._flat.device:
                    self._prepared_bf16 = torch.empty(int(lib().svs_unet_prepared_bf16_bytes()), dtype=torch.uint8, device=self._flat.device)
                check(lib().svs_unet_prepare_eval_bf16(ptr(self._prepared), ptr(self._prepared_bf16), _lib.stream_ptr()),
                      "svs_unet_prepare_eval_bf16")
                self._prepared_bf16_key = key
            ws = self._workspace("eval_bf16", B, H, W)
            check(lib().svs_unet_forward_eval_bf16(ptr(self._prepared_bf16), ptr(mix), ptr(mask), B, H, W, ptr(ws), ws.numel(),
                                                   _lib.stream_ptr()), "svs_unet_forward_eval_bf16")
            return mask
        if self.eval_precision != "fp32":
            raise ValueError(f"eval_precision must be 'fp32' or 'bf16', got {self.eval_precision!r}")
        ws = self._workspace("eval", B, H, W)
        check(lib().svs_unet_forward_eval(ptr(self._prepared), ptr(mix), ptr(mask), B, H, W, ptr(ws), ws.numel(),
                                          _lib.stream_ptr()), "svs_unet_forward_eval")
        return mask

    def _eval_key(self):
        """Everything that can change the folded eval weights: in-place edits of any parameter / BatchNorm buffer (each
        parameter has its own version counter: p.data is re-pointed at the flat buffer), and `_param_epoch`, which every
        path that updates them through raw pointers bumps (training forwards, Adam, broadcasts, re-flattening)."""
        return (self._flat._version, self._bn_flat._version, self._param_epoch,
                sum(p._version for p in self._param_list), sum(b._version for b in self._bn_views()))

    def _bn_views(self):
        for bn in self._bn_list:
            yield bn.running_mean
            yield bn.running_var

    def _train_forward(self, mix):
        B, _, H, W = mix.shape
        ws = self._workspace("train", B, H, W)
        self._drop = self._drop_masks(B)
        self._generation += 1
        self._param_epoch += 1          # the kernels update the BatchNorm running statistics through raw pointers
        mask = torch.empty_like(mix)
        check(lib().svs_unet_train_forward(ptr(self._flat), ptr(self._bn_flat), ptr(self._nbt_flat), ptr(mix),
                                           ptr(self._drop), B, H, W, ptr(mask), ptr(ws), ws.numel(), _lib.stream_ptr()),
              "svs_unet_train_forward")
        return mask

    def _grad_target(self):
        """Flat buffer the backward kernels write into: the real one after zero_grad, else a scratch
        that is added afterwards (torch semantics: gradients accumulate until zeroed)."""
        self._attach_grads()
        if self._grads_clean:
            return self._gflat, None
        tmp = torch.empty_like(self._gflat)
        return tmp, tmp

    @_on_model_device
    def _train_backward(self, mix, mask, d_mask):
        B, _, H, W = mix.shape
        ws = self._workspace("train", B, H, W)
        target, tmp = self._grad_target()
        check(lib().svs_unet_train_backward(ptr(self._flat), ptr(target), ptr(mix), ptr(mask), ptr(d_mask), ptr(self._drop),
                                            B, H, W, ptr(ws), ws.numel(), _lib.stream_ptr()), "svs_unet_train_backward")
        if tmp is not None:
            self._gflat.add_(tmp)
        self._grads_clean = False

    @_on_model_device
    def graphed_forward(self, static_mix):
        """Captures the eval forward on `static_mix` (its storage is the graph's input: refill it in place) into a hipGraph
        and returns `replay() -> mask`, whose result lives in a static output tensor.  Every launch of the library goes to
        the caller's stream and nothing synchronises or allocates, so the 12 launches of a forward become one graph launch
        (an eager forward at small batch is bound by host launch cost, not by the GPU)."""
        assert not self.training, "graphed_forward is an inference path: call .eval() first"
        static_mix = self._check_input(static_mix)
        with torch.no_grad():
            self._eval_forward(static_mix)                         # prepared weights + workspace exist before the capture
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=static_mix.device)
        side.wait_stream(torch.cuda.current_stream(static_mix.device))
        with torch.cuda.stream(side), torch.no_grad():
            self._eval_forward(static_mix)
            side.synchronize()
            with torch.cuda.graph(graph, stream=side):
                static_out = self._eval_forward(static_mix)
        torch.cuda.current_stream(static_mix.device).wait_stream(side)
        key = self._prepared_key

        def replay():
            if self._prepared_key != key or self._eval_key() != key:
                raise RuntimeError("UNet weights changed since the graph was captured: capture again")
            graph.replay()
            return static_out
        replay.graph, replay.output, replay.input = graph, static_out, static_mix
        return replay

    @_on_model_device
    def forward(self, mix):
        """
            Generate the mask for the given mixture audio spectrogram

            Arg:    mix     (torch.Tensor)  - The mixture spectrogram which size is (B, 1, H, W), float32, on the GPU
            Ret:    The soft mask which size is (B, 1, H, W)
        """
        mix = self._check_input(mix)
        if not self.training:
            return self._eval_forward(mix)
        if torch.is_grad_enabled():
            return _TrainForward.apply(self._anchor, mix, self)
        return self._train_forward(mix)

    def _mr_workspace(self, B, W, hop):
        key = ("mr", B, W, hop)
        ws = self._ws.get(key)
        if ws is None:
            ws = torch.empty(int(lib().svs_unet_train_mr_workspace_bytes(B, W, hop)), dtype=torch.uint8, device=self._flat.device)
            self._ws[key] = ws
        return ws

    def _fwd_losses(self, mix, voc, loss_scale, mix_phase, voc_phase, alpha_mr, ws):
        """Forward + loss(es) + d(objective)/d(logit) left in the workspace.  Returns the L1 part as a device scalar; with
        phases and alpha_mr != 0 the multi-resolution STFT term of train.py:287-296 is part of the objective and its value
        is kept in `self.last_mr_loss`."""
        B, _, H, W = mix.shape
        L = lib()
        if mix_phase is None or not alpha_mr:
            self.last_mr_loss = None
            loss = torch.empty(1, dtype=torch.float32, device=mix.device)
            check(L.svs_unet_train_fwd_loss(ptr(self._flat), ptr(self._bn_flat), ptr(self._nbt_flat), ptr(mix), ptr(voc), ptr(self._drop),
                                            B, H, W, float(loss_scale), None, ptr(loss), ptr(ws), ws.numel(), _lib.stream_ptr()),
                  "svs_unet_train_fwd_loss")
            return loss[0]
        from .config import HOP_SIZE
        mix_phase, voc_phase = self._check_input(mix_phase), self._check_input(voc_phase)
        if mix_phase.shape != mix.shape or voc_phase.shape != mix.shape:
            raise ValueError("phase tensors must have the shape of the magnitude tiles")
        mr_ws = self._mr_workspace(B, W, HOP_SIZE)
        losses = torch.empty(2, dtype=torch.float32, device=mix.device)
        check(L.svs_unet_train_fwd_loss_mr(ptr(self._flat), ptr(self._bn_flat), ptr(self._nbt_flat), ptr(mix), ptr(voc), ptr(mix_phase),
                                           ptr(voc_phase), ptr(self._drop), B, H, W, HOP_SIZE, float(loss_scale), float(alpha_mr), None,
                                           ptr(losses), ptr(ws), ws.numel(), ptr(mr_ws), mr_ws.numel(), _lib.stream_ptr()),
              "svs_unet_train_fwd_loss_mr")
        self.last_mr_loss = losses[1]
        return losses[0]

    @_on_model_device
    def fwd_bwd(self, mix, voc, loss_scale=1.0, mix_phase=None, voc_phase=None, alpha_mr=0.0):
        """Fused training forward + loss + backward.  Objective: loss_scale * L1 terms (train.py:274-283) [+ alpha_mr *
        MR-STFT of the re-synthesised waveforms (train.py:287-296) when the phase tiles are given].  Gradients land in the
        flat gradient buffer; returns the unscaled L1 part as a device scalar (no host sync); the MR part, when
        computed, is `self.last_mr_loss`."""
        mix, voc = self._check_input(mix), self._check_input(voc)
        B, _, H, W = mix.shape
        ws = self._workspace("train", B, H, W)
        self._drop = self._drop_masks(B)
        self._generation += 1
        self._param_epoch += 1
        target, tmp = self._grad_target()
        if mix_phase is None or not alpha_mr:
            self.last_mr_loss = None
            loss = torch.empty(1, dtype=torch.float32, device=mix.device)
            check(lib().svs_unet_train_fwd_bwd(ptr(self._flat), ptr(target), ptr(self._bn_flat), ptr(self._nbt_flat), ptr(mix),
                                               ptr(voc), ptr(self._drop), B, H, W, float(loss_scale), None, ptr(loss), ptr(ws),
                                               ws.numel(), _lib.stream_ptr()), "svs_unet_train_fwd_bwd")
            loss = loss[0]
        else:
            loss = self._fwd_losses(mix, voc, loss_scale, mix_phase, voc_phase, alpha_mr, ws)
            check(lib().svs_unet_train_bwd_part(ptr(self._flat), ptr(target), ptr(mix), ptr(self._drop), B, H, W, 4, ptr(ws),
                                                ws.numel(), _lib.stream_ptr()), "svs_unet_train_bwd_part")
        if tmp is not None:
            self._gflat.add_(tmp)
        self._grads_clean = False
        return loss

    @_on_model_device
    def fwd_bwd_overlapped(self, mix, voc, loss_scale, grad_sync, mix_phase=None, voc_phase=None, alpha_mr=0.0):
        """Same result as fwd_bwd, as five library calls so that the gradient exchange overlaps the backward:
        forward + loss; backward of the decoder half (its gradients occupy the tail of the flat buffer) followed
        at once by an asynchronous all-reduce of that tail; the conv6 block and its all-reduce; conv5 + conv4 and
        theirs; conv3..conv1 and theirs.  Returns (loss, [work handles])."""
        mix, voc = self._check_input(mix), self._check_input(voc)
        B, _, H, W = mix.shape
        ws = self._workspace("train", B, H, W)
        self._drop = self._drop_masks(B)
        self._generation += 1
        self._param_epoch += 1
        self._attach_grads()
        assert self._grads_clean, "overlapped exchange needs zero_grad() first (it overwrites the flat gradient buffer)"
        L = lib()
        loss = self._fwd_losses(mix, voc, loss_scale, mix_phase, voc_phase, alpha_mr, ws)
        split = int(L.svs_unet_param_offset(24))           # first decoder tensor (deconv1.weight)
        c6 = int(L.svs_unet_param_offset(20))              # conv6.weight: the conv6 block is 13 of the encoder's 17.5 MB
        c4 = int(L.svs_unet_param_offset(12))              # conv4.weight: conv5 + conv4 are 4.1 MB, conv3..conv1 0.26 MB
        handles = []
        # decoder -> conv6 block -> conv5 + conv4 -> conv3..conv1: only the last, 0.26 MB piece is exchanged after the backward
        # has ended (the backward of the three shallow encoder blocks, ~0.5 ms, hides the 4.1 MB before it)
        main = torch.cuda.current_stream(mix.device)
        if self._xstream is None:
            self._xstream = torch.cuda.Stream(device=mix.device)
        xs = self._xstream
        for part, sl in ((0, self._gflat[split:]), (2, self._gflat[c6:split]), (5, self._gflat[c4:c6]), (6, self._gflat[:c4])):
            check(L.svs_unet_train_bwd_part(ptr(self._flat), ptr(self._gflat), ptr(mix), ptr(self._drop), B, H, W, part, ptr(ws),
                                            ws.numel(), _lib.stream_ptr()), "svs_unet_train_bwd_part")
            # the exchange is issued from a stream that waits for this part on BOTH compute streams (the backward's own and
            # the library's weight-gradient stream); the backward itself goes on without waiting for either
            xs.wait_stream(main)
            check(L.svs_unet_train_bwd_sync(xs.cuda_stream), "svs_unet_train_bwd_sync")
            with torch.cuda.stream(xs):
                handles.append(grad_sync.reduce_async(sl))
        self._grads_clean = False
        return loss, handles

    def train_step(self, mix, voc, loss_scale=1.0, grad_sync=None, mix_phase=None, voc_phase=None, alpha_mr=0.0):
        """zero_grad + fwd_bwd + (optional gradient all-reduce) + Adam: the whole of train.py:271-300
        (L1 terms).  `grad_sync` is the data-parallel hook (parallel.GradAllReduce): with `reduce_async` the
        exchange of the decoder half overlaps the encoder half's backward, otherwise `grad_sync(flat_grad)`
        runs after the backward."""
        self.optim.zero_grad()
        if grad_sync is not None and getattr(grad_sync, "overlap", False):
            loss, handles = self.fwd_bwd_overlapped(mix, voc, loss_scale, grad_sync, mix_phase, voc_phase, alpha_mr)
            for h in handles:
                h.wait()
        else:
            loss = self.fwd_bwd(mix, voc, loss_scale, mix_phase, voc_phase, alpha_mr)
            if grad_sync is not None:
                grad_sync(self._gflat)
        self.optim.step()
        return loss

    def backward(self, mix, voc):
        """
            Update the parameters for the given mixture spectrogram and the pure vocal spectrogram
            (model.py:203-220 with the L1 terms of train.py:274-283; see the module docstring).
        """
        loss = self.train_step(mix, voc)
        self.loss_list_total.append(loss.item())
