"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU restatement (plain torch fp32/fp64, functional style) of the reference's U-Net hot
path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned: tests/test_oracle_golden.py checks every function here against the fixtures in
tests/golden/ that oracle/gen_golden.py captured by importing the reference itself
(/root/reference/model.py with an empty `auraloss` stub) in the build container.

What it follows (all citations into /root/reference):
  forward            model.py:169-201  (encoder model.py:47-76, decoder model.py:79-109)
  BatchNorm details  torch.nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1) as used at model.py:49..107
  train step         train.py:265-300, L1 terms only (train.py:274-283), nn.L1Loss semantics
                     (config.py:33,44; SURVEY.md section 0 explains why not WeightedL1Loss)
  Adam               model.py:116  (lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay)
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LEAKY_SLOPE = 0.2          # model.py:50
ENC = ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6")
DEC = ("deconv1", "deconv2", "deconv3", "deconv4", "deconv5", "deconv6")


def to_torch_state(np_state, dtype=torch.float32):
    """numpy closed-form state (synth.closed_form_state) -> OrderedDict of torch tensors."""
    out = OrderedDict()
    for k, v in np_state.items():
        t = torch.from_numpy(v.copy()) if hasattr(v, "shape") else torch.tensor(v)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def param_keys(state):
    """The 46 trainable entries, in state_dict (= model.parameters()) order."""
    return [k for k in state if not (k.endswith("running_mean") or k.endswith("running_var")
                                     or k.endswith("num_batches_tracked"))]


def _batchnorm(x, state, prefix, training, update):
    """BatchNorm2d as the reference uses it (model.py:49,54,...,105).  Training: normalise with
    the batch mean and the BIASED variance over (N,H,W); running stats move by momentum 0.1
    towards the batch mean and the UNBIASED variance; num_batches_tracked += 1."""
    g, b = state[prefix + ".weight"], state[prefix + ".bias"]
    if training:
        dims = (0, 2, 3)
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        if update:
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                unb = var * (n / max(n - 1, 1))
                state[prefix + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                state[prefix + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unb.detach())
                state[prefix + ".num_batches_tracked"] += 1
    else:
        mean, var = state[prefix + ".running_mean"], state[prefix + ".running_var"]
    inv = torch.rsqrt(var + BN_EPS)
    return (x - mean[None, :, None, None]) * (inv * g)[None, :, None, None] + b[None, :, None, None]


def _deconv_output_padding(in_size, out_size):
    """ConvTranspose2d(k5, s2, p2) with output_size= (model.py:183-198): the natural size is
    2*in-1; output_padding makes up the difference and must be 0 or 1."""
    op = []
    for i, o in zip(in_size, out_size):
        d = o - (2 * i - 1)
        if d not in (0, 1):
            raise ValueError(f"requested output size {o} unreachable from input {i}")
        op.append(d)
    return tuple(op)


def forward(state, mix, training=False, dropout_masks=None, update_stats=True, taps=None):
    """mask = UNet.forward(mix)  (model.py:169-201).

    dropout_masks: list of five (B, C) tensors with values {0, 2} applied as Dropout2d(0.5) in
    training mode (model.py:83,89,95,101,107); None disables dropout (p -> 0).
    taps: optional dict that receives every intermediate (for per-layer parity tests)."""
    x = mix
    skips = []
    for i, name in enumerate(ENC):
        r = F.conv2d(x, state[name + ".0.weight"], state[name + ".0.bias"], stride=2, padding=2)
        z = _batchnorm(r, state, name + ".1", training, update_stats)
        x = F.leaky_relu(z, LEAKY_SLOPE)
        if taps is not None:
            taps[name + ".raw"], taps[name + ".out"] = r, x
        skips.append(x)
    sizes = [mix.shape[-2:]] + [s.shape[-2:] for s in skips]   # sizes[k] = input size of encoder k+1
    d = skips[5]
    for i, name in enumerate(DEC):
        inp = d if i == 0 else torch.cat([d, skips[5 - i]], dim=1)   # prev first, skip second
        target = sizes[5 - i]
        op = _deconv_output_padding(inp.shape[-2:], target)
        r = F.conv_transpose2d(inp, state[name + ".weight"], state[name + ".bias"],
                               stride=2, padding=2, output_padding=op)
        if taps is not None:
            taps[name + ".raw"] = r
        if i == 5:
            d = r
            break
        z = _batchnorm(r, state, name + "_BAD.0", training, update_stats)
        a = F.relu(z)
        if training and dropout_masks is not None:
            a = a * dropout_masks[i][:, :, None, None].to(a.dtype)
        if taps is not None:
            taps[name + ".out"] = a
        d = a
    mask = torch.sigmoid(d)
    if taps is not None:
        taps["mask"] = mask
    return mask


def l1_mask_loss(mask, mix, voc):
    """train.py:275-283 with model.crit = nn.L1Loss():  mean|mask*mix - voc| +
    mean|(1-mask)*mix - clamp(mix-voc, 0)|."""
    pred_vocal = mask * mix
    pred_accomp = (1 - mask) * mix
    target_accomp = torch.clamp(mix - voc, min=0.0)
    return (pred_vocal - voc).abs().mean() + (pred_accomp - target_accomp).abs().mean()


def new_adam_state(state):
    return {"step": 0,
            "exp_avg": {k: torch.zeros_like(state[k]) for k in param_keys(state)},
            "exp_avg_sq": {k: torch.zeros_like(state[k]) for k in param_keys(state)}}


def adam_update(state, grads, opt, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults as constructed at model.py:116 (no amsgrad, no decay):
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""
    opt["step"] += 1
    t = opt["step"]
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    with torch.no_grad():
        for k in param_keys(state):
            g = grads[k]
            m, v = opt["exp_avg"][k], opt["exp_avg_sq"][k]
            m.mul_(beta1).add_(g, alpha=1 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
            state[k].addcdiv_(m, denom, value=-lr / bc1)


def train_step(state, opt, mix, voc, dropout_masks=None, loss_scale=1.0, lr=1e-3, apply_update=True):
    """One optimisation step (train.py:265-300, L1 part): forward in training mode, the two L1
    terms, backward, Adam.  Returns (loss, grads dict).  `loss_scale` is the reference's
    alpha_L1 (train.py:24,296): it multiplies the loss that is differentiated."""
    keys = param_keys(state)
    leaves = {}
    for k in keys:
        leaves[k] = state[k].detach().clone().requires_grad_(True)
    work = OrderedDict((k, leaves.get(k, v)) for k, v in state.items())
    mask = forward(work, mix, training=True, dropout_masks=dropout_masks, update_stats=True)
    loss = l1_mask_loss(mask, mix, voc)
    (loss * loss_scale).backward()
    grads = {k: leaves[k].grad.detach() for k in keys}
    for k, v in work.items():           # running stats were updated in `work`
        if k not in leaves:
            state[k] = v
    if apply_update:
        adam_update(state, grads, opt, lr=lr)
    return float(loss.detach()), grads
