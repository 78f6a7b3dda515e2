"""Per-kernel parity (-m gpu): every C-ABI block entry point against a plain PyTorch fp64 CPU
computation of the same op (torch.nn.functional on NCHW tensors), on seeded inputs.  Tolerances are
relative to the output's magnitude; fp32 MFMA is an exact fmaf chain, so the only error is
summation order (~1e-6 relative)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import stft_oracle as so
from svs_unet_pytorch_amd import _lib, synth

pytestmark = pytest.mark.gpu

DEV = "cuda"


def L():
    return _lib.lib()


def S():
    return _lib.stream_ptr()


def rnd(shape, seed, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((synth.uniform(seed, n) * (hi - lo) + lo).reshape(shape).astype(np.float32))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def relerr(got, want):
    want = want.double()
    return ((got.double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def ws_tensor(nbytes):
    return torch.empty(max(int(nbytes), 16) + 4096, dtype=torch.uint8, device=DEV)


def pack_gather(w):      # (N,C,5,5) -> device packed
    N, C = w.shape[:2]
    wd = w.to(DEV).contiguous()
    out = torch.empty(N * C * 25, device=DEV)
    _lib.check(L().svs_pack_weight_gather(wd.data_ptr(), out.data_ptr(), N, C, S()))
    return out


def pack_parity(w):      # (C,N,5,5) -> device packed
    C, N = w.shape[:2]
    wd = w.to(DEV).contiguous()
    out = torch.empty(N * C * 25, device=DEV)
    _lib.check(L().svs_pack_weight_parity(wd.data_ptr(), out.data_ptr(), C, N, S()))
    return out


def test_pack_layouts(report):
    w = rnd((32, 16, 5, 5), 1)
    got = pack_gather(w).cpu().view(32, 25, 16)
    want = w.permute(0, 2, 3, 1).reshape(32, 25, 16)
    assert torch.equal(got, want)
    wt = rnd((16, 32, 5, 5), 2)      # (C,N,5,5)
    got = pack_parity(wt).cpu()
    off = 0
    for ph in (0, 1):
        for pw in (0, 1):
            sub = wt[:, :, ph::2, pw::2]                      # (C,N,nth,ntw)
            want = sub.permute(1, 2, 3, 0).reshape(-1)        # [n][th][tw][c]
            assert torch.equal(got[off:off + want.numel()], want), (ph, pw)
            off += want.numel()
    assert off == wt.numel()


ENC_CASES = [
    # B, H, W, C, N
    (2, 64, 32, 16, 32),      # conv2-like, cfg 256x32
    (2, 32, 16, 32, 64),      # conv3-like, 64x64 / 128x64
    (3, 16, 8, 64, 128),      # conv4-like, 128x128
    (2, 8, 4, 128, 256),      # conv5-like, small M, split-K
    (1, 16, 4, 256, 512),     # conv6 at B=1 (M=16): 32x128 tile, split-K
    (1, 33, 9, 16, 32),       # odd sizes
    (4, 128, 32, 16, 16),     # N=16 tile
    (8, 64, 32, 32, 64),      # larger M for 128x64
]


@pytest.mark.parametrize("rows", ["bhw", "whb"])          # GEMM row order: planner default / batch-innermost + tap skipping forced
@pytest.mark.parametrize("B,H,W,C,N", ENC_CASES)
def test_enc_block_fwd(B, H, W, C, N, rows, report, tune):
    if rows == "whb":
        tune("CONV_SKIP", 2)
    x = rnd((B, C, H, W), 10)
    w = rnd((N, C, 5, 5), 11, -0.1, 0.1)
    b = rnd((N,), 12)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    want = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=2)
    xd = nhwc(x).to(DEV)
    wp = pack_gather(w)
    bd = b.to(DEV)
    # write into the second half of a wider buffer (the skip-concat layout)
    y = torch.full((B, Ho, Wo, 2 * N), 7.0, device=DEV)
    ws = ws_tensor(L().svs_enc_block_workspace_bytes(B, H, W, C, N))
    _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), bd.data_ptr(), None, None, 0.0,
                                     y.data_ptr() + 4 * N, 2 * N, N, 0, ws.data_ptr(), ws.numel(), S()))
    torch.cuda.synchronize()
    assert torch.all(y[..., :N] == 7.0), "wrote outside its channel slice"
    e = relerr(nchw(y[..., N:]), want)
    assert report(f"enc_fwd[{rows}] raw B{B} {H}x{W} C{C} N{N}", e, 2e-5)
    # eval epilogue: scale/shift + leaky, bias folded by the caller -> here bias=None
    sc, sh = rnd((N,), 13, 0.5, 1.5), rnd((N,), 14)
    want2 = F.leaky_relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=2) * sc.double()[None, :, None, None]
                         + sh.double()[None, :, None, None], 0.2)
    y2 = torch.empty((B, Ho, Wo, N), device=DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), None, scd.data_ptr(), shd.data_ptr(), 0.2,
                                     y2.data_ptr(), N, N, 0, ws.data_ptr(), ws.numel(), S()))
    e = relerr(nchw(y2), want2)
    assert report(f"enc_fwd[{rows}] epi B{B} {H}x{W} C{C} N{N}", e, 2e-5)
    # accumulate
    _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), None, scd.data_ptr(), shd.data_ptr(), 0.2,
                                     y2.data_ptr(), N, N, 1, ws.data_ptr(), ws.numel(), S()))
    e = relerr(nchw(y2), 2 * want2)
    assert report(f"enc_fwd acc B{B} {H}x{W} C{C} N{N}", e, 2e-5)


@pytest.mark.parametrize("B,H,W", [(2, 64, 32), (1, 33, 9), (3, 70, 50), (40, 64, 32)])
def test_conv2_gather_window(B, H, W, report, tune):
    """conv2 forward in the LDS-window form (gather_window_kernel: 16 -> 32 channels), forced on small and odd shapes; raw output with
    bias into a strided view, the BatchNorm partial rows it leaves (through svs_unet-independent ops: their sum must equal the
    channel sums of the output), and the eval epilogue."""
    C, N = 16, 32
    tune("CONV_GWINDOW", 2)
    buf = ctypes.create_string_buffer(128)
    x = rnd((B, C, H, W), 10)
    w = rnd((N, C, 5, 5), 11, -0.1, 0.1)
    b = rnd((N,), 12)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    want = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=2)
    xd = torch.full((B, H, W, C + 4), 5.0, device=DEV)
    xd[..., :C] = nhwc(x).to(DEV)
    wp = pack_gather(w)
    bd = b.to(DEV)
    y = torch.full((B, Ho, Wo, 2 * N), 7.0, device=DEV)
    ws = ws_tensor(L().svs_enc_block_workspace_bytes(B, H, W, C, N))
    _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), C + 4, B, H, W, C, wp.data_ptr(), bd.data_ptr(), None, None, 0.0,
                                     y.data_ptr() + 4 * N, 2 * N, N, 0, ws.data_ptr(), ws.numel(), S()))
    torch.cuda.synchronize()
    assert torch.all(y[..., :N] == 7.0), "wrote outside its channel slice"
    assert report(f"conv2 window raw B{B} {H}x{W}", relerr(nchw(y[..., N:]), want), 2e-5)
    sc, sh = rnd((N,), 13, 0.5, 1.5), rnd((N,), 14)
    want2 = F.leaky_relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=2) * sc.double()[None, :, None, None]
                         + sh.double()[None, :, None, None], 0.2)
    y2 = torch.empty((B, Ho, Wo, N), device=DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), C + 4, B, H, W, C, wp.data_ptr(), None, scd.data_ptr(), shd.data_ptr(), 0.2,
                                     y2.data_ptr(), N, N, 0, ws.data_ptr(), ws.numel(), S()))
    assert report(f"conv2 window epi B{B} {H}x{W}", relerr(nchw(y2), want2), 2e-5)


@pytest.mark.parametrize("kind,B,H,W,C,N", [("conv", 4, 32, 16, 128, 256), ("conv", 2, 64, 32, 32, 64), ("deconv", 4, 8, 4, 512, 128), ("wgrad", 8, 16, 8, 128, 256)])
def test_mfma_split_mode_accuracy(kind, B, H, W, C, N, report, tune):
    """The optional product mode of the GEMM kernels (csrc/mfma_split.h: fp32 operands split exactly into three bf16 limbs, six
    limb products on the bf16 MFMA, fp32 accumulation) against float64, beside the default fp32-MFMA kernels on the same
    operands -- operands with four decades of dynamic range, error relative to sum |a b| (the scale of a dot product's
    rounding error).  The mode must be at least as accurate as the fp32 MFMA (measured: slightly better)."""
    g = torch.Generator().manual_seed(7)
    spread = lambda shape: (torch.rand(shape, generator=g) - 0.5) * torch.exp(4.0 * (torch.rand(shape, generator=g) - 0.5))
    errs = {}
    if kind == "conv":
        x, w = spread((B, C, H, W)), spread((N, C, 5, 5)) * 0.1
        want = F.conv2d(x.double(), w.double(), None, stride=2, padding=2)
        mag = F.conv2d(x.double().abs(), w.double().abs(), None, stride=2, padding=2)
        xd, wp = nhwc(x).to(DEV), pack_gather(w)
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        ws = ws_tensor(L().svs_enc_block_workspace_bytes(B, H, W, C, N))
        for mode in (-1, 1):
            tune("MFMA_SPLIT", mode)
            y = torch.empty((B, Ho, Wo, N), device=DEV)
            _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), N, N, 0,
                                             ws.data_ptr(), ws.numel(), S()))
            errs[mode] = ((nchw(y).cpu().double() - want).abs() / mag)
    elif kind == "deconv":
        x, w = spread((B, C, H, W)), spread((C, N, 5, 5)) * 0.1
        want = F.conv_transpose2d(x.double(), w.double(), None, stride=2, padding=2, output_padding=1)
        mag = F.conv_transpose2d(x.double().abs(), w.double().abs(), None, stride=2, padding=2, output_padding=1)
        xd, wp = nhwc(x).to(DEV), pack_parity(w)
        ws = ws_tensor(L().svs_dec_block_workspace_bytes(B, H, W, C, 2 * H, 2 * W, N))
        for mode in (-1, 1):
            tune("MFMA_SPLIT", mode)
            y = torch.empty((B, 2 * H, 2 * W, N), device=DEV)
            _lib.check(L().svs_dec_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), None, None, None, 0.0, y.data_ptr(), N, 2 * H, 2 * W, N, 0,
                                             ws.data_ptr(), ws.numel(), S()))
            errs[mode] = ((nchw(y).cpu().double() - want).abs() / mag)
    else:                                                        # weight gradient of an encoder block: S = output side (N), L = input side (C)
        dy, x = spread((B, N, H, W)), spread((B, C, 2 * H, 2 * W))
        xq = x.double().requires_grad_(False)
        wz = torch.zeros((N, C, 5, 5), dtype=torch.float64, requires_grad=True)
        F.conv2d(xq, wz, None, stride=2, padding=2).backward(dy.double())
        want = wz.grad
        wa = torch.zeros((N, C, 5, 5), dtype=torch.float64, requires_grad=True)
        F.conv2d(xq.abs(), wa, None, stride=2, padding=2).backward(dy.double().abs())
        mag = wa.grad
        sd, ld = nhwc(dy).to(DEV), nhwc(x).to(DEV)
        ws = ws_tensor(L().svs_block_bwd_weight_workspace_bytes(B, H, W, N, C))
        for mode in (-1, 1):
            tune("MFMA_SPLIT", mode)
            dw = torch.empty(N * C * 25, device=DEV)
            _lib.check(L().svs_enc_block_bwd_weight(sd.data_ptr(), N, B, H, W, N, ld.data_ptr(), C, 2 * H, 2 * W, C, dw.data_ptr(), None,
                                                    ws.data_ptr(), ws.numel(), S()))
            errs[mode] = ((dw.view(N, C, 5, 5).cpu().double() - want).abs() / mag)
    e32, esp = errs[-1], errs[1]
    assert report(f"fp32 MFMA      {kind} B{B} {H}x{W} C{C} N{N}: max err / sum|ab|", e32.max().item(), 6e-7)
    assert report(f"split-bf16 mode {kind} B{B} {H}x{W} C{C} N{N}: max err / sum|ab|", esp.max().item(), 6e-7)
    assert report(f"split-bf16 mean error relative to the fp32 MFMA's ({kind} C{C} N{N})", (esp.mean() / e32.mean()).item(), 1.1)


@pytest.mark.parametrize("form", [0, 2], ids=["thread-per-pixel", "tiled"])
def test_enc_block_fwd_c1(form, report, tune):
    tune("CONV_C1_TILED", form)          # both forms of the single-channel convolution at every size (the planner picks per channel count)
    for (B, H, W, N) in ((2, 64, 32, 16), (1, 33, 17, 16), (2, 32, 32, 32), (3, 21, 9, 32), (1, 4, 2, 16), (1, 300, 140, 32)):
        x = rnd((B, 1, H, W), 20, 0, 1)
        w = rnd((N, 1, 5, 5), 21, -0.2, 0.2)
        b = rnd((N,), 22)
        want = F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=2)
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        y = torch.full((B, Ho, Wo, N + 8), 7.0, device=DEV)       # strided output view: the pad columns stay untouched
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), 1, B, H, W, 1, wd.data_ptr(), bd.data_ptr(), None, None, 0.0,
                                         y.data_ptr(), N + 8, N, 0, None, 0, S()))
        assert torch.all(y[..., N:] == 7.0)
        assert report(f"enc_fwd_c1 B{B} {H}x{W} N{N}", relerr(nchw(y[..., :N]), want), 1e-5)
        # folded-BatchNorm epilogue + LeakyReLU, then the same call accumulating onto its own result
        sc, sh = rnd((N,), 23, 0.5, 1.5), rnd((N,), 24)
        want2 = F.leaky_relu(F.conv2d(x.double(), w.double(), None, stride=2, padding=2) * sc.double()[None, :, None, None]
                             + sh.double()[None, :, None, None], 0.2)
        y2 = torch.empty((B, Ho, Wo, N), device=DEV)
        scd, shd = sc.to(DEV), sh.to(DEV)
        for acc in (0, 1):
            _lib.check(L().svs_enc_block_fwd(xd.data_ptr(), 1, B, H, W, 1, wd.data_ptr(), None, scd.data_ptr(), shd.data_ptr(), 0.2,
                                             y2.data_ptr(), N, N, acc, None, 0, S()))
            assert report(f"enc_fwd_c1 epi acc={acc} B{B} {H}x{W} N{N}", relerr(nchw(y2), (1 + acc) * want2), 1e-5)


DEC_CASES = [
    # B, H, W, C, N, Ho, Wo
    (2, 8, 2, 512, 256, 16, 4),     # deconv1
    (2, 16, 4, 512, 128, 32, 8),    # deconv2
    (2, 16, 8, 256, 64, 32, 16),    # deconv3-like
    (2, 32, 16, 128, 32, 64, 32),   # deconv4-like
    (2, 64, 16, 64, 16, 128, 32),   # deconv5-like (N=16)
    (1, 9, 2, 64, 64, 17, 4),       # odd height (513-row path)
    (1, 5, 7, 32, 32, 9, 13),       # odd both
]


@pytest.mark.parametrize("rows", ["bhw", "whb"])
@pytest.mark.parametrize("B,H,W,C,N,Ho,Wo", DEC_CASES)
def test_dec_block_fwd(B, H, W, C, N, Ho, Wo, rows, report, tune):
    if rows == "whb":
        tune("CONV_SKIP", 2)
    x = rnd((B, C, H, W), 30)
    w = rnd((C, N, 5, 5), 31, -0.1, 0.1)
    b = rnd((N,), 32)
    op = (Ho - (2 * H - 1), Wo - (2 * W - 1))
    want = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=2, output_padding=op)
    xd = nhwc(x).to(DEV)
    wp = pack_parity(w)
    bd = b.to(DEV)
    y = torch.full((B, Ho, Wo, 2 * N), -3.0, device=DEV)
    ws = ws_tensor(L().svs_dec_block_workspace_bytes(B, H, W, C, Ho, Wo, N))
    _lib.check(L().svs_dec_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), bd.data_ptr(), None, None, 0.0,
                                     y.data_ptr(), 2 * N, Ho, Wo, N, 0, ws.data_ptr(), ws.numel(), S()))
    torch.cuda.synchronize()
    assert torch.all(y[..., N:] == -3.0)
    assert report(f"dec_fwd[{rows}] raw B{B} {H}x{W}->{Ho}x{Wo} C{C} N{N}", relerr(nchw(y[..., :N]), want), 2e-5)
    sc, sh = rnd((N,), 33, 0.5, 1.5), rnd((N,), 34)
    want2 = F.relu(F.conv_transpose2d(x.double(), w.double(), None, stride=2, padding=2, output_padding=op)
                   * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    y2 = torch.empty((B, Ho, Wo, N), device=DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    _lib.check(L().svs_dec_block_fwd(xd.data_ptr(), C, B, H, W, C, wp.data_ptr(), None, scd.data_ptr(), shd.data_ptr(), 0.0,
                                     y2.data_ptr(), N, Ho, Wo, N, 0, ws.data_ptr(), ws.numel(), S()))
    assert report(f"dec_fwd[{rows}] epi B{B} {H}x{W}->{Ho}x{Wo} C{C} N{N}", relerr(nchw(y2), want2), 2e-5)


WINDOW_CASES = [
    # B, H, W, C, N, Ho, Wo  (the LDS-window parity kernel, forced on shapes that would not fill the GPU)
    (2, 64, 16, 64, 16, 128, 32),
    (1, 9, 18, 64, 16, 17, 35),         # ragged tiles, odd output
    (2, 16, 32, 32, 16, 32, 64),
    (1, 13, 17, 32, 16, 26, 33),
    (1, 3, 5, 64, 16, 6, 10),           # smaller than one tile
    (2, 16, 16, 128, 32, 32, 32),       # two staging phases, two column tiles (deconv4 forward)
    (1, 9, 18, 128, 16, 18, 35),
    (2, 16, 16, 64, 32, 31, 32),        # conv3 backward-data
    (1, 11, 20, 32, 32, 21, 40),
]


# 32 output channels: once with the two 16-channel halves in separate blocks (what grids below 256 tiles get), once whole
@pytest.mark.parametrize("B,H,W,C,N,Ho,Wo,force", [c + (2,) for c in WINDOW_CASES] + [c + (3,) for c in WINDOW_CASES if c[4] == 32])
def test_parity_window_kernel(B, H, W, C, N, Ho, Wo, force, report, tune):
    tune("CONV_WINDOW", force)
    buf = ctypes.create_string_buffer(128)
    L().svs_describe_plan(1, B, H, W, C, Ho, Wo, N, buf, 128)
    assert buf.value.decode().startswith("parity_window_kernel"), buf.value
    x = rnd((B, C, H, W), 35)
    w = rnd((C, N, 5, 5), 36, -0.1, 0.1)
    b = rnd((N,), 37)
    op = (Ho - (2 * H - 1), Wo - (2 * W - 1))
    want = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=2, output_padding=op)
    xd = torch.full((B, H, W, C + 8), 7.0, device=DEV)          # strided input view: the pad columns must never be read
    xd[..., :C] = nhwc(x).to(DEV)
    wp, bd = pack_parity(w), b.to(DEV)
    y = torch.full((B, Ho, Wo, 2 * N), -3.0, device=DEV)
    ws = ws_tensor(64)
    _lib.check(L().svs_dec_block_fwd(xd.data_ptr(), C + 8, B, H, W, C, wp.data_ptr(), bd.data_ptr(), None, None, 0.0,
                                     y.data_ptr(), 2 * N, Ho, Wo, N, 0, ws.data_ptr(), ws.numel(), S()))
    torch.cuda.synchronize()
    assert torch.all(y[..., N:] == -3.0)
    assert report(f"window dec_fwd B{B} {H}x{W}->{Ho}x{Wo} C{C} N{N}", relerr(nchw(y[..., :N]), want), 2e-5)
    # accumulate + folded-BN epilogue, and bit-identical to the direct kernel's arithmetic order is NOT required: compare to fp64
    sc, sh = rnd((N,), 38, 0.5, 1.5), rnd((N,), 39)
    base = rnd((B, Ho, Wo, N), 44)
    want2 = F.relu(F.conv_transpose2d(x.double(), w.double(), None, stride=2, padding=2, output_padding=op)
                   * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]) + nchw(base).double()
    y2 = base.clone().to(DEV)
    scd, shd = sc.to(DEV), sh.to(DEV)
    _lib.check(L().svs_dec_block_fwd(xd.data_ptr(), C + 8, B, H, W, C, wp.data_ptr(), None, scd.data_ptr(), shd.data_ptr(), 0.0,
                                     y2.data_ptr(), N, Ho, Wo, N, 1, ws.data_ptr(), ws.numel(), S()))
    assert report(f"window dec_fwd epi+acc B{B} {H}x{W}->{Ho}x{Wo} C{C} N{N}", relerr(nchw(y2), want2), 2e-5)


def test_out_block_fwd(report):
    for (B, H, W, Ho, Wo) in ((2, 32, 16, 64, 32), (1, 17, 8, 33, 16), (1, 9, 5, 17, 9)):
        C = 32
        x = rnd((B, C, H, W), 40)
        w = rnd((C, 1, 5, 5), 41, -0.2, 0.2)
        b = rnd((1,), 42)
        op = (Ho - (2 * H - 1), Wo - (2 * W - 1))
        want = torch.sigmoid(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=2, output_padding=op))
        xd, wd, bd = nhwc(x).to(DEV), w.to(DEV), b.to(DEV)
        y = torch.empty((B, 1, Ho, Wo), device=DEV)
        _lib.check(L().svs_out_block_fwd(xd.data_ptr(), C, B, H, W, C, wd.data_ptr(), bd.data_ptr(), y.data_ptr(), Ho, Wo, 1, S()))
        assert report(f"out_block B{B} {H}x{W}->{Ho}x{Wo}", relerr(y, want), 1e-5)


@pytest.mark.parametrize("B,H,W,C,N", [(2, 32, 16, 32, 64), (2, 16, 8, 128, 256), (1, 33, 9, 16, 32), (4, 64, 32, 16, 32)])
def test_enc_block_bwd(B, H, W, C, N, report):
    x = rnd((B, C, H, W), 50).double().requires_grad_(True)
    w = rnd((N, C, 5, 5), 51, -0.1, 0.1).double().requires_grad_(True)
    b = rnd((N,), 52).double().requires_grad_(True)
    y = F.conv2d(x, w, b, stride=2, padding=2)
    dy = rnd(tuple(y.shape), 53)
    y.backward(dy.double())
    Ho, Wo = y.shape[-2:]
    dyd = nhwc(dy).to(DEV)
    xd = nhwc(x.detach().float()).to(DEV)
    wpar = pack_parity(w.detach().float())          # conv weight (N,C,..) read as (in=N, out=C)
    dx = torch.zeros((B, H, W, C), device=DEV)
    ws = ws_tensor(max(L().svs_dec_block_workspace_bytes(B, Ho, Wo, N, H, W, C),
                       L().svs_block_bwd_weight_workspace_bytes(B, Ho, Wo, N, C)))
    _lib.check(L().svs_enc_block_bwd_data(dyd.data_ptr(), N, B, Ho, Wo, N, wpar.data_ptr(), dx.data_ptr(), C, H, W, C, 0,
                                          ws.data_ptr(), ws.numel(), S()))
    assert report(f"enc_bwd_data B{B} {H}x{W} C{C} N{N}", relerr(nchw(dx), x.grad), 2e-5)
    dw = torch.empty((N, C, 5, 5), device=DEV)
    db = torch.empty((N,), device=DEV)
    _lib.check(L().svs_enc_block_bwd_weight(dyd.data_ptr(), N, B, Ho, Wo, N, xd.data_ptr(), C, H, W, C, dw.data_ptr(), db.data_ptr(),
                                            ws.data_ptr(), ws.numel(), S()))
    assert report(f"enc_bwd_weight B{B} {H}x{W} C{C} N{N}", relerr(dw, w.grad), 2e-5)
    assert report(f"enc_bwd_bias B{B} {H}x{W} C{C} N{N}", relerr(db, b.grad), 2e-5)


@pytest.mark.parametrize("B,H,W,C,N,Ho,Wo", [(2, 16, 4, 512, 128, 32, 8), (2, 32, 16, 128, 32, 64, 32), (2, 64, 16, 64, 16, 128, 32),
                                             (1, 9, 2, 64, 64, 17, 4)])
def test_dec_block_bwd(B, H, W, C, N, Ho, Wo, report):
    x = rnd((B, C, H, W), 60).double().requires_grad_(True)
    w = rnd((C, N, 5, 5), 61, -0.1, 0.1).double().requires_grad_(True)
    b = rnd((N,), 62).double().requires_grad_(True)
    op = (Ho - (2 * H - 1), Wo - (2 * W - 1))
    y = F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=op)
    dy = rnd(tuple(y.shape), 63)
    y.backward(dy.double())
    dyd = nhwc(dy).to(DEV)
    xd = nhwc(x.detach().float()).to(DEV)
    wgat = pack_gather(w.detach().float())          # convT weight (C,N,..) read as (n=C, c=N)
    dx = torch.zeros((B, H, W, C), device=DEV)
    ws = ws_tensor(max(L().svs_enc_block_workspace_bytes(B, Ho, Wo, N, C),
                       L().svs_block_bwd_weight_workspace_bytes(B, H, W, C, N)))
    _lib.check(L().svs_dec_block_bwd_data(dyd.data_ptr(), N, B, Ho, Wo, N, wgat.data_ptr(), dx.data_ptr(), C, H, W, C, 0,
                                          ws.data_ptr(), ws.numel(), S()))
    assert report(f"dec_bwd_data B{B} {H}x{W} C{C} N{N}", relerr(nchw(dx), x.grad), 2e-5)
    dw = torch.empty((C, N, 5, 5), device=DEV)
    db = torch.empty((N,), device=DEV)
    _lib.check(L().svs_dec_block_bwd_weight(xd.data_ptr(), C, B, H, W, C, dyd.data_ptr(), N, Ho, Wo, N, dw.data_ptr(), db.data_ptr(),
                                            ws.data_ptr(), ws.numel(), S()))
    assert report(f"dec_bwd_weight B{B} {H}x{W} C{C} N{N}", relerr(dw, w.grad), 2e-5)
    assert report(f"dec_bwd_bias B{B} {H}x{W} C{C} N{N}", relerr(db, b.grad), 2e-5)


@pytest.mark.parametrize("kind,B,H,W,C,N", [("enc", 16, 8, 4, 64, 128), ("enc", 32, 16, 16, 32, 128), ("dec", 16, 4, 2, 256, 128),
                                            ("dec", 16, 8, 8, 128, 64)])
def test_wgrad_padding_skip(kind, B, H, W, C, N, report, tune):
    """Weight gradients with batch-innermost pixels and padding-only K-tiles skipped (forced on small shapes), against
    torch fp64; strided operand views."""
    tune("WGRAD_SKIP", 2)
    _wgrad_case(kind, B, H, W, C, N, report, "wgrad skip")


@pytest.mark.parametrize("kind,B,H,W,C,N", [("enc", 2, 16, 32, 16, 32), ("enc", 1, 18, 40, 32, 64), ("enc", 3, 8, 64, 16, 64),
                                            ("dec", 2, 8, 16, 64, 16), ("dec", 1, 6, 20, 128, 32), ("dec", 2, 4, 16, 32, 16),
                                            ("enc", 1, 34, 34, 32, 128)])
def test_wgrad_window_kernel(kind, B, H, W, C, N, report, tune):
    """The LDS-window weight-gradient kernel (shallow layers), forced on small and ragged shapes."""
    tune("WGRAD_WINDOW", 2)
    buf = ctypes.create_string_buffer(128)
    if kind == "enc":
        L().svs_describe_plan(2, B, (H + 1) // 2, (W + 1) // 2, N, 0, 0, C, buf, 128)
    else:
        L().svs_describe_plan(2, B, H, W, C, 0, 0, N, buf, 128)
    assert buf.value.decode().startswith("wgrad_window_kernel"), buf.value
    _wgrad_case(kind, B, H, W, C, N, report, "wgrad window")


def _wgrad_case(kind, B, H, W, C, N, report, tag):
    if kind == "enc":
        x = rnd((B, C, H, W), 80).double()
        w = rnd((N, C, 5, 5), 81, -0.1, 0.1).double().requires_grad_(True)
        y = F.conv2d(x, w, None, stride=2, padding=2)
    else:
        x = rnd((B, C, H, W), 80).double()
        w = rnd((C, N, 5, 5), 81, -0.1, 0.1).double().requires_grad_(True)
        y = F.conv_transpose2d(x, w, None, stride=2, padding=2, output_padding=1)
    dy = rnd(tuple(y.shape), 83)
    y.backward(dy.double())
    Ho, Wo = y.shape[2], y.shape[3]
    dyd = torch.full((B, Ho, Wo, N + 4), 9.0, device=DEV)
    dyd[..., :N] = nhwc(dy).to(DEV)
    xd = torch.full((B, H, W, C + 8), 9.0, device=DEV)
    xd[..., :C] = nhwc(x.float()).to(DEV)
    dw = torch.empty(tuple(w.shape), device=DEV)
    if kind == "enc":
        ws = ws_tensor(L().svs_block_bwd_weight_workspace_bytes(B, Ho, Wo, N, C))
        _lib.check(L().svs_enc_block_bwd_weight(dyd.data_ptr(), N + 4, B, Ho, Wo, N, xd.data_ptr(), C + 8, H, W, C, dw.data_ptr(), None,
                                                ws.data_ptr(), ws.numel(), S()))
    else:
        ws = ws_tensor(L().svs_block_bwd_weight_workspace_bytes(B, H, W, C, N))
        _lib.check(L().svs_dec_block_bwd_weight(xd.data_ptr(), C + 8, B, H, W, C, dyd.data_ptr(), N + 4, Ho, Wo, N, dw.data_ptr(), None,
                                                ws.data_ptr(), ws.numel(), S()))
    assert report(f"{tag} {kind} B{B} {H}x{W} C{C} N{N}", relerr(dw, w.grad), 2e-5)


def test_single_channel_bwd(report):
    # conv1: dw, db ; deconv6: dw, db, dx
    B, H, W = 2, 64, 32
    x = rnd((B, 1, H, W), 70, 0, 1).double()
    w = rnd((16, 1, 5, 5), 71, -0.2, 0.2).double().requires_grad_(True)
    b = rnd((16,), 72).double().requires_grad_(True)
    y = F.conv2d(x, w, b, stride=2, padding=2)
    dy = rnd(tuple(y.shape), 73)
    y.backward(dy.double())
    dyd, xd = nhwc(dy).to(DEV), x.float().to(DEV)
    dw, db = torch.empty((16, 1, 5, 5), device=DEV), torch.empty(16, device=DEV)
    ws = ws_tensor(L().svs_block_bwd_weight_workspace_bytes(B, H // 2, W // 2, 16, 1))
    _lib.check(L().svs_enc_block_bwd_weight(dyd.data_ptr(), 16, B, H // 2, W // 2, 16, xd.data_ptr(), 1, H, W, 1, dw.data_ptr(),
                                            db.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    assert report("conv1_bwd_weight", relerr(dw, w.grad), 2e-5)
    assert report("conv1_bwd_bias", relerr(db, b.grad), 2e-5)

    C = 32
    x = rnd((B, C, H // 2, W // 2), 74).double().requires_grad_(True)
    w = rnd((C, 1, 5, 5), 75, -0.2, 0.2).double().requires_grad_(True)
    b = rnd((1,), 76).double().requires_grad_(True)
    y = F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)
    dy = rnd(tuple(y.shape), 77)
    y.backward(dy.double())
    xd, dyd, wd = nhwc(x.detach().float()).to(DEV), dy.to(DEV), w.detach().float().to(DEV)
    dw, db = torch.empty((C, 1, 5, 5), device=DEV), torch.empty(1, device=DEV)
    dx = torch.empty((B, H // 2, W // 2, C), device=DEV)
    ws = ws_tensor(L().svs_block_bwd_weight_workspace_bytes(B, H // 2, W // 2, C, 1))
    _lib.check(L().svs_dec_block_bwd_weight(xd.data_ptr(), C, B, H // 2, W // 2, C, dyd.data_ptr(), 1, H, W, 1, dw.data_ptr(),
                                            db.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    _lib.check(L().svs_dec_block_bwd_data(dyd.data_ptr(), 1, B, H, W, 1, wd.data_ptr(), dx.data_ptr(), C, H // 2, W // 2, C, 0,
                                          ws.data_ptr(), ws.numel(), S()))
    assert report("deconv6_bwd_weight", relerr(dw, w.grad), 2e-5)
    assert report("deconv6_bwd_bias", relerr(db, b.grad), 2e-5)
    assert report("deconv6_bwd_data", relerr(nchw(dx), x.grad), 2e-5)


@pytest.mark.parametrize("B,H,W", [(4, 128, 128), (4, 130, 126), (5, 96, 150)])
def test_single_channel_wgrad_mfma(B, H, W, report):
    """The MFMA form of the single-channel weight gradients (>= 256 tiles of 4 x 16 pixels), conv1 (16 channels) and deconv6
    (32 channels), ragged tiles and odd large-grid sizes included; strided S view."""
    Hs, Ws = (H + 1) // 2, (W + 1) // 2
    x = rnd((B, 1, H, W), 90, 0, 1).double()
    w = rnd((16, 1, 5, 5), 91, -0.2, 0.2).double().requires_grad_(True)
    y = F.conv2d(x, w, None, stride=2, padding=2)
    dy = rnd(tuple(y.shape), 92)
    y.backward(dy.double())
    dyd = torch.full((B, Hs, Ws, 20), 9.0, device=DEV)
    dyd[..., :16] = nhwc(dy).to(DEV)
    xd = x.float().to(DEV)
    dw = torch.empty((16, 1, 5, 5), device=DEV)
    ws = ws_tensor(L().svs_block_bwd_weight_workspace_bytes(B, Hs, Ws, 16, 1))
    _lib.check(L().svs_enc_block_bwd_weight(dyd.data_ptr(), 20, B, Hs, Ws, 16, xd.data_ptr(), 1, H, W, 1, dw.data_ptr(), None,
                                            ws.data_ptr(), ws.numel(), S()))
    assert report(f"conv1 wgrad (MFMA) B{B} {H}x{W}", relerr(dw, w.grad), 2e-5)
    C = 32
    x = rnd((B, C, Hs, Ws), 93).double()
    w = rnd((C, 1, 5, 5), 94, -0.2, 0.2).double().requires_grad_(True)
    op = (H - (2 * Hs - 1), W - (2 * Ws - 1))
    y = F.conv_transpose2d(x, w, None, stride=2, padding=2, output_padding=op)
    dy = rnd(tuple(y.shape), 95)
    y.backward(dy.double())
    xd = torch.full((B, Hs, Ws, C + 4), 9.0, device=DEV)
    xd[..., :C] = nhwc(x.float()).to(DEV)
    dyd = dy.to(DEV)
    dw = torch.empty((C, 1, 5, 5), device=DEV)
    ws = ws_tensor(L().svs_block_bwd_weight_workspace_bytes(B, Hs, Ws, C, 1))
    _lib.check(L().svs_dec_block_bwd_weight(xd.data_ptr(), C + 4, B, Hs, Ws, C, dyd.data_ptr(), 1, H, W, 1, dw.data_ptr(), None,
                                            ws.data_ptr(), ws.numel(), S()))
    assert report(f"deconv6 wgrad (MFMA) B{B} {H}x{W}", relerr(dw, w.grad), 2e-5)


@pytest.mark.parametrize("inline", [-1, 0], ids=["inline-finalise", "finalise-launch"])     # svs_bn_bwd: the apply kernel folds the partial rows itself / bn_bwd_finalize launch
@pytest.mark.parametrize("B,H,W,C,slope,use_drop", [(4, 16, 8, 64, 0.2, False), (3, 8, 4, 256, 0.0, True), (2, 64, 32, 16, 0.2, False),
                                                     (2, 4, 2, 512, 0.0, True), (16, 64, 64, 16, 0.2, False)])     # (the last: 256 partial rows, above the inline limit)
def test_bn_train_fwd_bwd(B, H, W, C, slope, use_drop, inline, report):
    _lib.tuning("BN_INLINE", inline)
    try:
        _bn_train_fwd_bwd(B, H, W, C, slope, use_drop, report)
    finally:
        _lib.tuning("BN_INLINE", -1)


def _bn_train_fwd_bwd(B, H, W, C, slope, use_drop, report):
    x = (rnd((B, C, H, W), 80) * 2 + 0.3).double().requires_grad_(True)
    gamma = rnd((C,), 81, 0.5, 1.5).double().requires_grad_(True)
    beta = rnd((C,), 82, -0.2, 0.2).double().requires_grad_(True)
    rm0, rv0 = rnd((C,), 83, -0.1, 0.1), rnd((C,), 84, 0.5, 1.5)
    rm, rv = rm0.clone().double(), rv0.clone().double()
    drop = None
    if use_drop:
        bits = (synth.u32(5, np.arange(B * C, dtype=np.uint64)) >> np.uint32(31)).astype(np.float32) * 2
        drop = torch.from_numpy(bits.reshape(B, C))
    z = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    a = F.leaky_relu(z, slope)
    if drop is not None:
        a = a * drop.double()[:, :, None, None]
    dy = rnd(tuple(a.shape), 85)
    a.backward(dy.double())

    P = B * H * W
    xd = nhwc(x.detach().float()).to(DEV)
    gd, bd = gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
    rmd, rvd = rm0.to(DEV), rv0.to(DEV)
    nbt = torch.tensor([3], dtype=torch.int64, device=DEV)
    mean, invstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws = ws_tensor(L().svs_bn_workspace_bytes(P, C))
    y = torch.empty((B, H, W, 2 * C), device=DEV)
    dd = drop.to(DEV).contiguous() if drop is not None else None
    _lib.check(L().svs_bn_stats(xd.data_ptr(), C, P, C, ws.data_ptr(), ws.numel(), S()))
    _lib.check(L().svs_bn_finalize(ws.data_ptr(), P, C, 1e-5, 0.1, rmd.data_ptr(), rvd.data_ptr(), nbt.data_ptr(), mean.data_ptr(),
                                   invstd.data_ptr(), S()))
    _lib.check(L().svs_bn_act_apply(xd.data_ptr(), C, P, C, H * W, gd.data_ptr(), bd.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                    slope, _lib.ptr(dd), y.data_ptr() + 4 * C, 2 * C, S()))
    tag = f"bn B{B} {H}x{W} C{C}"
    assert report(tag + " fwd", relerr(nchw(y[..., C:]), a.detach()), 1e-5)
    assert report(tag + " running_mean", relerr(rmd, rm), 1e-5)
    assert report(tag + " running_var", relerr(rvd, rv), 1e-5)
    assert int(nbt.item()) == 4
    dyw = torch.zeros((B, H, W, 2 * C), device=DEV)
    dyw[..., C:] = nhwc(dy).to(DEV)
    d_raw = torch.empty((B, H, W, C), device=DEV)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    _lib.check(L().svs_bn_bwd(dyw.data_ptr() + 4 * C, 2 * C, xd.data_ptr(), C, P, C, H * W, gd.data_ptr(), bd.data_ptr(), mean.data_ptr(),
                              invstd.data_ptr(), slope, _lib.ptr(dd), d_raw.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                              ws.numel(), S()))
    assert report(tag + " bwd dx", relerr(nchw(d_raw), x.grad), 2e-5)
    assert report(tag + " bwd dgamma", relerr(dg, gamma.grad), 2e-5)
    assert report(tag + " bwd dbeta", relerr(db, beta.grad), 2e-5)


def test_bn_fold_and_loss_and_adam(report):
    C = 64
    g, b, rm, rv, cb = rnd((C,), 90, 0.5, 1.5), rnd((C,), 91), rnd((C,), 92), rnd((C,), 93, 0.5, 1.5), rnd((C,), 94)
    sc, sh = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    d = [t.to(DEV) for t in (g, b, rm, rv, cb)]
    _lib.check(L().svs_bn_fold(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), 1e-5,
                               sc.data_ptr(), sh.data_ptr(), C, S()))
    s_want = g.double() / torch.sqrt(rv.double() + 1e-5)
    assert report("bn_fold scale", relerr(sc, s_want), 1e-6)
    assert report("bn_fold shift", relerr(sh, b.double() + (cb.double() - rm.double()) * s_want), 1e-6)

    # loss: train.py:274-283 with nn.L1Loss
    n = 4 * 512 * 128
    mix_np, voc_np = synth.tiles(4)
    mix, voc = torch.from_numpy(mix_np).double(), torch.from_numpy(voc_np).double()
    logit = (rnd((4, 1, 512, 128), 95) * 3).double().requires_grad_(True)
    mask = torch.sigmoid(logit)
    loss = (mask * mix - voc).abs().mean() + ((1 - mask) * mix - torch.clamp(mix - voc, min=0)).abs().mean()
    (loss * 166.66).backward()
    md = mask.detach().float().to(DEV)
    dl = torch.empty(n, device=DEV)
    lo = torch.empty(1, device=DEV)
    ws = ws_tensor(L().svs_l1_mask_loss_workspace_bytes(n))
    mixd, vocd = torch.from_numpy(mix_np).to(DEV), torch.from_numpy(voc_np).to(DEV)
    _lib.check(L().svs_l1_mask_loss_fwd_bwd(md.data_ptr(), mixd.data_ptr(), vocd.data_ptr(), n, 166.66, dl.data_ptr(), lo.data_ptr(),
                                            ws.data_ptr(), ws.numel(), S()))
    assert report("l1_loss value", abs(lo.item() - loss.item()) / loss.item(), 1e-6)
    assert report("l1_loss d_logit", relerr(dl.view_as(logit), logit.grad), 1e-4)

    # Adam: three steps against torch.optim.Adam
    n = 10007
    p0, gs = rnd((n,), 96), [rnd((n,), 97 + i, -0.01, 0.01) for i in range(3)]
    pt = p0.clone().double().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3)
    pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, g in enumerate(gs):
        pt.grad = g.double()
        opt.step()
        gd = g.to(DEV)
        _lib.check(L().svs_adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999, 1e-8, i + 1, 1.0, S()))
    assert report("adam param", (pd.cpu().double() - pt.detach()).abs().max().item(), 1e-6)
    assert report("adam exp_avg", relerr(m, opt.state[pt]["exp_avg"]), 1e-5)
    assert report("adam exp_avg_sq", relerr(v, opt.state[pt]["exp_avg_sq"]), 1e-5)


def test_synth_matches_host(report):
    n = 100003
    out = torch.empty(n, device=DEV)
    _lib.check(L().svs_fill_uniform(out.data_ptr(), n, 7, (5 << 32) + 11, 2.0, -1.0, S()))
    want = synth.uniform(7, n, (5 << 32) + 11) * np.float32(2.0) - np.float32(1.0)
    assert np.array_equal(out.cpu().numpy(), want)
    mix, voc = torch.empty((3, 1, 64, 32), device=DEV), torch.empty((3, 1, 64, 32), device=DEV)
    _lib.check(L().svs_fill_tiles(mix.data_ptr(), voc.data_ptr(), 3, 64, 32, 9, S()))
    m_np, v_np = synth.tiles(3, 64, 32, first_tile=9)
    assert np.array_equal(mix.cpu().numpy(), m_np) and np.array_equal(voc.cpu().numpy(), v_np)
    masks = synth.dropout_masks(6, seed=99, step=3, rank=2)
    for layer, mk in enumerate(masks):
        out = torch.empty(mk.shape, device=DEV)
        _lib.check(L().svs_dropout_mask(out.data_ptr(), mk.shape[0], mk.shape[1], layer, 99, 3, 2, S()))
        assert np.array_equal(out.cpu().numpy(), mk)
    flat = torch.empty(6 * 496, device=DEV)                     # the five masks in one launch: same bits, same layout
    _lib.check(L().svs_dropout_masks_all(flat.data_ptr(), 6, 99, 3, 2, S()))
    assert np.array_equal(flat.cpu().numpy(), np.concatenate([m.reshape(-1) for m in masks]))


def test_stft_istft(report):
    from oracle import stft_oracle as so
    for n in (20000, 97536, 100000):
        y = synth.audio(n)
        yd = torch.from_numpy(y).to(DEV)
        T = L().svs_stft_frames(n, 768)
        assert T == 1 + n // 768
        mag = torch.empty((513, T), device=DEV)
        ph = torch.empty((513, T, 2), device=DEV)
        _lib.check(L().svs_stft_fwd(yd.data_ptr(), n, 1024, 768, mag.data_ptr(), ph.data_ptr(), S()))
        d = torch.stft(torch.from_numpy(y).double(), 1024, 768, 1024, torch.hann_window(1024, dtype=torch.float64), center=True,
                       pad_mode="constant", return_complex=True)
        assert d.shape == (513, T)
        scale = d.abs().max().item()
        e = (mag.cpu().double() - d.abs()).abs().max().item() / scale
        assert report(f"stft mag n={n} vs torch.stft", e, 2e-6)
        got_c = torch.view_as_complex(ph.cpu().double()) * mag.cpu().double()
        assert report(f"stft complex n={n} vs torch.stft", (got_c - d).abs().max().item() / scale, 2e-6)
        m_o, p_o = so.magphase(so.stft(y))
        assert report(f"stft mag n={n} vs oracle", np.abs(mag.cpu().numpy() - m_o).max() / scale, 2e-6)
        # inverse, unit-phasor form (data.py:159)
        ws = ws_tensor(L().svs_istft_workspace_bytes(1024, 768, T))
        out = torch.empty(768 * (T - 1), device=DEV)
        _lib.check(L().svs_istft(mag.data_ptr(), ph.data_ptr(), 0, 1024, 768, T, out.data_ptr(), ws.data_ptr(), ws.numel(), S()))
        want = so.istft(m_o * p_o)
        interior = slice(1024, -1024)
        e = np.abs(out.cpu().numpy() - want)[interior].max()
        assert report(f"istft n={n} vs oracle (interior)", e, 2e-5)
        # round trip: interior samples come back (hop 768 envelope min 0.043 amplifies fp32 noise ~23x)
        L_ = 768 * (T - 1)
        e = np.abs(out.cpu().numpy()[interior] - y[:L_][interior]).max()
        assert report(f"stft->istft round trip n={n}", e, 5e-5)
    # angle form against torch.istft with the arguments of train.py:51-58
    T = 128
    mag = torch.from_numpy(synth.uniform(3, 513 * T).reshape(513, T))
    mag[0] = 0
    ang = torch.from_numpy((synth.uniform(4, 513 * T) * 2 * np.pi - np.pi).astype(np.float32).reshape(513, T))
    ang[0] = 0
    want = torch.istft(torch.polar(mag.double(), ang.double()), n_fft=1024, hop_length=768, win_length=1024,
                       window=torch.hann_window(1024, dtype=torch.float64), return_complex=False)
    md, ad = mag.to(DEV), ang.to(DEV)
    ws = ws_tensor(L().svs_istft_workspace_bytes(1024, 768, T))
    out = torch.empty(768 * (T - 1), device=DEV)
    _lib.check(L().svs_istft(md.data_ptr(), ad.data_ptr(), 1, 1024, 768, T, out.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    e = (out.cpu().double() - want).abs()[1024:-1024].max().item() / want.abs().max().item()
    assert report("istft angle form vs torch.istft (train.py:51-58)", e, 2e-5)
    # peak normalise (data.py:162-164)
    pk = torch.empty(1, device=DEV)
    _lib.check(L().svs_absmax(out.data_ptr(), out.numel(), pk.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    assert abs(pk.item() - out.abs().max().item()) == 0
    o2 = out.clone()
    _lib.check(L().svs_scale_by_inv(o2.data_ptr(), o2.numel(), pk.data_ptr(), 0.9, S()))
    assert report("peak normalise", (o2 - out / out.abs().max() * 0.9).abs().max().item(), 1e-6)


@pytest.mark.parametrize("hop", [256, 512, 384, 128, 100, 50, 640])
def test_istft_any_hop(hop, report):
    """The inverse for hops other than the config's 768 (data.py:24-25 `--hop_size`; config.py:14-25 records runs at
    HOP_SIZE = 256: four frames per sample): hop < 512 goes through the general overlap-add kernel, hop >= 512 through the
    two-frames-per-sample one.  Checked against torch.istft in float64 with the arguments of train.py:51-58 / data.py:159
    (both phase forms, two channels with a mask), the stft -> istft round trip, and the adjoint (`svs_istft_bwd_mask`)
    against autograd through torch.istft."""
    from svs_unet_pytorch_amd.data import istft, specific_istft, stft_magphase
    win = torch.hann_window(1024, dtype=torch.float64)
    n = 40000 if hop >= 100 else 12000
    y = synth.audio(n)
    mag, ph = stft_magphase(torch.from_numpy(y).to(DEV), 1024, hop)
    T = 1 + n // hop
    assert mag.shape == (513, T)
    d = torch.stft(torch.from_numpy(y).double(), 1024, hop, 1024, win, center=True, pad_mode="constant", return_complex=True)
    assert report(f"stft hop={hop} vs torch.stft", (torch.view_as_complex(torch.view_as_real(ph).cpu().double()) * mag.cpu().double() - d).abs().max().item()
                  / d.abs().max().item(), 2e-6)
    got = istft(mag, ph, 1024, hop).cpu().double()
    want = torch.istft(d, n_fft=1024, hop_length=hop, win_length=1024, window=win, return_complex=False)
    assert got.shape == want.shape == (hop * (T - 1),)
    # hops above n_fft / 2 have envelope troughs (0.043 at hop 768) that amplify fp32 noise, and the envelope falls to zero at
    # both ends of the signal: interior only there; with four or more frames on every sample the whole signal is compared
    interior = slice(None) if hop <= 256 else slice(1024, -1024)
    tol = 2e-5 if 1024 % hop == 0 else 1e-4
    assert report(f"istft hop={hop} phasor form vs torch.istft (interior)", (got - want).abs()[interior].max().item() / want.abs().max().item(), tol)
    assert report(f"stft->istft round trip hop={hop}", (got[interior] - torch.from_numpy(y).double()[:hop * (T - 1)][interior]).abs().max().item(), 5 * tol)
    # angle form, batch of 3 training-shaped tiles (DC row dropped), T = 40 frames
    B, Tt = 3, 40
    m = synth.uniform(3, B * 512 * Tt).reshape(B, 1, 512, Tt)
    a = (synth.uniform(4, B * 512 * Tt) * 2 * np.pi - np.pi).astype(np.float32).reshape(B, 1, 512, Tt)
    got = specific_istft(torch.from_numpy(m).to(DEV), torch.from_numpy(a).to(DEV), 1024, hop).cpu().double()
    m64 = torch.nn.functional.pad(torch.from_numpy(m).double(), (0, 0, 1, 0)).requires_grad_(True)
    a64 = torch.nn.functional.pad(torch.from_numpy(a).double(), (0, 0, 1, 0))
    want = torch.istft(torch.polar(m64, a64).squeeze(1), n_fft=1024, hop_length=hop, win_length=1024, window=win, return_complex=False)
    assert got.shape == (B, 1, hop * (Tt - 1))
    e = (got[:, 0] - want.detach()).abs()[:, interior].max().item() / want.abs().max().item()
    assert report(f"specific_istft hop={hop} vs torch.istft (interior)", e, tol)
    # adjoint: d sum(w * wav) / d mag through torch.istft, against svs_istft_bwd_mask with mix = 1, mask = 1/2 (factor 1/4)
    wgt = torch.from_numpy(synth.uniform(8, B * hop * (Tt - 1)).reshape(B, hop * (Tt - 1))).double() - 0.5
    (want * wgt).sum().backward()
    dmag = m64.grad[:, :, 1:, :]
    ones, half = torch.ones((B, 1, 512, Tt), device=DEV), torch.full((B, 1, 512, Tt), 0.5, device=DEV)
    d_logit = torch.zeros((B, 1, 512, Tt), device=DEV)
    dw_d, a_d = wgt.float().to(DEV), torch.from_numpy(a).to(DEV)       # (named: a temporary's storage would be reused by the next one)
    _lib.check(L().svs_istft_bwd_mask(dw_d.data_ptr(), a_d.data_ptr(), ones.data_ptr(), half.data_ptr(),
                                      d_logit.data_ptr(), 4.0, B, 1024, hop, Tt, S()))
    assert report(f"istft_bwd_mask hop={hop} vs autograd(torch.istft)", (d_logit.cpu().double() - dmag).abs().max().item() / dmag.abs().max().item(), 5 * tol)


def test_specific_istft_golden(golden, report):
    """svs_istft (angle form) through data.specific_istft against the output of the reference's own specific_istft
    (train.py:33-60, captured by oracle/gen_golden_train.py from the reference function object)."""
    from svs_unet_pytorch_amd.data import specific_istft
    g = golden("specific_istft.npz")
    T = 128
    mag = synth.uniform(3, 2 * 512 * T).reshape(2, 1, 512, T)
    ang = (synth.uniform(4, 2 * 512 * T) * 2 * np.pi - np.pi).astype(np.float32).reshape(2, 1, 512, T)
    got = specific_istft(torch.from_numpy(mag).to(DEV), torch.from_numpy(ang).to(DEV)).cpu().numpy()
    want = g["wav"]
    assert got.shape == want.shape == (2, 1, 97536)
    e = np.abs(got - want)[..., 1024:-1024].max() / np.abs(want).max()
    assert report("specific_istft vs reference function (interior)", e, 2e-5)


# ------------------------------------------------------------------------------------------------
# batched / tiled signal kernels (csrc/stft.hip, csrc/mrstft.hip)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [200000, 98304 + 5, 768 * 130])      # T = 261 (3 tiles, ragged), 129 (2 tiles, 127 pad), 131
def test_stft_tiles_istft_tiles(n, report):
    """Forward transform straight into network tiles (DC row dropped, last tile zero-padded) with frame-major phasors, and
    the fused inverse (mask applied on load, overlap-add in LDS, per-channel peak), two channels in one launch, against the
    numpy oracle of data.py:78-109,151-166 / inference.py:65-127."""
    from svs_unet_pytorch_amd.data import istft_from_tiles, stft_to_tiles
    y_np = np.stack([synth.audio(n, 0), synth.audio(n, 1) * 0.5])
    tiles, phase, peak, T = stft_to_tiles(torch.from_numpy(y_np).to(DEV))
    assert T == 1 + n // 768
    n_tiles = T // 128 + (1 if T % 128 else 0)
    assert tiles.shape == (2, n_tiles, 1, 512, 128) and phase.shape == (2, T, 513)
    mask = torch.from_numpy(synth.uniform(9, tiles.numel()).reshape(tiles.shape)).to(DEV)
    got_plain = istft_from_tiles(tiles, None, phase, T).cpu().numpy()
    got_masked = istft_from_tiles(tiles, mask, phase, T, invert=True, peak=0.9).cpu().numpy()
    tiles_h, phase_h, mask_h = tiles.cpu().numpy(), phase.cpu().numpy(), mask.cpu().numpy()
    for c in range(2):
        d = so.stft(y_np[c])
        mag_o, ph_o = so.magphase(d)
        scale = mag_o.max()
        assert report(f"stft_tiles n={n} ch{c} peak", abs(peak[c].item() - scale) / scale, 2e-6)
        full = tiles_h[c, :, 0].transpose(1, 0, 2).reshape(512, n_tiles * 128)
        assert report(f"stft_tiles n={n} ch{c} magnitude", np.abs(full[:, :T] - mag_o[1:]).max() / scale, 2e-6)
        assert np.all(full[:, T:] == 0)                                         # tile padding (inference.py:90-92)
        big = mag_o.T > 1e-3 * scale
        assert report(f"stft_tiles n={n} ch{c} phasors", np.abs(phase_h[c] - ph_o.T)[big].max(), 2e-4)
        # inverse: DC bin absent (inference.py:123 puts a ZERO row back), mixture phase
        spec = np.concatenate([np.zeros((1, T), np.float32), full[:, :T]], axis=0) * ph_o
        want = so.istft(spec)
        e = np.abs(got_plain[c] - want)[1024:-1024].max() / np.abs(want).max()
        assert report(f"istft_tiles n={n} ch{c} (interior)", e, 2e-5)
        mfull = 1.0 - mask_h[c, :, 0].transpose(1, 0, 2).reshape(512, n_tiles * 128)[:, :T]
        spec_m = np.concatenate([np.zeros((1, T), np.float32), full[:, :T] * mfull], axis=0) * ph_o
        want_m = so.istft(spec_m)
        e = np.abs(got_masked[c] / 0.9 * np.abs(want_m).max() - want_m)[1024:-1024].max() / np.abs(want_m).max()
        assert report(f"istft_tiles n={n} ch{c} masked + peak-normalised (interior)", e, 5e-5)
        assert report(f"istft_tiles n={n} ch{c} peak 0.9", abs(np.abs(got_masked[c]).max() - 0.9), 1e-5)


def test_istft_bwd_mask(golden, report):
    """Transpose of specific_istft fused with |S| = mask * mix: against the gradient that autograd gave on the reference's
    own specific_istft (tests/golden/specific_istft.npz) and the oracle's adjoint."""
    g = golden("specific_istft.npz")
    B, T = 2, 128
    ang = (synth.uniform(4, B * 512 * T) * 2 * np.pi - np.pi).astype(np.float32).reshape(B, 1, 512, T)
    wgt = (synth.uniform(8, B * 97536).reshape(B, 1, 97536).astype(np.float64) - 0.5)
    mix = synth.uniform(10, B * 512 * T).reshape(B, 1, 512, T)
    mask = synth.uniform(11, B * 512 * T).reshape(B, 1, 512, T) * 0.8 + 0.1
    d0 = synth.uniform(12, B * 512 * T).reshape(B, 1, 512, T) - 0.5
    d_logit = torch.from_numpy(d0.copy()).to(DEV)
    dw = torch.from_numpy(wgt.astype(np.float32)).to(DEV)
    ang_d, mix_d, mask_d = torch.from_numpy(ang).to(DEV), torch.from_numpy(mix).to(DEV), torch.from_numpy(mask).to(DEV)
    _lib.check(L().svs_istft_bwd_mask(dw.data_ptr(), ang_d.data_ptr(), mix_d.data_ptr(), mask_d.data_ptr(), d_logit.data_ptr(), 0.37, B, 1024,
                                      768, T, S()))
    dmag = so.specific_istft_adjoint(wgt, ang)                                   # pinned against the reference in the CPU suite
    want = d0 + 0.37 * dmag * mix * mask * (1 - mask)
    scale = np.abs(0.37 * dmag * mix * mask * (1 - mask)).max()
    assert report("istft_bwd_mask vs oracle adjoint", np.abs(d_logit.cpu().numpy() - want).max() / scale, 2e-5)
    # and directly against the reference's autograd numbers: rows 0..3 of tile 0
    den = (0.37 * mix * mask * (1 - mask))[0, 0, :4]
    got_dmag = (d_logit.cpu().numpy() - d0)[0, 0, :4] / den
    big = den > 0.02                                      # (dividing the fp32 "+=" back out amplifies its rounding where mix*m*(1-m) is tiny)
    assert report("istft_bwd_mask vs reference autograd (rows 0-3)", np.abs(got_dmag - g["dmag_tile0_rows"])[big].max() / np.abs(g["dmag_tile0_rows"]).max(), 2e-4)


@pytest.mark.parametrize("B,L_", [(2, 97536), (3, 20000)])
def test_mrstft_loss_and_gradient(B, L_, report):
    """Multi-resolution STFT loss (train.py:26,293) value and gradient against the torch restatement of its published
    definition (oracle/mrstft_oracle.py, float64 + autograd).  auraloss itself: parity unpinned."""
    from oracle import mrstft_oracle as mo
    x = (synth.uniform(20, B * L_).reshape(B, L_) - 0.5) * 0.4
    y = x * 0.7 + (synth.uniform(21, B * L_).reshape(B, L_) - 0.5) * 0.2
    want_loss, want_grad = mo.mrstft_loss_and_grad(torch.from_numpy(x).double(), torch.from_numpy(y).double())
    # fp32 noise scale: the log-magnitude term weighs every bin by 1/|X|^2, so the gradient is as accurate as the SMALL
    # bins of an fp32 FFT are -- torch's own fp32 run deviates from its fp64 run by 2e-4 .. 8e-4 depending on the host's FFT
    # library (measured on two CPUs).  The kernel transforms the predicted and the target frame in ONE complex FFT, so a
    # bin's rounding error is relative to max(|X|, |Y|) rather than |X|: a few times torch's fp32 error, hence the 2e-3 floor
    _, g32 = mo.mrstft_loss_and_grad(torch.from_numpy(x), torch.from_numpy(y))
    noise_l2 = ((g32.double() - want_grad).norm() / want_grad.norm()).item()
    noise_max = ((g32.double() - want_grad).abs().max() / want_grad.abs().max()).item()
    xd, yd = torch.from_numpy(x).to(DEV), torch.from_numpy(y).to(DEV)
    ws = ws_tensor(L().svs_mrstft_workspace_bytes(B, L_))
    loss = torch.zeros(1, device=DEV)
    dx = torch.empty_like(xd)
    _lib.check(L().svs_mrstft_loss_fwd_bwd(xd.data_ptr(), yd.data_ptr(), B, L_, 2.5, loss.data_ptr(), dx.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    assert report(f"mrstft loss B={B} L={L_}", abs(loss.item() - want_loss) / want_loss, 1e-5)
    got = dx.cpu().double() / 2.5
    assert report(f"mrstft gradient rel-L2 B={B} L={L_}", ((got - want_grad).norm() / want_grad.norm()).item(), max(6 * noise_l2, 2e-3))
    assert report(f"mrstft gradient max B={B} L={L_}", ((got - want_grad).abs().max() / want_grad.abs().max()).item(), max(6 * noise_max, 3e-3))
    # value only (d_x = NULL) and bitwise reproducibility of the gradient
    loss2, dx2 = torch.zeros(1, device=DEV), torch.empty_like(xd)
    _lib.check(L().svs_mrstft_loss_fwd_bwd(xd.data_ptr(), yd.data_ptr(), B, L_, 2.5, loss2.data_ptr(), dx2.data_ptr(), ws.data_ptr(), ws.numel(), S()))
    assert torch.equal(dx, dx2) and loss.item() == loss2.item()
