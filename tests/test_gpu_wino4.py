"""Winograd F(4,3)-along-y form of the f16x3 3x3 convolution (csrc/conv_wino4_f16x3.hip) through the C ABI: the six-image V writer
against a torch restatement of GroupNorm-apply + transform, the packed weights against G g over the kernel rows, and the convolution
against fp64 torch next to the F(2,3) form, the direct f16x3 kernel and the exact fp32 MFMA engine on the same problem.  The F(4,3)
constants amplify rounding: the bar is 4x the exact-fp32 engine's error (F(2,3) and the direct kernel are held to 2x)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

dev = "cuda"

BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                  dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)


@pytest.fixture(scope="module")
def ops():
    from diffusion_uncertainty_amd import ops as o
    return o


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def v_reference(a):
    """a: the conv's input image [N][H][W][C] (fp64) -> the six F(4,3) input transforms over row groups, [6][N][H/4][W][C]."""
    N, H, W, C = a.shape
    ap = F.pad(a, (0, 0, 0, 0, 1, 4))                         # rows 4j - 1 .. 4j + 4
    d = torch.stack([ap[:, k:k + H:4] for k in range(6)])     # [6][N][H/4][W][C]
    return torch.einsum("fk,knjwc->fnjwc", BT, d)


def decode_v(pl):
    buf = pl.buf.cpu()                                       # [f][tile][C/32][hi | lo][128 positions][32 channels]
    N, H, W, C = pl.shape
    h = buf.view(torch.float16).double()
    v = (h[:, :, :, 0] + h[:, :, :, 1]) / pl.scale          # [f][tile][cb][128][32]
    v = v.permute(0, 1, 3, 2, 4).reshape(6, -1, C)[:, :N * (H // 4) * W]
    return v.reshape(6, N, H // 4, W, C)


def identity_stats(N, groups=32):
    return torch.tensor([0.0, 1.0], device=dev).repeat(N * groups, 1).contiguous()


@pytest.mark.parametrize("N,H,W,C1,C2,resample,film,drop", [
    (2, 16, 16, 64, 0, 0, False, False), (1, 32, 32, 32, 32, 0, True, False), (3, 8, 16, 96, 0, 0, True, True), (5, 8, 8, 64, 32, 0, True, False),
    (2, 16, 4, 32, 0, 0, False, False), (2, 16, 16, 64, 0, 1, False, False), (2, 32, 32, 64, 0, -1, False, False), (1, 64, 64, 256, 0, 0, True, False),
    (2, 16, 16, 768, 512, 0, False, False), (1, 128, 128, 32, 0, 0, True, False)])
def test_wino4_writer_vs_torch(ops, N, H, W, C1, C2, resample, film, drop):
    C = C1 + C2
    x1 = rnd(1, N, H, W, C1) * 2
    x2 = rnd(2, N, H, W, C2) if C2 else None
    gamma, beta = 1 + 0.3 * rnd(3, C), 0.2 * rnd(4, C)
    fl = 0.2 * rnd(5, N, 2 * C) if film else None
    dm = (torch.rand(N, H, W, C, generator=torch.Generator().manual_seed(6)) > 0.1).float() / 0.9 if drop else None
    dv = lambda t: None if t is None else t.to(dev)
    stats = ops.groupnorm_stats(dv(x1), dv(x2))
    pl = ops.groupnorm_apply(dv(x1), stats, dv(gamma), dv(beta), x2=dv(x2), film=dv(fl), act=1, resample=resample, dropmask=dv(dm),
                             split="f16x3w4")
    assert pl.fmt == "f16x3w4" and pl.scale == 0.125 * ops.F16X3_ACT_SCALE
    st = stats.cpu().double().reshape(N, 32, 2)
    x = (x1 if x2 is None else torch.cat([x1, x2], -1)).double()
    cpg = C // 32
    mean, rstd = st[:, :, 0].repeat_interleave(cpg, 1)[:, None, None, :], st[:, :, 1].repeat_interleave(cpg, 1)[:, None, None, :]
    y = (x - mean) * rstd * gamma.double() + beta.double()
    if film:
        y = y * (1 + fl[:, None, None, :C].double()) + fl[:, None, None, C:].double()
    y = F.silu(y)
    if drop:
        y = y * dm.double()
    if resample > 0:
        y = y.repeat_interleave(2, 1).repeat_interleave(2, 2)
    elif resample < 0:
        y = nhwc(F.avg_pool2d(nchw(y), 2))
    assert pl.shape == tuple(y.shape)
    ref = v_reference(y)
    got = decode_v(pl)
    err = (got - ref).abs().max().item()
    print(f"V planes max abs err {err:.2e} (|V| max {ref.abs().max().item():.1f})")
    assert err < 2e-6 * max(1.0, ref.abs().max().item())


def test_wino4_weight_pack(ops):
    """The packed weights decode to U_f = G g over the kernel rows, in the stage order [column tile][channel block][f][dx] x
    [hi | lo][128 channels][32 channels], scaled per output channel by a power of two."""
    Cout, Cin = 160, 64
    w = rnd(7, Cout, Cin, 3, 3) * torch.logspace(-2, 1, Cout).reshape(-1, 1, 1, 1)
    packed = ops.pack_weight_f16x3_wino4(w.to(dev)).cpu()
    ntn, K = 2, 18 * Cin
    plane = ntn * 128 * K
    h = packed[:4 * plane].view(torch.float16).double().reshape(ntn, Cin // 32, 6, 3, 2, 128, 32)      # [tn][cb][f][dx][hi|lo][col][c]
    inv = packed[4 * plane:].view(torch.float32).double()
    assert inv.numel() == 256
    val = (h[:, :, :, :, 0] + h[:, :, :, :, 1]).permute(0, 4, 1, 5, 2, 3).reshape(ntn * 128, Cin, 6, 3)  # [cout][ci][f][dx]
    U = torch.einsum("fk,ockx->ocfx", G, w.double())                                                  # [Cout][Cin][f][dx]
    got = val[:Cout] * inv[:Cout, None, None, None]
    err = ((got - U).abs() / U.abs().amax(dim=(1, 2, 3), keepdim=True)).max().item()
    assert err < 2.0 ** -20, err
    assert (val[Cout:] == 0).all()
    e = torch.log2(inv[:Cout])
    assert (e == e.round()).all()
    assert (val[:Cout].abs().amax(dim=(1, 2, 3)) < 32768).all() and (val[:Cout].abs().amax(dim=(1, 2, 3)) >= 16384 * (1 - 2 ** -10)).all()


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 32, 32), (1, 16, 16, 64, 96), (2, 8, 16, 96, 64), (1, 32, 32, 256, 256),
                                             (5, 16, 16, 128, 512), (3, 64, 64, 64, 256), (2, 128, 128, 32, 128), (4, 16, 8, 64, 36),
                                             (130, 16, 16, 64, 64), (6, 8, 8, 96, 160), (3, 16, 4, 64, 64), (128, 8, 8, 128, 128),
                                             # > 256 tiles with a ragged last row tile and a half-empty column tile: every workgroup runs the
                                             # tile-after-epilogue path (counted wait, always-issued stores) with masked lanes
                                             (1030, 16, 8, 32, 192), (2, 64, 64, 1024, 128)])
def test_conv3x3_wino4_vs_fp64(ops, N, H, W, Cin, Cout):
    assert ops.conv3x3_wino4_supported(N, H, W, Cin, Cout)
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    w = w * torch.logspace(-3, 1, Cout).reshape(-1, 1, 1, 1)
    x = x * torch.logspace(-4, 1.5, Cin).reshape(1, -1, 1, 1)
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    scale_out = ref.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)
    st = identity_stats(N)
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    xd = nhwc(x).to(dev)
    xv4 = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16x3w4")
    wp4 = ops.pack_weight_f16x3_wino4(w.to(dev))
    y = ops.conv3x3_bf16x3(xv4, wp4, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    assert (getattr(y, "_dua_colsums", None) is not None) == ((H * W) % 256 == 0)
    xv2 = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16x3w")
    y2 = ops.conv3x3_bf16x3(xv2, ops.pack_weight_f16x3_wino(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev))
    y32 = ops.conv3x3(xd, ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev))
    err = ((nchw(y.cpu()).double() - ref).abs() / scale_out).max().item()
    err2 = ((nchw(y2.cpu()).double() - ref).abs() / scale_out).max().item()
    err32 = ((nchw(y32.cpu()).double() - ref).abs() / scale_out).max().item()
    rms = lambda t: ((nchw(t.cpu()).double() - ref) / scale_out).pow(2).mean().sqrt().item()
    print(f"F(4,3) f16x3 rel err {err:.2e} (rms {rms(y):.2e})  F(2,3) {err2:.2e} (rms {rms(y2):.2e})  fp32-MFMA {err32:.2e} (rms {rms(y32):.2e})")
    assert err < max(4 * err32, 4e-6), (err, err2, err32)
    if (H * W) % 256 == 0 and Cout % 32 == 0:
        s1 = ops.groupnorm_stats(y)
        s2 = ops.groupnorm_stats(y.clone())
        np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=2e-4, atol=2e-5)
    y0 = ops.conv3x3_bf16x3(xv4, wp4, None, Cout)
    ref0 = F.conv2d(x.double(), w.double(), padding=1)
    assert ((nchw(y0.cpu()).double() - ref0).abs() / scale_out).max().item() < max(4 * err32, 4e-6)
    y1 = ops.conv3x3_bf16x3(xv4, wp4, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    assert torch.equal(y, y1)


def decode_v32(pl):
    buf = pl.buf.cpu().double()                              # [f][tile][C/32][128 positions][32 channels] floats
    N, H, W, C = pl.shape
    v = buf.permute(0, 1, 3, 2, 4).reshape(6, -1, C)[:, :N * (H // 4) * W]
    return v.reshape(6, N, H // 4, W, C)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 32, 32), (1, 16, 16, 64, 96), (2, 8, 16, 96, 64), (1, 32, 32, 256, 256), (5, 16, 16, 128, 160),
                                             (3, 64, 64, 64, 128), (4, 16, 8, 64, 36), (130, 16, 16, 64, 64), (6, 8, 8, 96, 160), (3, 16, 4, 64, 64),
                                             (1030, 16, 8, 32, 100), (2, 128, 128, 32, 64)])
def test_conv3x3_f32_wino4_vs_fp64(ops, N, H, W, Cin, Cout):
    """The F(4,3) form in EXACT fp32 (csrc/conv_wino4_f32.hip: k_wconv432, mode "fp32" of the UNet): the fp32 V writer
    against the torch transform, the conv (bias, residual, column sums; ragged tiles) against fp64 next to the F(2,3) fp32 kernel and the
    direct fp32 MFMA engine.  The F(4,3) constants amplify rounding: the bar is 4x the direct engine's error, as for the f16x3 form."""
    assert ops.conv3x3_wino4_supported(N, H, W, Cin, Cout)
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    w = w * torch.logspace(-3, 1, Cout).reshape(-1, 1, 1, 1)
    x = x * torch.logspace(-4, 1.5, Cin).reshape(1, -1, 1, 1)
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    scale_out = ref.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)
    st = identity_stats(N)
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    xd = nhwc(x).to(dev)
    xv4 = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f32w4")
    assert xv4.fmt == "f32w4"
    vref = v_reference(nhwc(x).double())
    verr = ((decode_v32(xv4) - vref).abs().max() / vref.abs().max()).item()
    assert verr < 1e-6, verr
    wp4 = ops.pack_weight_f32_wino4(w.to(dev))
    y = ops.conv3x3_bf16x3(xv4, wp4, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    assert (getattr(y, "_dua_colsums", None) is not None) == ((H * W) % 256 == 0)
    xv2 = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f32w")
    y2 = ops.conv3x3_bf16x3(xv2, ops.pack_weight_f32_wino(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev))
    y32 = ops.conv3x3(xd, ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev))
    err, err2, err32 = (((nchw(t.cpu()).double() - ref).abs() / scale_out).max().item() for t in (y, y2, y32))
    print(f"F(4,3) fp32 rel err {err:.2e}  F(2,3) fp32 {err2:.2e}  direct fp32-MFMA {err32:.2e}")
    assert err < max(4 * err32, 4e-6), (err, err2, err32)
    if (H * W) % 256 == 0 and Cout % 32 == 0:
        s1 = ops.groupnorm_stats(y)                          # from the conv's column sums
        s2 = ops.groupnorm_stats(y.clone())                  # by the stand-alone kernel
        np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=2e-4, atol=2e-5)
    y0 = ops.conv3x3_bf16x3(xv4, wp4, None, Cout)
    ref0 = F.conv2d(x.double(), w.double(), padding=1)
    assert ((nchw(y0.cpu()).double() - ref0).abs() / scale_out).max().item() < max(4 * err32, 4e-6)
    assert torch.equal(y, ops.conv3x3_bf16x3(xv4, wp4, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True))
    if N <= 6:                                               # the resampled identity skip read in the epilogue (unet_openai.py:238-243)
        for mode, rshape in ((1, (N, Cout, H // 2, W // 2)), (-1, (N, Cout, 2 * H, 2 * W))):
            r = rnd(42 + mode, *rshape)
            full = F.interpolate(r, scale_factor=2, mode="nearest") if mode > 0 else F.avg_pool2d(r, 2)
            ya = ops.conv3x3_bf16x3(xv4, wp4, None, Cout, residual=nhwc(r).to(dev), res_resample=mode)
            yb = ops.conv3x3_bf16x3(xv4, wp4, None, Cout, residual=nhwc(full).to(dev))
            assert torch.equal(ya, yb), mode


def test_fp32_unet_runs_f43_and_matches_reference(golden):
    """Mode "fp32" of the full-size ADM-64 / ADM-128 with the default form choice (F(4,3) wherever its kernel takes the shape): inside the
    1e-3 bar against the reference's own output (G5) at the distance of the F(2,3) and direct forms; block-level C ABI == per-op
    composition bit for bit; deterministic."""
    from conftest import T
    from oracle import adm
    from test_gpu_models import build_adm
    from diffusion_uncertainty_amd.guided_diffusion.unet_openai import _Ctx
    for name, cfg in (("adm64", adm.ADM64), ("adm128", adm.ADM128)):
        g = golden(f"g5_{name}_full")
        m, _ = build_adm(dict(cfg), int(g["seed"]))
        m.precision = "fp32"
        x, t, y = T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev)
        hw = int(g["x"].shape[-1])
        assert _Ctx(m).wino_form(1, hw, hw, cfg["model_channels"], cfg["model_channels"]) == 4
        w4 = m(x, t, y=y)
        assert torch.equal(w4, m(x, t, y=y))
        m.use_block_abi = False
        wop = m(x, t, y=y)
        m.use_block_abi = True
        m.winograd = 2
        w2 = m(x, t, y=y)
        m.winograd = False
        d = m(x, t, y=y)
        e4, e2, ed = (np.abs(a.cpu().numpy() - g["out"]).max() for a in (w4, w2, d))
        print(f"{name} fp32: F(4,3) err {e4:.3e}, F(2,3) err {e2:.3e}, direct err {ed:.3e}, |F(4,3) - direct| {(w4 - d).abs().max().item():.3e}")
        assert e4 < 1e-3 and e4 < 2 * max(e2, ed) + 2e-6
        assert (w4 - d).abs().max().item() < 5e-5 and not torch.equal(w4, w2)
        assert torch.equal(w4, wop)
        del m
        torch.cuda.empty_cache()


def test_wino4_resampled_residual_in_epilogue(ops):
    """The identity skip of an ADM up / down block read through the resampling in the conv's epilogue (unet_openai.py:238-243)."""
    N, H, W, C = 3, 16, 16, 64
    x, w = rnd(40, N, C, H, W), rnd(41, C, C, 3, 3) / math.sqrt(9 * C)
    st = identity_stats(N)
    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    xv = ops.groupnorm_apply(nhwc(x).to(dev), st, one, zero, act=0, split="f16x3w4")
    wp = ops.pack_weight_f16x3_wino4(w.to(dev))
    for mode, rshape in ((1, (N, C, H // 2, W // 2)), (-1, (N, C, 2 * H, 2 * W))):
        r = rnd(42 + mode, *rshape)
        full = F.interpolate(r, scale_factor=2, mode="nearest") if mode > 0 else F.avg_pool2d(r, 2)
        y = ops.conv3x3_bf16x3(xv, wp, None, C, residual=nhwc(r).to(dev), res_resample=mode)
        y_ref = ops.conv3x3_bf16x3(xv, wp, None, C, residual=nhwc(full).to(dev))
        assert torch.equal(y, y_ref), mode


def test_conv3x3_wino4_unsupported_shapes(ops):
    assert ops.conv3x3_wino4_supported(2, 8, 8, 64, 64)              # small map: taken, without the column-sum output
    assert not ops.conv3x3_wino4_supported(2, 6, 16, 64, 64)         # H % 4
    assert not ops.conv3x3_wino4_supported(2, 4, 8, 64, 64)          # H*W % 64
    assert not ops.conv3x3_wino4_supported(2, 16, 24, 64, 64)        # W not a power of two
    assert not ops.conv3x3_wino4_supported(2, 16, 256, 64, 64)       # W > 128
    assert not ops.conv3x3_wino4_supported(2, 16, 16, 48, 64)        # Cin % 32


def test_wino4_pick_rule(ops):
    """dua_conv3x3_f16x3_wino_pick: F(4,3) where whole rounds of tiles on 256 workgroups favour it (the 128^2 .. 32^2 levels of ADM-128 at
    batch 128), F(2,3) where its twice-as-many tiles fill the chip better (the 8^2 level), nothing where neither kernel takes the shape."""
    assert ops.conv3x3_wino_pick(128, 128, 128, 256, 256) == 4
    assert ops.conv3x3_wino_pick(128, 64, 64, 512, 512) == 4
    assert ops.conv3x3_wino_pick(128, 32, 32, 768, 512) == 4
    assert ops.conv3x3_wino_pick(128, 8, 8, 1024, 1024) == 2
    assert ops.conv3x3_wino_pick(128, 16, 16, 768, 768) == 2         # 384 F(4,3) tiles = 2 rounds at 75 %
    assert ops.conv3x3_wino_pick(128, 16, 16, 1024, 1024) == 4
    assert ops.conv3x3_wino_pick(2, 128, 128, 256, 256) == 2          # small batch: 128 F(4,3) tiles would leave half the chip idle
    assert ops.conv3x3_wino_pick(2, 6, 16, 64, 64) == 2 and ops.conv3x3_wino_pick(2, 16, 256, 64, 64) == 0


def test_winograd43_unet_vs_reference_and_direct(golden):
    """Full-size ADM-64 / ADM-128 in f16x3 with EVERY 3x3 conv the F(4,3) kernel takes in that form (model.winograd = 4): inside the 1e-3
    bar against the reference's own output (G5) at the distance of the direct form, within 5e-5 of the direct form, deterministic, and
    the block-level C-ABI path bit-identical to the per-op composition with the same operands."""
    from conftest import T
    from oracle import adm
    from test_gpu_models import build_adm
    for name, cfg in (("adm64", adm.ADM64), ("adm128", adm.ADM128)):
        g = golden(f"g5_{name}_full")
        m, _ = build_adm(dict(cfg), int(g["seed"]))
        m.precision = "f16x3"
        x, t, y = T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev)
        m.winograd = 4
        w = m(x, t, y=y)
        w2 = m(x, t, y=y)
        m.use_block_abi = False
        wop = m(x, t, y=y)
        m.use_block_abi = True
        m.winograd = 2
        v = m(x, t, y=y)
        m.winograd = False
        d = m(x, t, y=y)
        m.precision = "fp32"
        e32 = np.abs(m(x, t, y=y).cpu().numpy() - g["out"]).max()
        ew, ev, ed = (np.abs(a.cpu().numpy() - g["out"]).max() for a in (w, v, d))
        print(f"{name} f16x3: F(4,3) err {ew:.3e}, F(2,3) err {ev:.3e}, direct err {ed:.3e}, exact-fp32 engine err {e32:.3e}, |F(4,3) - direct| {(w - d).abs().max().item():.3e}")
        assert ew < 1e-3 and ew < 2 * max(ed, e32) + 2e-6
        assert (w - d).abs().max().item() < 5e-5 and not torch.equal(w, v)
        assert torch.equal(w, w2) and torch.equal(w, wop)
        del m
        torch.cuda.empty_cache()


@pytest.mark.parametrize("N,H,Cin,Cout", [(128, 128, 256, 256), (128, 64, 512, 512)])
def test_headline_geometry_writer_wconv4(ops, N, H, Cin, Cout):
    """The geometry of the headline job itself in the F(4,3) form (BASELINE configs[1]: batch 128 at 128^2 x 256 channels = 2^19 positions
    x 6 frequencies, 6.4 GB of V planes, 8 192 tiles on 256 persistent workgroups; and 64^2 x 512): GroupNorm + FiLM + SiLU writer ->
    k_wconv4 with a residual and the column sums, against the direct f16x3 kernel on planes written from the same input (every tile index,
    descriptor rebase and persistent-loop path of the full-size launch), and against fp64 on sampled images.  Shapes of init_model.py:21 at
    --batch-size 128 (readme.md:116)."""
    torch.cuda.empty_cache()
    W = H
    assert ops.conv3x3_wino_pick(N, H, W, Cin, Cout) == 4
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(N, H, W, Cin, device=dev, generator=g)
    x += 0.5 * torch.randn(N, 1, 1, Cin, device=dev, generator=g)                   # per-image channel offsets: statistics that differ per image
    gamma, beta = torch.randn(Cin, device=dev, generator=g), 0.3 * torch.randn(Cin, device=dev, generator=g)
    film = 0.5 * torch.randn(N, 2 * Cin, device=dev, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(9 * Cin)
    b = 0.1 * torch.randn(Cout, device=dev, generator=g)
    res = torch.randn(N, H, W, Cout, device=dev, generator=g)
    st = ops.groupnorm_stats(x)
    xv = ops.groupnorm_apply(x, st, gamma, beta, film=film, act=1, split="f16x3w4")
    yw = ops.conv3x3_bf16x3(xv, ops.pack_weight_f16x3_wino4(w), b, Cout, residual=res, colsum=True)
    del xv
    xs = ops.groupnorm_apply(x, st, gamma, beta, film=film, act=1, split="f16x3")
    yd = ops.conv3x3_bf16x3(xs, ops.pack_weight_f16x3(w), b, Cout, residual=res, colsum=True)
    del xs
    scale = yd.abs().max().item()
    dmax = 0.0
    for i in range(0, N, 16):                                                        # chunked: keeps the temporaries small
        dmax = max(dmax, (yw[i:i + 16] - yd[i:i + 16]).abs().max().item())
    print(f"N={N} {H}^2 {Cin}->{Cout}: max |F(4,3) - direct| = {dmax:.3e} (output scale {scale:.2f})")
    assert dmax < 5e-5 * max(1.0, scale)
    s_w, s_ref = ops.groupnorm_stats(yw), ops.groupnorm_stats(yd.clone())            # (a clone carries no column sums: the statistics kernel)
    np.testing.assert_allclose(s_w.cpu().numpy(), s_ref.cpu().numpy(), rtol=3e-4, atol=3e-5)
    stc = st.cpu().double().reshape(N, 32, 2)
    for i in (0, 41, 86, N - 1):
        xi = x[i].cpu().double().permute(2, 0, 1)[None]                               # [1, C, H, W]
        mean, rstd = stc[i, :, 0].repeat_interleave(Cin // 32), stc[i, :, 1].repeat_interleave(Cin // 32)
        a = (xi - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1) * gamma.cpu().double().view(1, -1, 1, 1) + beta.cpu().double().view(1, -1, 1, 1)
        fi = film[i].cpu().double()
        a = F.silu(a * (1 + fi[:Cin]).view(1, -1, 1, 1) + fi[Cin:].view(1, -1, 1, 1))
        ref = F.conv2d(a, w.cpu().double(), b.cpu().double(), padding=1) + res[i].cpu().double().permute(2, 0, 1)[None]
        e = (yw[i].cpu().double().permute(2, 0, 1)[None] - ref).abs().max().item()
        print(f"  image {i}: max |F(4,3) - fp64| = {e:.3e}")
        assert e < 3e-5 * max(1.0, ref.abs().max().item()), (i, e)
    del x, yw, yd, res
    torch.cuda.empty_cache()


def test_wino4_range_counters(ops):
    """The F(4,3) V writer counts the transformed operands whose scaled magnitude (scale 8) leaves fp16's range, as the other f16x3
    writers do (dua_range_check, tests/test_gpu_range.py)."""
    N, H, W, C = 2, 16, 16, 64
    st = identity_stats(N)
    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    x = rnd(1, N, H, W, C).clamp(-8, 8)
    x[x.abs() < 1e-2] = 0.5
    x[0, 3, 4, 5] = 9000.0                                         # V = 4 d .. : |V| * 8 far beyond 65504
    prev = ops.range_check(True)
    try:
        ops.groupnorm_apply(x.to(dev), st, one, zero, act=0, split="f16x3w4")
        sat, sub = ops.range_check_read()
    finally:
        ops.range_check(prev)
    v = v_reference(x.double()).float() * 8.0
    assert sat == int((v.abs() > 65504).sum()) and sat > 0
    assert sub == int(((v.abs() > 0) & (v.abs() < 2.0 ** -14)).sum())


def test_wino4_zero_padding_forms_agree(ops):
    """k_wconv4 takes the dx = -1 / +1 fragments of its image-edge lanes from an LDS address beyond the allocation (zeros), like k_wconv.  The
    same kernel built with -DDUA_RS_ZERO_SELECT=1 (scripts/diag/build/libdua_hip_select.so, made by __graft_entry__.build()) reads the
    in-range fragment and zeroes it in registers; both must agree bit for bit."""
    import ctypes
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alt = os.path.join(root, "scripts", "diag", "build", "libdua_hip_select.so")
    if not os.path.exists(alt):
        import __graft_entry__
        __graft_entry__.build()
    A = ctypes.CDLL(alt)
    from diffusion_uncertainty_amd import _lib
    fn = A.dua_conv3x3_f16x3_wino4
    fn.restype, fn.argtypes = _lib.SIGNATURES["dua_conv3x3_f16x3_wino4"]
    for (N, H, W, Cin, Cout) in [(8, 32, 32, 64, 256), (2, 128, 128, 32, 128), (16, 16, 16, 96, 256), (130, 8, 8, 64, 128), (9, 64, 64, 128, 64)]:
        x, w, b = rnd(50, N, Cin, H, W), rnd(51, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(52, Cout) * 0.1
        xd = nhwc(x).to(dev)
        xv = ops.groupnorm_apply(xd, identity_stats(N), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3w4")
        packed = ops.pack_weight_f16x3_wino4(w.to(dev))
        bd = b.to(dev)
        y0 = ops.conv3x3_bf16x3(xv, packed, bd, Cout)
        y1 = torch.empty_like(y0)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = fn(ctypes.c_void_p(xv.buf.data_ptr()), ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(bd.data_ptr()), None, 0,
                ctypes.c_void_p(y1.data_ptr()), N, H, W, Cin, Cout, None, xv.scale, st)
        assert rc == 0
        assert torch.equal(y0, y1), (N, H, W, Cin, Cout)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        assert (nchw(y0.cpu()).double() - ref).abs().max().item() < 3e-5
