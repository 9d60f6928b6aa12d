"""Winograd F(2,3)-along-y form of the f16x3 3x3 convolution (csrc/conv_wino_f16x3.hip) through the C ABI: the V-plane writer
against a plain torch restatement of GroupNorm-apply + transform, the packed weights against the transform of the kernel rows, and
the convolution against fp64 torch, held to the error of the exact fp32 MFMA engine (the bar of the direct f16x3 kernel)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

dev = "cuda"


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
    """a: the conv's input image [N][H][W][C] (fp64) -> the four F(2,3) input transforms over row pairs, [4][N][H/2][W][C]."""
    N, H, W, C = a.shape
    ap = F.pad(a, (0, 0, 0, 0, 1, 1))                         # one zero row above and below
    d0, d1, d2, d3 = ap[:, 0:H:2], ap[:, 1:H + 1:2], ap[:, 2:H + 2:2], ap[:, 3:H + 3:2]
    return torch.stack([d0 - d2, d1 + d2, d2 - d1, d1 - d3])


def decode_v(pl):
    buf = pl.buf.cpu()                                       # [f][tile][C/32][hi | lo][128 positions][32 channels]
    N, H, W, C = pl.shape
    h = buf.view(torch.float16).double()
    v = (h[:, :, :, 0] + h[:, :, :, 1]) / pl.scale          # [f][tile][cb][128][32]
    v = v.permute(0, 1, 3, 2, 4).reshape(4, -1, C)[:, :N * (H // 2) * W]
    return v.reshape(4, N, H // 2, W, C)


def identity_stats(N, groups=32):
    return torch.tensor([0.0, 1.0], device=dev).repeat(N * groups, 1).contiguous()


@pytest.mark.parametrize("N,H,W,C1,C2,resample,film,drop", [
    (2, 16, 16, 64, 0, 0, False, False), (1, 32, 32, 32, 32, 0, True, False), (3, 8, 16, 96, 0, 0, True, True), (5, 8, 8, 64, 32, 0, True, False), (3, 8, 4, 32, 0, 0, False, False),
    (2, 16, 16, 64, 0, 1, False, False), (2, 32, 32, 64, 0, -1, False, False), (1, 64, 64, 256, 0, 0, True, False),
    (2, 16, 16, 768, 512, 0, False, False)])
def test_wino_writer_vs_torch(ops, N, H, W, C1, C2, resample, film, drop):
    """GroupNorm * gamma + beta (+FiLM) -> SiLU (-> dropout mask / 2x resampling) -> V_f planes: against the same chain in fp64 torch
    (virtual concat, both resampling modes, FiLM, dropout mask); hi + lo carries 22 bits of V * scale."""
    C = C1 + C2
    x1 = rnd(1, N, H, W, C1) * 2
    x2 = rnd(2, N, H, W, C2) if C2 else None
    gamma, beta = 1 + 0.3 * rnd(3, C), 0.2 * rnd(4, C)
    fl = 0.2 * rnd(5, N, 2 * C) if film else None
    dm = (torch.rand(N, H, W, C, generator=torch.Generator().manual_seed(6)) > 0.1).float() / 0.9 if drop else None
    dv = lambda t: None if t is None else t.to(dev)
    stats = ops.groupnorm_stats(dv(x1), dv(x2))
    pl = ops.groupnorm_apply(dv(x1), stats, dv(gamma), dv(beta), x2=dv(x2), film=dv(fl), act=1, resample=resample, dropmask=dv(dm),
                             split="f16x3w")
    assert pl.fmt == "f16x3w" and pl.scale == 0.5 * ops.F16X3_ACT_SCALE
    # reference chain from the kernel's own statistics (they have their own parity test)
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


def test_wino_weight_pack(ops):
    """The packed weights decode to U_f = G g over the kernel rows, in the stage order [column tile][channel block][fi][dx] x
    [hi | lo][group][128 channels][32 channels], scaled per output channel by a power of two."""
    Cout, Cin = 160, 64
    w = rnd(7, Cout, Cin, 3, 3) * torch.logspace(-2, 1, Cout).reshape(-1, 1, 1, 1)
    packed = ops.pack_weight_f16x3_wino(w.to(dev)).cpu()
    ntn, K = 2, 6 * Cin
    plane = ntn * 256 * K
    h = packed[:4 * plane].view(torch.float16).double().reshape(ntn, Cin // 32, 2, 3, 2, 2, 128, 32)   # [tn][cb][fi][dx][hi|lo][g][col][c]
    inv = packed[4 * plane:].view(torch.float32).double()
    assert inv.numel() == 256
    val = (h[:, :, :, :, 0] + h[:, :, :, :, 1]).permute(0, 4, 5, 1, 2, 3, 6)      # [tn][g][col][cb][fi][dx][c]
    g0, g1, g2 = w[:, :, 0].double(), w[:, :, 1].double(), w[:, :, 2].double()    # [Cout][Cin][dx]
    U = [g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2]
    fmap = {(0, 0): 1, (0, 1): 2, (1, 0): 0, (1, 1): 3}
    for (g, fi), f in fmap.items():
        got = val[:, g, :, :, fi]                                                # [tn][col][cb][dx][c]
        got = got.permute(0, 1, 2, 4, 3).reshape(ntn * 128, Cin, 3)[:Cout] * inv[:Cout, None, None]
        err = ((got - U[f]).abs() / U[f].abs().amax(dim=(1, 2), keepdim=True)).max().item()
        assert err < 2.0 ** -20, (g, fi, err)
        pad = val[:, g, :, :, fi].permute(0, 1, 2, 4, 3).reshape(ntn * 128, Cin, 3)[Cout:]
        assert (pad == 0).all()
    e = torch.log2(inv[:Cout])
    assert (e == e.round()).all()


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 32, 32), (1, 16, 16, 64, 96), (2, 8, 16, 96, 64), (1, 32, 32, 256, 256),
                                             (5, 16, 16, 128, 512), (3, 64, 64, 64, 256), (2, 128, 128, 32, 128), (4, 16, 8, 64, 36),
                                             (130, 16, 16, 64, 64), (6, 8, 8, 96, 160), (3, 8, 4, 64, 64), (128, 8, 8, 128, 128),
                                             # 514 tiles on 256 workgroups with a ragged last row tile and a half-empty column tile: every
                                             # workgroup runs the tile-after-epilogue path (counted wait, always-issued stores) with masked lanes
                                             (513, 16, 8, 32, 192)])
def test_conv3x3_wino_vs_fp64(ops, N, H, W, Cin, Cout):
    """The whole Winograd path (writer -> packed weights -> kernel) against fp64, next to the direct f16x3 kernel and the exact fp32
    MFMA engine on the same problem: same bar as test_conv3x3_f16x3 (within 2x of the fp32 engine's error)."""
    assert ops.conv3x3_wino_supported(N, H, W, Cin, Cout)
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    w = w * torch.logspace(-3, 1, Cout).reshape(-1, 1, 1, 1)
    x = x * torch.logspace(-4, 1.5, Cin).reshape(1, -1, 1, 1)
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    scale_out = ref.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)
    st = identity_stats(N)
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    xd = nhwc(x).to(dev)
    xv = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16x3w")
    xs = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16x3")
    y = ops.conv3x3_bf16x3(xv, ops.pack_weight_f16x3_wino(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    assert (getattr(y, "_dua_colsums", None) is not None) == ((H * W) % 128 == 0)
    yd = ops.conv3x3_bf16x3(xs, ops.pack_weight_f16x3(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    y32 = ops.conv3x3(xd, ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev))
    err = ((nchw(y.cpu()).double() - ref).abs() / scale_out).max().item()
    errd = ((nchw(yd.cpu()).double() - ref).abs() / scale_out).max().item()
    err32 = ((nchw(y32.cpu()).double() - ref).abs() / scale_out).max().item()
    print(f"winograd f16x3 rel err {err:.2e}  direct f16x3 {errd:.2e}  fp32-MFMA {err32:.2e}")
    assert err < max(2 * err32, 2e-6), (err, errd, err32)
    # GroupNorm statistics from the epilogue's column sums equal those recomputed from the output
    if (H * W) % 128 == 0 and Cout % 32 == 0:
        s1 = ops.groupnorm_stats(y)
        s2 = ops.groupnorm_stats(y.clone())
        np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=2e-4, atol=2e-5)
    # no residual / no bias / no statistics
    y0 = ops.conv3x3_bf16x3(xv, ops.pack_weight_f16x3_wino(w.to(dev)), None, Cout)
    ref0 = F.conv2d(x.double(), w.double(), padding=1)
    assert ((nchw(y0.cpu()).double() - ref0).abs() / scale_out).max().item() < max(2 * err32, 2e-6)
    # deterministic
    y1 = ops.conv3x3_bf16x3(xv, ops.pack_weight_f16x3_wino(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    assert torch.equal(y, y1)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 64, 96), (3, 64, 64, 64, 256), (130, 16, 16, 64, 64), (513, 16, 8, 32, 192), (128, 8, 8, 128, 128)])
def test_conv3x3_f16w_single_product(ops, N, H, W, Cin, Cout):
    """Mode "f16w" (csrc/conv_wino_f16x3.hip k_wconv1): the Winograd conv with ONE fp16 product per tap on the hi images only.
    (a) operands that are exactly representable in the hi planes (small dyadic rationals): the lo halves are zero, so the single-product
    kernel must equal the three-product kernel BIT FOR BIT (same hi*hi products in the same order) and fp64 to fp32-accumulation level;
    (b) random operands: an 11-bit-operand error (fp16 autocast class), far above f16x3's and far below a wrong-tap error."""
    assert ops.conv3x3_wino_supported(N, H, W, Cin, Cout)
    g = torch.Generator().manual_seed(7)
    st = identity_stats(N)
    one, zero = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
    # (a) exact operands
    x = torch.randint(-64, 65, (N, Cin, H, W), generator=g).float() / 16.0
    w = (2 * torch.randint(-16, 17, (Cout, Cin, 3, 3), generator=g)).float() / 64.0
    b, res = rnd(31, Cout), rnd(32, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    xd = nhwc(x).to(dev)
    wp = ops.pack_weight_f16x3_wino(w.to(dev))
    v1 = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16w")
    v3 = ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16x3w")
    assert torch.equal(decode_v(v1), decode_v(v3))          # the same hi images, and nothing left for the lo images (valid positions)
    y1 = ops.conv3x3_bf16x3(v1, wp, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    y3 = ops.conv3x3_bf16x3(v3, wp, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    assert torch.equal(y1, y3)
    e = ((nchw(y1.cpu()).double() - ref).abs().max() / ref.abs().max()).item()
    assert e < 2e-6, e
    if (H * W) % 128 == 0:
        assert torch.equal(y1._dua_colsums, y3._dua_colsums)
    # (b) random operands
    x, w = rnd(33, N, Cin, H, W), rnd(34, Cout, Cin, 3, 3) / math.sqrt(9 * Cin)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xd = nhwc(x).to(dev)
    wp = ops.pack_weight_f16x3_wino(w.to(dev))
    y1 = ops.conv3x3_bf16x3(ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16w"), wp, None, Cout)
    y3 = ops.conv3x3_bf16x3(ops.groupnorm_apply(xd, st, one, zero, act=0, split="f16x3w"), wp, None, Cout)
    sc = ref.abs().max()
    e1, e3 = ((nchw(y1.cpu()).double() - ref).abs().max() / sc).item(), ((nchw(y3.cpu()).double() - ref).abs().max() / sc).item()
    print(f"f16w rel err {e1:.2e} (f16x3 {e3:.2e})")
    assert 10 * e3 < e1 < 2e-3, (e1, e3)


@pytest.mark.parametrize("N,H,Cin,Cout", [(128, 128, 256, 256), (128, 64, 512, 512)])
def test_headline_geometry_writer_wconv(ops, N, H, Cin, Cout):
    """The geometry of the headline job itself (BASELINE configs[1]: batch 128 at 128^2 x 256 channels = 2^21 positions, 8.6 GB of V
    planes, 32 768 tiles on 256 persistent workgroups; and 64^2 x 512): GroupNorm + FiLM + SiLU writer -> k_wconv with a residual and
    the column sums, against the direct f16x3 kernel on planes written from the same input (every tile index, descriptor rebase and
    persistent-loop path of the full-size launch), and against fp64 on sampled images.  Shapes of init_model.py:21 at --batch-size 128
    (readme.md:116)."""
    torch.cuda.empty_cache()
    W = H
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(N, H, W, Cin, device=dev, generator=g)
    x += 0.5 * torch.randn(N, 1, 1, Cin, device=dev, generator=g)                   # per-image channel offsets: statistics that differ per image
    gamma, beta = torch.randn(Cin, device=dev, generator=g), 0.3 * torch.randn(Cin, device=dev, generator=g)
    film = 0.5 * torch.randn(N, 2 * Cin, device=dev, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) / math.sqrt(9 * Cin)
    b = 0.1 * torch.randn(Cout, device=dev, generator=g)
    res = torch.randn(N, H, W, Cout, device=dev, generator=g)
    st = ops.groupnorm_stats(x)
    xv = ops.groupnorm_apply(x, st, gamma, beta, film=film, act=1, split="f16x3w")
    yw = ops.conv3x3_bf16x3(xv, ops.pack_weight_f16x3_wino(w), b, Cout, residual=res, colsum=True)
    del xv
    xs = ops.groupnorm_apply(x, st, gamma, beta, film=film, act=1, split="f16x3")
    yd = ops.conv3x3_bf16x3(xs, ops.pack_weight_f16x3(w), b, Cout, residual=res, colsum=True)
    del xs
    scale = yd.abs().max().item()
    dmax = 0.0
    for i in range(0, N, 16):                                                        # chunked: keeps the temporaries small
        dmax = max(dmax, (yw[i:i + 16] - yd[i:i + 16]).abs().max().item())
    print(f"N={N} {H}^2 {Cin}->{Cout}: max |winograd - direct| = {dmax:.3e} (output scale {scale:.2f})")
    assert dmax < 3e-5 * max(1.0, scale)
    # the next GroupNorm's statistics from the epilogue's column sums: both kernels, every image
    s_w, s_d = ops.groupnorm_stats(yw), ops.groupnorm_stats(yd)
    s_ref = ops.groupnorm_stats(yd.clone())                                          # (a clone carries no column sums: the statistics kernel)
    np.testing.assert_allclose(s_w.cpu().numpy(), s_ref.cpu().numpy(), rtol=3e-4, atol=3e-5)
    np.testing.assert_allclose(s_d.cpu().numpy(), s_ref.cpu().numpy(), rtol=3e-4, atol=3e-5)
    # fp64 on sampled images (first, last, and two in the middle whose tiles sit behind a few thousand others)
    stc = st.cpu().double().reshape(N, 32, 2)
    for i in (0, 41, 86, N - 1):
        xi = x[i].cpu().double().permute(2, 0, 1)[None]                               # [1, C, H, W]
        mean, rstd = stc[i, :, 0].repeat_interleave(Cin // 32), stc[i, :, 1].repeat_interleave(Cin // 32)
        a = (xi - mean.view(1, -1, 1, 1)) * rstd.view(1, -1, 1, 1) * gamma.cpu().double().view(1, -1, 1, 1) + beta.cpu().double().view(1, -1, 1, 1)
        fi = film[i].cpu().double()
        a = F.silu(a * (1 + fi[:Cin]).view(1, -1, 1, 1) + fi[Cin:].view(1, -1, 1, 1))
        ref = F.conv2d(a, w.cpu().double(), b.cpu().double(), padding=1) + res[i].cpu().double().permute(2, 0, 1)[None]
        e = (yw[i].cpu().double().permute(2, 0, 1)[None] - ref).abs().max().item()
        print(f"  image {i}: max |winograd - fp64| = {e:.3e}")
        assert e < 2e-5 * max(1.0, ref.abs().max().item()), (i, e)
    del x, yw, yd, res
    torch.cuda.empty_cache()


def test_conv3x3_wino_unsupported_shapes(ops):
    assert ops.conv3x3_wino_supported(2, 8, 8, 64, 64)              # small map: taken, without the column-sum output
    assert not ops.conv3x3_wino_supported(2, 15, 16, 64, 64)        # odd H
    assert not ops.conv3x3_wino_supported(2, 4, 4, 64, 64)          # H*W % 32
    assert not ops.conv3x3_wino_supported(2, 16, 24, 64, 64)        # W not a power of two
    assert not ops.conv3x3_wino_supported(1, 256, 256, 128, 128)    # W > 128
    assert not ops.conv3x3_wino_supported(2, 16, 16, 48, 64)        # Cin % 32
    assert ops.conv3x3_wino_supported(128, 128, 128, 512, 256)


def test_winograd_unet_vs_reference_and_direct(golden):
    """Full-size ADM-64 / ADM-128 in f16x3 with the 3x3 convs in the Winograd form (the default) and in the direct form: both inside
    the 1e-3 bar against the reference's own output (G5), within 3e-5 of each other, the Winograd run deterministic, and the block-level
    C-ABI path bit-identical to the per-op composition with the Winograd operands."""
    from conftest import T
    from oracle import adm
    from test_gpu_models import build_adm
    for name, cfg in (("adm64", adm.ADM64), ("adm128", adm.ADM128)):
        g = golden(f"g5_{name}_full")
        m, _ = build_adm(dict(cfg), int(g["seed"]))
        m.precision = "f16x3"
        x, t, y = T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev)
        assert m.winograd
        w = m(x, t, y=y)
        w2 = m(x, t, y=y)
        m.use_block_abi = False
        wop = m(x, t, y=y)
        m.use_block_abi = True
        m.winograd = False
        d = m(x, t, y=y)
        ew, ed = np.abs(w.cpu().numpy() - g["out"]).max(), np.abs(d.cpu().numpy() - g["out"]).max()
        print(f"{name} f16x3: winograd err {ew:.3e}, direct err {ed:.3e}, |winograd - direct| {(w - d).abs().max().item():.3e}")
        assert ew < 1e-3 and ed < 1e-3 and ew < 2 * ed + 2e-6
        assert (w - d).abs().max().item() < 3e-5 and not torch.equal(w, d)
        assert torch.equal(w, w2) and torch.equal(w, wop)
        del m
        torch.cuda.empty_cache()


def test_wino_zero_padding_forms_agree(ops):
    """k_wconv takes the dx = -1 / +1 fragments of its image-edge lanes from an LDS address beyond the allocation (zeros).  The same
    kernel built with -DDUA_RS_ZERO_SELECT=1 (scripts/diag/build/libdua_hip_select.so, made by __graft_entry__.build()) reads the
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
    fn = A.dua_conv3x3_f16x3_wino
    fn.restype, fn.argtypes = _lib.SIGNATURES["dua_conv3x3_f16x3_wino"]
    for (N, H, W, Cin, Cout) in [(8, 32, 32, 64, 256), (2, 128, 128, 32, 128), (16, 16, 16, 96, 256), (130, 8, 8, 64, 128)]:
        x, w, b = rnd(50, N, Cin, H, W), rnd(51, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(52, Cout) * 0.1
        xd = nhwc(x).to(dev)
        xv = ops.groupnorm_apply(xd, identity_stats(N), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3w")
        packed = ops.pack_weight_f16x3_wino(w.to(dev))
        bd = b.to(dev)
        y0 = ops.conv3x3_bf16x3(xv, packed, bd, Cout)
        y1 = torch.empty_like(y0)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = fn(ctypes.c_void_p(xv.buf.data_ptr()), ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(bd.data_ptr()), None, 0,
                ctypes.c_void_p(y1.data_ptr()), N, H, W, Cin, Cout, None, xv.scale, st)
        assert rc == 0
        assert torch.equal(y0, y1), (N, H, W, Cin, Cout)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        assert (nchw(y0.cpu()).double() - ref).abs().max().item() < 2e-5
        # the exact-fp32 kernel likewise
        fn32 = A.dua_conv3x3_f32_wino
        fn32.restype, fn32.argtypes = _lib.SIGNATURES["dua_conv3x3_f32_wino"]
        xv32 = ops.groupnorm_apply(xd, identity_stats(N), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f32w")
        pk32 = ops.pack_weight_f32_wino(w.to(dev))
        z0 = ops.conv3x3_bf16x3(xv32, pk32, bd, Cout)
        z1 = torch.empty_like(z0)
        rc = fn32(ctypes.c_void_p(xv32.buf.data_ptr()), ctypes.c_void_p(pk32.data_ptr()), ctypes.c_void_p(bd.data_ptr()), None, 0,
                  ctypes.c_void_p(z1.data_ptr()), N, H, W, Cin, Cout, None, st)
        assert rc == 0 and torch.equal(z0, z1), (N, H, W, Cin, Cout)


def test_wino_resampled_residual_in_epilogue(ops):
    """The identity skip of an ADM up / down block (unet_openai.py:238-243: x_upd = nearest 2x upsampling / 2x2 average pooling of x)
    read through the resampling in the conv's epilogue (res_resample = +1 / -1) must be bit-identical to dua_resample2x followed by
    the plain residual add, in both Winograd kernels, including a map smaller than one tile and a non-power-of-two height."""
    for (N, H, W, Cin, Cout) in [(4, 32, 32, 64, 128), (3, 8, 4, 32, 64), (2, 24, 16, 32, 96), (5, 16, 16, 64, 256), (300, 16, 16, 32, 160)]:
        x, w, b = rnd(60, N, Cin, H, W), rnd(61, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(62, Cout) * 0.1
        xd, bd = nhwc(x).to(dev), b.to(dev)
        for fmt, pack in (("f16x3w", ops.pack_weight_f16x3_wino), ("f32w", ops.pack_weight_f32_wino)):
            xv = ops.groupnorm_apply(xd, identity_stats(N), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split=fmt)
            packed = pack(w.to(dev))
            for mode, rshape in ((1, (N, H // 2, W // 2, Cout)), (-1, (N, 2 * H, 2 * W, Cout))):
                r = torch.randn(rshape, generator=torch.Generator().manual_seed(63 + mode)).to(dev)
                rr = ops.resample2x(r, mode)
                assert tuple(rr.shape) == (N, H, W, Cout)
                y0 = ops.conv3x3_bf16x3(xv, packed, bd, Cout, residual=rr)
                y1 = ops.conv3x3_bf16x3(xv, packed, bd, Cout, residual=r, res_resample=mode)
                assert torch.equal(y0, y1), (fmt, mode, N, H, W, Cin, Cout)
                ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + nchw(rr.cpu()).double()
                assert (nchw(y1.cpu()).double() - ref).abs().max().item() < 2e-5


def test_headline_job_variants_diverge_alike():
    """The bench's own workload at full size (BASELINE configs[1]: ADM-128, 50 DDIM steps, M=5, num_zigzag=3, UC window (40, 10),
    device noise, here on 2 images), free-running over the WHOLE job (100 denoiser forwards per image, the Monte-Carlo chains fed by
    the same Philox streams) in four arithmetic variants: exact fp32 MFMA, exact fp32 MFMA with the stand-alone GroupNorm statistics
    kernel (only the summation order of the statistics differs), f16x3 with direct 3x3 convs, f16x3 with the Winograd form (the
    headline).  A 50-step chain of a random-weight denoiser is chaotic at this length: ANY fp32-rounding-level difference grows by
    three to five orders of magnitude (measured over four seeds, scripts/diag/chain_divergence.py, DESIGN.md section 2: 7e-3 ... 0.44 on the score
    between the two EXACT-fp32 variants), so no per-pixel bar can hold between two fp32-equivalent implementations over the full job -
    the parity bars are per call, teacher-forced and on the 10-step chains (test_gpu_models.py).  What this test pins is that the
    Winograd form is not an outlier: its distance from the exact mode is of the order of the distances between the other pairs."""
    import argparse
    from oracle import adm
    from test_gpu_models import build_adm
    from diffusion_uncertainty_amd.generate_samples import generate_samples_model_scheduler_class_conditioned_from_tensor as gen
    from diffusion_uncertainty_amd.init_model import init_scheduler_imagenet128
    from diffusion_uncertainty_amd.schedulers_uncertainty.get_uncertainty_scheduler import get_uncertainty_scheduler
    from diffusion_uncertainty_amd.schedulers_uncertainty.noise import DeviceNoise
    m, _ = build_adm(dict(adm.ADM128), 128)
    g = torch.Generator().manual_seed(49395)
    X_T, y = torch.randn(2, 3, 128, 128, generator=g), torch.randint(0, 1000, (2,), generator=g)
    res = {}
    for name, prec, wino, fuse in (("fp32", "fp32", True, True), ("fp32b", "fp32", True, False), ("direct", "f16x3", False, True),
                                   ("wino", "f16x3", True, True)):
        m.precision, m.winograd, m.fuse_gn_stats = prec, wino, fuse
        args = argparse.Namespace(scheduler_type="uncertainty_zigzag_centered", M=5, start_step_uc=40, num_steps_uc=10, predict_next=True,
                                  eta=0.0, num_zigzag=3, uncertainty_distance=1)
        s = get_uncertainty_scheduler(args, None, m, init_scheduler_imagenet128())
        s.noise_source = DeviceNoise(seed=3)
        s.set_timesteps(50)
        res[name] = gen(X_T, y, 2, torch.device(dev), m, s)

    def dist(a, b):
        es = (res[a]["score"] - res[b]["score"]).abs().max().item()
        d = (res[a]["gen_images"].int() - res[b]["gen_images"].int()).abs()
        return es, d.max().item(), (d > 0).float().mean().item()

    pairs = {k: dist(*k.split("-")) for k in ("fp32b-fp32", "direct-fp32", "wino-fp32", "wino-direct")}
    for k, (es, dm, df) in pairs.items():
        print(f"headline job {k}: score {es:.2e}, image LSB diffs {df:.2e} (max {dm})")
    for r in res.values():
        assert torch.isfinite(r["score"]).all() and torch.isfinite(r["uncertainty"]).all() and (r["uncertainty"] >= 0).all()
    others = max(pairs["fp32b-fp32"][0], pairs["direct-fp32"][0])
    assert pairs["wino-fp32"][0] <= 10 * others + 1e-3, pairs


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 32, 32), (1, 16, 16, 64, 96), (1, 32, 32, 256, 256), (3, 64, 64, 64, 256),
                                             (2, 128, 128, 32, 128), (6, 8, 8, 96, 160), (130, 16, 16, 64, 64), (513, 16, 8, 32, 192)])
def test_conv3x3_wino_fp32_vs_fp64(ops, N, H, W, Cin, Cout):
    """The exact-fp32 Winograd path (fp32 V images -> fp32 packed weights -> k_wconv32 on v_mfma_f32_16x16x4_f32) against fp64, next to
    the direct fp32 MFMA conv on the same problem: within 2x of its error; V images equal the fp64 transform to fp32 rounding."""
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    w = w * torch.logspace(-3, 1, Cout).reshape(-1, 1, 1, 1)
    x = x * torch.logspace(-4, 1.5, Cin).reshape(1, -1, 1, 1)
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    scale_out = ref.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)
    xd = nhwc(x).to(dev)
    xv = ops.groupnorm_apply(xd, identity_stats(N), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f32w")
    assert xv.fmt == "f32w"
    v = xv.buf.cpu().double().permute(0, 1, 3, 2, 4).reshape(4, -1, Cin)[:, :N * (H // 2) * W].reshape(4, N, H // 2, W, Cin)
    vref = v_reference(nhwc(x).double())
    assert (v - vref).abs().max().item() <= 2.0 ** -23 * vref.abs().max().item()
    pk = ops.pack_weight_f32_wino(w.to(dev))
    y = ops.conv3x3_bf16x3(xv, pk, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    y32 = ops.conv3x3(xd, ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev))
    err = ((nchw(y.cpu()).double() - ref).abs() / scale_out).max().item()
    err32 = ((nchw(y32.cpu()).double() - ref).abs() / scale_out).max().item()
    print(f"winograd fp32 rel err {err:.2e}  direct fp32-MFMA {err32:.2e}")
    assert err < max(2 * err32, 2e-6), (err, err32)
    if (H * W) % 128 == 0 and Cout % 32 == 0:
        s1, s2 = ops.groupnorm_stats(y), ops.groupnorm_stats(y.clone())
        np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=2e-4, atol=2e-5)
    assert torch.equal(y, ops.conv3x3_bf16x3(xv, pk, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True))


def test_winograd_unet_fp32_vs_reference_and_direct(golden):
    """Full-size ADM-64 / ADM-128 in the exact fp32 mode with the 3x3 convs in the Winograd form (default) and in the direct form: both
    inside the 1e-3 bar against the reference's own output (G5), within 3e-5 of each other; block-level path == per-op composition."""
    from conftest import T
    from oracle import adm
    from test_gpu_models import build_adm
    for name, cfg in (("adm64", adm.ADM64), ("adm128", adm.ADM128)):
        g = golden(f"g5_{name}_full")
        m, _ = build_adm(dict(cfg), int(g["seed"]))
        assert m.precision == "fp32" and m.winograd
        x, t, y = T(g["x"]).to(dev), T(g["t"]).to(dev), T(g["y"]).to(dev)
        w = m(x, t, y=y)
        m.use_block_abi = False
        wop = m(x, t, y=y)
        m.use_block_abi = True
        m.winograd = False
        d = m(x, t, y=y)
        ew, ed = np.abs(w.cpu().numpy() - g["out"]).max(), np.abs(d.cpu().numpy() - g["out"]).max()
        print(f"{name} fp32: winograd err {ew:.3e}, direct err {ed:.3e}, |winograd - direct| {(w - d).abs().max().item():.3e}")
        assert ew < 1e-3 and ed < 1e-3 and ew < 2 * ed + 2e-6
        assert (w - d).abs().max().item() < 3e-5 and not torch.equal(w, d) and torch.equal(w, wop)
        del m
        torch.cuda.empty_cache()


@pytest.mark.parametrize("N,H,W,Cin,keep,cout", [(2, 16, 16, 64, 3, 6), (1, 8, 24, 32, 6, 6), (3, 32, 32, 256, 3, 6), (2, 5, 7, 32, 1, 4)])
def test_small_cout_conv_as_tap_gemm(ops, N, H, W, Cin, keep, cout):
    """The UNet's `out` conv (256 -> 6, 3 kept) as ONE one-tap GEMM over the pixels (nine tap columns per kept channel) plus the tap
    gather, in fp32 and in f16x3: against fp64 conv2d, held to the direct fp32 conv kernel's error on the same problem."""
    x, w, b = rnd(60, N, Cin, H, W), rnd(61, cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(62, cout) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)[:, :keep]
    xd = nhwc(x).to(dev)
    wt = ops.tap_column_weight(w.to(dev), keep)
    assert wt.shape == ((9 * keep + 31) // 32 * 32, Cin)
    bk = b[:keep].contiguous().to(dev)
    y32 = ops.conv3x3_tap_gather(ops.gemm(xd.view(-1, Cin), wt), bk, N, H, W, keep)
    xs = ops.groupnorm_apply(xd, identity_stats(N), torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev), act=0, split="f16x3")
    y16 = ops.conv3x3_tap_gather(ops.gemm_bf16x3_presplit(xs, ops.pack_weight_f16x3(wt), wt.shape[0]), bk, N, H, W, keep)
    yd = ops.conv3x3(xd, ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), keep, nchw_out=True)
    e32, e16, ed = [(t.cpu().double() - ref).abs().max().item() for t in (y32, y16, yd)]
    print(f"small-Cout conv: tap GEMM fp32 {e32:.2e}, f16x3 {e16:.2e}, direct fp32 conv {ed:.2e}")
    assert y32.shape == (N, keep, H, W) and e32 < max(2 * ed, 2e-6) and e16 < max(2 * ed, 2e-6)
