"""The one-tap f16x3 GEMM kernel (csrc/gemm_f16x3.hip: k_hgemm / k_hgemm_reduce) behind dua_gemm_f16x3_presplit*, through the C ABI.

Reference = fp64 product of the DECODED planes (hi + lo) / scale with the fp32 weights: what remains is the weight split (2^-22), the
dropped lo*lo term and fp32 accumulation, i.e. the error class of the exact fp32 MFMA engine.  Shapes: U-ViT-H/2's linears
(uvit/uvit.py:54-120: K = 1152 / 4608, N = 1152 / 3456 / 4608, rows not a multiple of the 256-row tile), ADM's qkv / proj_out / 1x1
skip shapes (unet_openai.py:219-223,287,295), ragged rows / columns, every tile width, with and without the stream-K tail."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = "cuda"


@pytest.fixture(scope="module")
def ops():
    from diffusion_uncertainty_amd import ops as o
    return o


@pytest.fixture(scope="module")
def L():
    from diffusion_uncertainty_amd import _lib
    return _lib.lib()


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def planes_of(ops, x, scale):
    """f16x3 planes of x * scale built in torch (saturating hi, fp16 remainder) + their exact fp64 decode."""
    s = (x.to(dev) * scale).clamp(-65504.0, 65504.0)
    hi = s.half()
    lo = (s - hi.float()).half()
    buf = torch.stack([hi, lo]).contiguous().view(torch.int16)
    dec = (hi.double() + lo.double()) / scale
    return ops.SplitPlanes(buf, tuple(x.shape), "f16x3", scale), dec.cpu()


def decode_planes(pl):
    b = pl.buf.view(torch.float16).double()
    return ((b[0] + b[1]) / pl.scale).cpu()


def gelu64(v):
    return 0.5 * v * (1.0 + torch.erf(v * 0.7071067811865476))


def check_colsums(c, M, N):
    ws = c._dua_colsums_rows.cpu().double()
    nb = (M + 63) // 64
    pad = torch.zeros(nb * 64, N, dtype=torch.float64)
    pad[:M] = c.cpu().double()
    blk = pad.reshape(nb, 64, N)
    w2 = ws.reshape(nb, 2, N)
    assert ((w2[:, 0] - blk.sum(1)).abs().max() / blk.abs().sum(1).max()).item() < 1e-5
    assert ((w2[:, 1] - (blk * blk).sum(1)).abs().max() / (blk * blk).sum(1).max()).item() < 1e-5


SHAPES = [(300, 96, 64), (1000, 192, 96), (257, 320, 224), (4096, 256, 256), (2048, 768, 512), (16512, 1152, 1152), (5000, 3456, 1152),
          (3000, 1152, 4608), (8192, 1024, 1024), (777, 132, 160), (130000, 512, 512)]


@pytest.mark.parametrize("nt", [0, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_presplit_gemm_vs_fp64(ops, L, M, N, K, nt):
    a = rnd(1, M, K) * torch.logspace(-2, 1, K)
    w = rnd(2, N, K) / math.sqrt(K) * torch.logspace(-1, 1, N).reshape(-1, 1)
    b, res = rnd(3, N) * 0.1, rnd(4, M, N)
    pl, dec = planes_of(ops, a, ops.F16X3_ACT_SCALE)
    ref = dec @ w.double().t() + b.double() + res.double()
    sc = ref.abs().amax(dim=0, keepdim=True)
    packed = ops.pack_weight_f16x3(w.to(dev))
    assert L.dua_set_hgemm_mode(nt, 1) == 0
    try:
        c = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
        c_again = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
        prev = ops.set_split_k(True, dev)
        try:
            ck = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
            ck2 = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
        finally:
            ops.set_split_k(prev, dev)
        L.dua_set_hgemm_mode(-1, 1)
        c_old = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev))
    finally:
        L.dua_set_hgemm_mode(0, 1)
    e = ((c.cpu().double() - ref).abs() / sc).max().item()
    ek = ((ck.cpu().double() - ref).abs() / sc).max().item()
    eo = ((c_old.cpu().double() - ref).abs() / sc).max().item()
    print(f"hgemm nt={nt} M={M} N={N} K={K}: rel err {e:.2e}, stream-K {ek:.2e}, previous engine {eo:.2e}")
    assert e < max(2 * eo, 2e-6), (e, eo)
    assert ek < max(2 * eo, 2e-6), (ek, eo)
    np.testing.assert_array_equal(c.cpu().numpy(), c_again.cpu().numpy())          # deterministic
    np.testing.assert_array_equal(ck.cpu().numpy(), ck2.cpu().numpy())
    check_colsums(c, M, N)
    check_colsums(ck, M, N)


@pytest.mark.parametrize("M,N,K", [(600, 256, 96), (16512, 4608, 1152), (1000, 1152, 64), (2500, 384, 1152)])
@pytest.mark.parametrize("act", [0, 2])
def test_presplit_gemm_planes_epilogue(ops, L, M, N, K, act):
    """fc1 -> GELU -> planes (uvit/timm.py:96-111): the planes decode to act(A W^T + b) within the split's precision."""
    a, w, b = rnd(5, M, K), rnd(6, N, K) / math.sqrt(K), rnd(7, N) * 0.5
    pl, dec = planes_of(ops, a, ops.F16X3_ACT_SCALE)
    ref = dec @ w.double().t() + b.double()
    if act == 2:
        ref = gelu64(ref)
    packed = ops.pack_weight_f16x3(w.to(dev))
    outs = []
    for sk in (False, True):
        prev = ops.set_split_k(sk, dev)
        try:
            o = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), act_out=act, split_out=True)
        finally:
            ops.set_split_k(prev, dev)
        assert o.fmt == "f16x3" and o.scale == ops.F16X3_RAW_SCALE
        e = (decode_planes(o) - ref).abs().max().item()
        print(f"planes epilogue act={act} stream-K={sk} M={M} N={N} K={K}: max err {e:.2e}")
        assert e < 4e-6 * max(1.0, ref.abs().max().item())
        outs.append(o)


def test_gelu_epilogue_tails(ops, L):
    """The GELU epilogue far outside the fitted range: a pre-activation of -1e6 or -1e30 gives ~0 (the tail term uses the
    clamped argument: it does not grow with |v|), +1e30 and +inf saturate the planes at their largest value instead of turning into
    inf - inf = NaN, moderate values keep the 4e-6 bar."""
    M, N, K = 600, 256, 96
    a, w = rnd(5, M, K), rnd(6, N, K) / math.sqrt(K)
    b = rnd(7, N) * 0.5
    b[0], b[1], b[2], b[3], b[4], b[5] = -1e6, -1e30, 1e30, float("inf"), 40.0, -40.0
    pl, dec = planes_of(ops, a, ops.F16X3_ACT_SCALE)
    o = ops.gemm_bf16x3_presplit(pl, ops.pack_weight_f16x3(w.to(dev)), N, bias=b.to(dev), act_out=2, split_out=True)
    got = decode_planes(o)
    assert torch.isfinite(got).all()                                               # no NaN, and the planes saturate instead of overflowing
    assert got[:, 0].abs().max().item() < 1e-6 and got[:, 1].abs().max().item() < 1e-6 and got[:, 5].abs().max().item() < 1e-6
    top = 65504.0 / o.scale
    assert (got[:, 2] > 0.99 * top).all() and (got[:, 3] > 0.99 * top).all()
    ref = gelu64(dec @ w.double().t() + b.double())
    assert (got[:, 4:] - ref[:, 4:]).abs().max().item() < 4e-6 * 41.0


@pytest.mark.parametrize("M,K1,K2,N", [(1000, 256, 256, 256), (16512, 1152, 1152, 1152), (4100, 512, 256, 512), (900, 64, 32, 96), (2048, 768, 0, 1024)])
def test_two_source_presplit_gemm(ops, L, M, K1, K2, N):
    """1x1 skip convolution / skip_linear on a virtual concatenation of two plane sets, with the f16x3 planes of the result beside it."""
    a1, a2 = rnd(8, M, K1) * 3, (rnd(9, M, K2) if K2 else None)
    w, b, res = rnd(10, N, K1 + K2) / math.sqrt(K1 + K2), rnd(11, N), rnd(12, M, N)
    p1, d1 = planes_of(ops, a1, ops.F16X3_RAW_SCALE)
    p2, d2 = planes_of(ops, a2, ops.F16X3_RAW_SCALE) if K2 else (None, None)
    dec = d1 if d2 is None else torch.cat([d1, d2], 1)
    ref = dec @ w.double().t() + b.double() + res.double()
    packed = ops.pack_weight_f16x3(w.to(dev))
    for sk in (False, True):
        prev = ops.set_split_k(sk, dev)
        try:
            c, xp = ops.gemm_f16x3_presplit2(p1, p2, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True, planes_out=True)
        finally:
            ops.set_split_k(prev, dev)
        e = (c.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        ex = (decode_planes(xp) - c.cpu().double()).abs().max().item()
        print(f"two-source M={M} K={K1}+{K2} N={N} stream-K={sk}: rel err {e:.2e}, planes vs fp32 result {ex:.2e}")
        assert e < 2e-6
        assert ex < 2.0 ** -21 * max(1.0, ref.abs().max().item())
        check_colsums(c, M, N)


@pytest.mark.parametrize("flags", [1 | 8, 1 | 16])
@pytest.mark.parametrize("M,N,K", [(16512, 1152, 1152), (5000, 3456, 1152), (2048, 768, 512), (40000, 256, 2304)])
def test_hgemm_ab_switches_select_correct_kernels(ops, L, M, N, K, flags):
    """dua_set_hgemm_mode's A/B switches - stream-K on the tail round only (8) / on the whole launch (16) - select between CORRECT kernels:
    same error class as the default, deterministic, column sums consistent."""
    a = rnd(21, M, K)
    w, b, res = rnd(22, N, K) / math.sqrt(K), rnd(23, N) * 0.1, rnd(24, M, N)
    pl, dec = planes_of(ops, a, ops.F16X3_ACT_SCALE)
    ref = dec @ w.double().t() + b.double() + res.double()
    packed = ops.pack_weight_f16x3(w.to(dev))
    prev = ops.set_split_k(True, dev)
    try:
        assert L.dua_set_hgemm_mode(0, flags) == 0
        c = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
        c2 = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
    finally:
        L.dua_set_hgemm_mode(0, 1)
        ops.set_split_k(prev, dev)
    e = (c.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"hgemm flags={flags} M={M} N={N} K={K}: rel err {e:.2e}")
    assert e < 2e-6
    np.testing.assert_array_equal(c.cpu().numpy(), c2.cpu().numpy())
    check_colsums(c, M, N)


def test_tile_width_rule_under_concurrent_streams_keeps_the_bits(ops, L):
    """dua_set_hgemm_concurrent(1) (set by _forward_streams while the batch runs in chunks on HIP streams) lets the one-tap GEMM choose its
    tile width by work instead of whole rounds of tiles: U-ViT's fc1 at 32 images (8256 rows, N = 4608: 594 tiles of 256 columns) moves from
    128-column to 256-column tiles.  The result - fp32 output and GELU -> planes - must be the same bits either way, and the setter returns
    the previous value."""
    M, K, N = 32 * 258, 1152, 4608
    x, w, b = rnd(31, M, K), rnd(32, N, K) / math.sqrt(K), rnd(33, N)
    pl, _ = planes_of(ops, x, ops.F16X3_ACT_SCALE)
    pk = ops.pack_weight_f16x3(w.to(dev))
    out = {}
    assert L.dua_set_hgemm_concurrent(0) == 0
    try:
        for conc in (0, 1):
            was = L.dua_set_hgemm_concurrent(conc)
            assert was == (0 if conc == 0 else 0)
            c = ops.gemm_bf16x3_presplit(pl, pk, N, bias=b.to(dev))
            g = ops.gemm_bf16x3_presplit(pl, pk, N, bias=b.to(dev), act_out=2, split_out=True)
            out[conc] = (c.clone(), g.buf.clone())
        assert L.dua_set_hgemm_concurrent(0) == 1
    finally:
        L.dua_set_hgemm_concurrent(0)
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
