"""Op-level parity of the HIP kernels (through the C ABI) against the CPU oracle / torch fp32 references."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import T

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


def test_ddim_step_vs_golden(ops, golden):
    from oracle import ddim, schedule
    g = golden("g2_ddim_step")
    eps, x = T(g["eps"]), T(g["x"])
    tabs = {0: schedule.Tables(), 1: schedule.Tables(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")}
    for c in range(int(g["n_cases"])):
        which, t, eta = g[f"c{c}_meta"]
        tab = tabs[int(which)]
        co = ddim.step_coefficients(tab, int(t), 50, float(eta))
        noise = None
        if eta > 0:
            torch.manual_seed(100 + c)
            noise = torch.randn_like(eps).to(dev)
        prev, x0 = ops.ddim_step(eps.to(dev), x.to(dev), co["sqrt_b_t"], co["sqrt_a_t"], co["sqrt_a_prev"], co["dir_coef"],
                                 sigma=float(co["sigma"]), noise=noise, clip=(int(which) == 0))
        # the schedule tables are rebuilt on this host: torch's linspace/cumprod differ by an ulp between CPU
        # generations, and x0 = (x - sb*eps)/sa amplifies that at high-noise steps, hence the scaled atol.
        a = 2e-6 * float(np.abs(g[f"c{c}_x0"]).max())
        np.testing.assert_allclose(x0.cpu().numpy(), g[f"c{c}_x0"], rtol=1e-6, atol=a)
        np.testing.assert_allclose(prev.cpu().numpy(), g[f"c{c}_prev"], rtol=1e-6, atol=a)


@pytest.mark.parametrize("n", [0, 1, 5, 1024, 3 * 128 * 128 * 2 + 3])
def test_elementwise_sizes(ops, n):
    a, b = rnd(1, n), rnd(2, n)
    out = ops.axpby(a.to(dev), b.to(dev), 0.75, -1.25)
    np.testing.assert_array_equal(out.cpu().numpy(), (0.75 * a + (-1.25) * b).numpy())
    if n:
        u8 = ops.to_uint8((a * 1.2).to(dev))
        ref = (((a * 1.2) / 2 + 0.5).clamp(0, 1) * 255.0).round().to(torch.uint8)
        np.testing.assert_array_equal(u8.cpu().numpy(), ref.numpy())
    # DPM-Solver++ kernels: same operation order as the reference's separate torch ops -> bit-exact
    c = rnd(3, n)
    st, at, c1, c2, rinv = (torch.tensor(v, dtype=torch.float32) for v in (0.83, 0.55, 0.91, -0.37, 1.7))
    x0 = ops.x0_from_eps(a.to(dev), b.to(dev), st, at)
    np.testing.assert_array_equal(x0.cpu().numpy(), ((b - st * a) / at).numpy())
    up = ops.dpmpp_2m_update(a.to(dev), b.to(dev), c.to(dev), c1, c2, 0.5 * c2, rinv)
    np.testing.assert_array_equal(up.cpu().numpy(), (c1 * a - c2 * b - 0.5 * c2 * (rinv * (b - c))).numpy())


@pytest.mark.parametrize("M", [2, 5, 10])
def test_mc_reductions(ops, M):
    st, e = rnd(3, M, 2, 3, 16, 16), rnd(4, 2, 3, 16, 16)
    u = ops.mc_reduce_centered(st.to(dev), e.to(dev))
    ref = (st - e[None]).pow(2).mean(0)
    np.testing.assert_allclose(u.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
    acc = torch.empty_like(e, device=dev)
    for m in range(M):
        ops.mc_centered_accum(st[m].to(dev), e.to(dev), acc, m, M)
    np.testing.assert_array_equal(acc.cpu().numpy(), u.cpu().numpy())
    v = ops.mc_reduce_var(st.to(dev))
    np.testing.assert_allclose(v.cpu().numpy(), torch.var(st, dim=0).numpy(), rtol=1e-5, atol=1e-7)


def test_quantile_mask_bit_exact_golden(ops, golden):
    g = golden("g7_guidance")
    for name in ("rand", "ties", "const", "adm128_like", "odd_n"):
        u = T(g[f"u_{name}"]).to(dev)
        for p in g["ps"]:
            mask, q = ops.quantile_mask(u, float(p))
            np.testing.assert_array_equal(q.cpu().numpy(), g[f"q_{name}_{p}"])
            np.testing.assert_array_equal(mask.cpu().numpy(), g[f"mask_{name}_{p}"])


def test_quantile_mask_full_size_properties(ops):
    """BASELINE size: [128, 3, 128, 128]; properties: exactly n - (hi+1) .. n - lo - 1 elements above q,
    agreement with the oracle's sort-based quantile."""
    from oracle import guidance
    u = (rnd(5, 128, 3, 128, 128) ** 2)
    mask, q = ops.quantile_mask(u.to(dev), 0.9)
    mref, qref = guidance.quantile_mask(u.numpy(), 0.9)
    np.testing.assert_array_equal(q.cpu().numpy(), qref)
    np.testing.assert_array_equal(mask.cpu().numpy(), mref)
    # negative values and -0.0 / +0.0 order correctly
    v = rnd(6, 4, 1000)
    v[:, :10] = 0.0
    v[:, 10:20] = -0.0
    for p in (0.0, 0.3, 0.5, 1.0):
        mask, q = ops.quantile_mask(v.to(dev), p)
        mref, qref = guidance.quantile_mask(v.numpy(), p)
        np.testing.assert_array_equal(q.cpu().numpy(), qref)
        np.testing.assert_array_equal(mask.cpu().numpy(), mref)


@pytest.mark.parametrize("shape", [(1, 4, 64, 64), (128, 3, 128, 128), (3, 3, 64, 64), (2, 4, 32, 32), (5, 1, 1, 4099), (2, 1, 1, 65536),
                                   (2, 1, 1, 65537), (3, 1, 1, 7), (4, 1, 1, 40000), (3, 1, 1, 20000), (2, 2, 128, 128), (5, 1, 1, 16388),
                                   (300, 3, 128, 128)])         # (600 workgroups of 1024 threads: more than the 256 CUs hold at once)
def test_quantile_mask_register_kernel_vs_streaming(ops, shape):
    """The register-resident kernels (sample read once: one workgroup per sample with an LDS histogram over the sample's own key range; two
    workgroups per sample with a sample bracket for 16384 < n <= 65536) against the streaming six-pass kernel and the oracle's sort-based
    quantile, on smooth, tied, constant, signed, huge-range, clustered and correlated inputs: q and mask bit for bit.  Prints the time per call."""
    from oracle import guidance
    from diffusion_uncertainty_amd import _lib
    L = _lib.lib()
    B = shape[0]
    n = int(np.prod(shape[1:]))
    g = torch.Generator().manual_seed(77)
    base = torch.randn(shape, generator=g)
    cases = {"chi2": base ** 2, "signed": base, "ties": (base * 3).round() / 3, "const": torch.full(shape, 0.25),
             "range": base * torch.logspace(-20, 20, n).reshape(shape[1:]), "cluster": 1.0 + base * 1e-6,
             "mostly_zero": torch.where(torch.rand(shape, generator=g) < 0.7, torch.zeros(shape), base ** 2),
             # a random walk along the flattened sample: neighbouring values strongly correlated (the several-workgroups kernel
             # brackets the target from 768 scattered float4 of the sample)
             "walk": base.reshape(B, -1).cumsum(1).reshape(shape)}
    for name, u in cases.items():
        if B * n > 2_000_000 and name not in ("chi2", "ties", "mostly_zero", "walk"):
            continue
        ud = u.contiguous().to(dev)
        for p in (0.9, 0.0, 1.0, 0.5) if B * n < 2_000_000 else (0.9,):
            mask, q = ops.quantile_mask(ud, p)
            L.dua_set_quantile_stream(1)
            try:
                mask_s, q_s = ops.quantile_mask(ud, p)
            finally:
                L.dua_set_quantile_stream(0)
            np.testing.assert_array_equal(q.cpu().numpy().view(np.uint32), q_s.cpu().numpy().view(np.uint32), err_msg=f"{name} p={p}")
            assert torch.equal(mask, mask_s), (name, p)
            if 16384 < n <= 65536 and n % 4 == 0:
                # the default above was the several-workgroups-per-sample kernel: also the one-workgroup register kernel (2) and the
                # several-workgroups kernel with its sample bracket declared missed (3: workgroup 0 selects over global memory), twice
                # in a row (each launch has to leave the workspace counters zeroed), and with every workgroup acting as if its partner had
                # never arrived (4: the way out of the bounded wait - each selects alone)
                for mode in (2, 3, 3, 4, 0):
                    L.dua_set_quantile_stream(mode)
                    try:
                        mask_m, q_m = ops.quantile_mask(ud, p)
                    finally:
                        L.dua_set_quantile_stream(0)
                    np.testing.assert_array_equal(q_m.cpu().numpy().view(np.uint32), q_s.cpu().numpy().view(np.uint32), err_msg=f"{name} p={p} mode={mode}")
                    assert torch.equal(mask_m, mask_s), (name, p, mode)
            if B * n <= 200_000:
                mref, qref = guidance.quantile_mask(u.reshape(B, -1).numpy(), p)
                np.testing.assert_array_equal(q.cpu().numpy(), qref, err_msg=f"{name} p={p}")
                np.testing.assert_array_equal(mask.cpu().numpy().reshape(B, -1), mref, err_msg=f"{name} p={p}")
    # kernel time: direct C-ABI calls on preallocated outputs (the Python wrapper's two allocations per call would otherwise be
    # what is measured once the kernel takes a few tens of microseconds)
    import ctypes
    lo, hi, wq = ops.quantile_rank_split(0.9, n)
    mk, qq = torch.empty(shape, device=dev, dtype=torch.uint8), torch.empty(B, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    alg = 2.25 * 4 * B * n                                   # >= 2P (read u twice) + P/4 (mask)
    report = {}
    for name in ("chi2", "signed", "const"):
        ud = cases[name].contiguous().to(dev)
        args = (ctypes.c_void_p(ud.data_ptr()), B, n, lo, hi, wq, ctypes.c_void_p(mk.data_ptr()), ctypes.c_void_p(qq.data_ptr()), st)
        times = {}
        for mode in (0, 1, 2):
            L.dua_set_quantile_stream(mode)
            try:
                for _ in range(5):
                    assert L.dua_quantile_mask(*args) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    L.dua_quantile_mask(*args)
                e1.record()
                torch.cuda.synchronize()
                times[mode] = e0.elapsed_time(e1) / 50 * 1e3
            finally:
                L.dua_set_quantile_stream(0)
        report[name] = times
        print(f"quantile_mask {shape} {name}: default dispatch {times[0]:.1f} us ({alg / times[0] / 1e6:.2f} TB/s algorithmic), "
              f"one-workgroup register kernel {times[2]:.1f} us ({alg / times[2] / 1e6:.2f} TB/s), "
              f"streaming kernel {times[1]:.1f} us ({alg / times[1] / 1e6:.2f} TB/s)")
    if shape == (128, 3, 128, 128):
        # regression guard.  One workgroup per sample: 1.17 TB/s (r01: 0.53; its floor, constant input, is 3.0).  Round 5, two workgroups per
        # sample bracketing the target from a common 3072-key sample (k_quantile_mask_multi): 2.9 TB/s measured (19.7 us, of
        # which ~3 us are the launch; scripts/diag/quantile_diag.py has the phases); VERDICT r04 item 9 asked for 3
        assert alg / report["chi2"][0] / 1e6 >= 2.0
        assert alg / report["chi2"][2] / 1e6 >= 0.9
    if shape == (1, 4, 64, 64):
        assert report["chi2"][0] < 20.0


def test_posterior_blend(ops):
    from oracle import guidance
    eps, u = rnd(7, 1, 4, 64, 64), rnd(8, 1, 4, 64, 64) ** 2 + 1e-3
    mask, _ = guidance.quantile_mask(u.numpy(), 0.9)
    out = ops.posterior_blend(eps.to(dev), u.to(dev), T(mask).to(dev), 5, 0.3)
    ref = guidance.posterior_blend(eps.numpy(), u.numpy(), mask, 5, 0.3)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-6, atol=1e-7)


def test_randn_statistics(ops):
    z = ops.randn((4, 3, 128, 128), dev, seed=1, offset=0).cpu()
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    assert torch.isfinite(z).all()
    z2 = ops.randn((4, 3, 128, 128), dev, seed=1, offset=0).cpu()
    assert torch.equal(z, z2)
    z3 = ops.randn((4, 3, 128, 128), dev, seed=1, offset=z.numel() // 4).cpu()
    assert not torch.equal(z, z3)
    k = float(((z ** 4).mean() / (z ** 2).mean() ** 2))
    assert abs(k - 3.0) < 0.1
    m = ops.dropout_mask((1 << 20,), dev, 0.1, seed=3, offset=0).cpu()
    assert set(np.unique(m.numpy()).tolist()) <= {0.0, np.float32(1 / 0.9).item()}
    assert abs((m > 0).float().mean().item() - 0.9) < 3e-3


def test_layout_roundtrip(ops):
    x = rnd(9, 3, 5, 6, 7)
    y = ops.nchw_to_nhwc(x.to(dev), 8)
    assert torch.equal(y[..., :5].cpu(), nhwc(x)) and (y[..., 5:] == 0).all()
    z = ops.nhwc_to_nchw(y, 3)
    assert torch.equal(z.cpu(), x[:, :3])


@pytest.mark.parametrize("C1,C2,H", [(32, 0, 8), (64, 32, 8), (192, 0, 16), (256, 0, 32), (576, 384, 4), (1024, 1024, 8), (1280, 0, 4),
                                     (64, 0, 64), (32, 32, 64), (64, 0, 128)])       # >= 64^2: the 8-channel-per-lane kernel
def test_groupnorm(ops, C1, C2, H):
    N = 2
    x1 = rnd(10, N, C1, H, H) * 2 + 0.5
    x2 = rnd(11, N, C2, H, H) - 1 if C2 else None
    xc = torch.cat([x1, x2], 1) if C2 else x1
    C = C1 + C2
    ga, be = 1 + 0.1 * rnd(12, C), 0.1 * rnd(13, C)
    film = rnd(14, N, 2 * C) * 0.5
    d1, d2 = nhwc(x1).to(dev), (nhwc(x2).to(dev) if C2 else None)
    stats = ops.groupnorm_stats(d1, d2)
    xr = xc.reshape(N, 32, -1).double()
    np.testing.assert_allclose(stats[..., 0].cpu().numpy(), xr.mean(-1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(stats[..., 1].cpu().numpy(), (xr.var(-1, unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-5)
    gn = F.group_norm(xc, 32, ga, be)
    # plain GN+SiLU
    y = ops.groupnorm_apply(d1, stats, ga.to(dev), be.to(dev), x2=d2, act=1)
    np.testing.assert_allclose(nchw(y.cpu()).numpy(), F.silu(gn).numpy(), rtol=1e-4, atol=1e-5)
    # FiLM + SiLU
    sc, sh = film[:, :C, None, None], film[:, C:, None, None]
    y = ops.groupnorm_apply(d1, stats, ga.to(dev), be.to(dev), x2=d2, film=film.to(dev), act=1)
    np.testing.assert_allclose(nchw(y.cpu()).numpy(), F.silu(gn * (1 + sc) + sh).numpy(), rtol=1e-4, atol=1e-5)
    # no activation (attention norm)
    y = ops.groupnorm_apply(d1, stats, ga.to(dev), be.to(dev), x2=d2, act=0)
    np.testing.assert_allclose(nchw(y.cpu()).numpy(), gn.numpy(), rtol=1e-4, atol=1e-5)
    if C2 == 0:
        y = ops.groupnorm_apply(d1, stats, ga.to(dev), be.to(dev), act=1, resample=-1)
        np.testing.assert_allclose(nchw(y.cpu()).numpy(), F.avg_pool2d(F.silu(gn), 2, 2).numpy(), rtol=1e-4, atol=1e-5)
        y = ops.groupnorm_apply(d1, stats, ga.to(dev), be.to(dev), act=1, resample=1)
        np.testing.assert_allclose(nchw(y.cpu()).numpy(), F.interpolate(F.silu(gn), scale_factor=2, mode="nearest").numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_array_equal(nchw(ops.resample2x(d1, -1).cpu()).numpy(), F.avg_pool2d(x1, 2, 2).numpy())
        np.testing.assert_array_equal(nchw(ops.resample2x(d1, 1).cpu()).numpy(), F.interpolate(x1, scale_factor=2, mode="nearest").numpy())
    if C % 32 == 0:
        # split outputs decode to the fp32 result: bf16 hi + lo (2^-17)
        want = nhwc(F.silu(gn * (1 + sc) + sh))
        for split, tol in ((True, 2e-5),):
            for res, ref in ((0, want), (1, nhwc(F.interpolate(nchw(want), scale_factor=2, mode="nearest")))) if C2 == 0 else ((0, want),):
                sp = ops.groupnorm_apply(d1, stats, ga.to(dev), be.to(dev), x2=d2, film=film.to(dev), act=1, split=split, resample=res)
                buf = sp.buf.view(torch.uint8).reshape(2, -1).cpu()
                dec = buf[0].view(torch.bfloat16).float() + buf[1].view(torch.bfloat16).float()
                assert (dec.reshape(ref.shape) - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 8, 8, 32, 32), (1, 16, 16, 64, 96), (3, 4, 4, 96, 64), (2, 16, 8, 3, 32),
                                             (2, 16, 16, 32, 6), (1, 32, 32, 256, 256), (2, 8, 8, 1024, 192)])
def test_conv3x3(ops, N, H, W, Cin, Cout):
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    wp = ops.pack_conv3x3_weight(w.to(dev))
    xd = ops.nchw_to_nhwc(x.to(dev), 4) if Cin <= 4 else nhwc(x).to(dev)
    y = ops.conv3x3(xd, wp, b.to(dev), Cout)
    np.testing.assert_allclose(nchw(y.cpu()).numpy(), ref.float().numpy(), rtol=1e-4, atol=2e-5)
    y = ops.conv3x3(xd, wp, b.to(dev), Cout, residual=nhwc(res).to(dev))
    np.testing.assert_allclose(nchw(y.cpu()).numpy(), (ref + res).float().numpy(), rtol=1e-4, atol=2e-5)
    y = ops.conv3x3(xd, wp, b.to(dev), Cout, nchw_out=True)
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("M,N,K1,K2", [(128, 128, 32, 0), (200, 72, 64, 0), (5, 2048, 256, 0), (300, 96, 64, 32), (64, 16, 1152, 0),
                                       (258, 1152, 16, 0), (130, 260, 100, 0)])
def test_gemm(ops, M, N, K1, K2):
    a1, a2 = rnd(30, M, K1), (rnd(31, M, K2) if K2 else None)
    w, b, r = rnd(32, N, K1 + K2) / math.sqrt(K1 + K2), rnd(33, N), rnd(34, M, N)
    a = torch.cat([a1, a2], 1) if K2 else a1
    ref = a.double() @ w.double().T
    c = ops.gemm(a1.to(dev), w.to(dev), a2=(a2.to(dev) if K2 else None))
    np.testing.assert_allclose(c.cpu().numpy(), ref.float().numpy(), rtol=1e-4, atol=2e-5)
    c = ops.gemm(a1.to(dev), w.to(dev), bias=b.to(dev), a2=(a2.to(dev) if K2 else None), residual=r.to(dev), alpha=0.5)
    np.testing.assert_allclose(c.cpu().numpy(), (0.5 * ref + b + r).float().numpy(), rtol=1e-4, atol=2e-5)
    c = ops.gemm(a1.to(dev), w.to(dev), bias=b.to(dev), a2=(a2.to(dev) if K2 else None), act_in=1)
    np.testing.assert_allclose(c.cpu().numpy(), (F.silu(a.double()) @ w.double().T + b).float().numpy(), rtol=1e-4, atol=2e-5)
    c = ops.gemm(a1.to(dev), w.to(dev), bias=b.to(dev), a2=(a2.to(dev) if K2 else None), act_out=2)
    np.testing.assert_allclose(c.cpu().numpy(), F.gelu(ref + b).float().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("B,T_,heads,ch", [(2, 64, 4, 32), (1, 16, 2, 64), (2, 256, 4, 96), (1, 1024, 4, 128), (3, 64, 4, 256), (2, 258, 16, 72), (1, 1, 2, 8), (2, 77, 3, 40), (1, 600, 2, 192)])
def test_qkv_attention(ops, B, T_, heads, ch):
    from oracle import adm
    C = heads * ch
    qkv = rnd(40, B, 3 * C, T_)                       # reference layout [B, 3C, T], NEW order (q|k|v thirds)
    ref = adm.qkv_attention(qkv, heads, True)         # [B, C, T]
    d = qkv.permute(0, 2, 1).reshape(B * T_, 3 * C).contiguous().to(dev)
    import diffusion_uncertainty_amd.ops as O
    try:
        for mode in ("gemm", "fused", "fused_x", "auto"):   # materialised-logits GEMM pair, fp32 flash kernel, split-bf16 flash kernel, default
            O.ATTENTION_MODE = mode
            out = ops.qkv_attention(d, B, T_, heads, ch)
            np.testing.assert_allclose(out.cpu().reshape(B, T_, C).permute(0, 2, 1).numpy(), ref.numpy(), rtol=1e-4, atol=2e-5, err_msg=mode)
            out = ops.qkv_attention(d, B, T_, heads, ch, precision="bf16x3")     # split-bf16 products only when ch, T % 32 == 0
            np.testing.assert_allclose(out.cpu().reshape(B, T_, C).permute(0, 2, 1).numpy(), ref.numpy(), rtol=1e-3, atol=1e-4, err_msg=mode)
    finally:
        O.ATTENTION_MODE = "auto"


@pytest.mark.parametrize("B,T_,heads,ch", [(2, 64, 4, 32), (1, 16, 2, 64), (2, 256, 4, 96), (1, 1024, 4, 128), (2, 258, 16, 72), (1, 1, 2, 8),
                                            (2, 77, 3, 40), (3, 64, 4, 256), (2, 256, 4, 192), (1, 100, 2, 160), (1, 300, 3, 224),
                                            (1, 130, 2, 132)])
def test_qkv_attention_f16x3(ops, B, T_, heads, ch):
    """The split flash kernel on fp16 hi / lo halves (f16x3): held to the error of the exact fp32 flash kernel against fp64
    (within 2x), on O(1) and on large-magnitude q / k / v; the planes output decodes to the fp32 output."""
    C = heads * ch
    for mag in (1.0, 8.0):
        qkv = (rnd(45, B * T_, 3 * C) * mag).to(dev)
        q, k, v = [t.reshape(B, T_, heads, ch).permute(0, 2, 1, 3).double().cpu() for t in qkv.split(C, dim=1)]
        ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(ch), -1) @ v).permute(0, 2, 1, 3).reshape(B * T_, C)
        out = ops.qkv_attention(qkv, B, T_, heads, ch, precision="f16x3")
        o32 = ops.qkv_attention(qkv, B, T_, heads, ch, precision="fp32")
        e, e32 = (out.cpu().double() - ref).abs().max().item(), (o32.cpu().double() - ref).abs().max().item()
        print(f"attention f16x3 mag {mag}: err {e:.2e}  fp32 err {e32:.2e}")
        assert e < max(2 * e32, 3e-6 * mag), (e, e32)
        if C % 32 == 0:
            pl = ops.qkv_attention(qkv, B, T_, heads, ch, precision="f16x3", planes_out=True)
            assert pl.fmt == "f16x3"
            h = pl.buf.view(torch.float16).reshape(2, B * T_, C).float().cpu()
            np.testing.assert_allclose(((h[0] + h[1]) / pl.scale).numpy(), out.cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_softmax_layernorm_embeddings(ops):
    from oracle import adm
    for cols in (1, 64, 100, 258, 1024, 2048):
        s = rnd(50, 37, cols) * 3
        d = s.to(dev).clone()
        ops.softmax_rows_(d, 37, cols)
        np.testing.assert_allclose(d.cpu().numpy(), torch.softmax(s, -1).numpy(), rtol=1e-5, atol=1e-7)
    for D in (64, 1152):
        x, ga, be = rnd(51, 2, 18, D) * 2 + 1, rnd(52, D), rnd(53, D)
        y = ops.layernorm(x.to(dev), ga.to(dev), be.to(dev))
        np.testing.assert_allclose(y.cpu().numpy(), F.layer_norm(x, (D,), ga, be).numpy(), rtol=1e-4, atol=1e-5)
    t = torch.tensor([0, 1, 500, 999])
    for dim in (256, 192, 193):
        e = ops.timestep_embedding(t.to(dev), dim)
        np.testing.assert_allclose(e.cpu().numpy(), adm.timestep_embedding(t, dim).numpy(), rtol=0, atol=2e-4)
    tab, idx, a = rnd(54, 10, 32), torch.tensor([3, 0, 9]), rnd(55, 3, 32)
    o = ops.embedding_add(a.to(dev), tab.to(dev), idx.to(dev))
    np.testing.assert_array_equal(o.cpu().numpy(), (a + tab[idx]).numpy())


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 8, 8, 32, 32), (1, 16, 16, 64, 96), (3, 4, 4, 96, 64), (2, 16, 16, 32, 6),
                                             (1, 32, 32, 256, 256), (2, 8, 8, 1024, 192)])
def test_conv3x3_bf16x3(ops, N, H, W, Cin, Cout):
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    planes = ops.pack_weight_bf16x3(w.to(dev))
    y = ops.conv3x3_bf16x3(nhwc(x).to(dev), planes, b.to(dev), Cout, residual=nhwc(res).to(dev))
    err = (nchw(y.cpu()).double() - (ref + res)).abs().max().item()
    assert err < 1e-4, err                                   # split-bf16: ~1e-5 on O(1) outputs
    y = ops.conv3x3_bf16x3(nhwc(x).to(dev), planes, b.to(dev), Cout, nchw_out=True)
    assert (y.cpu().double() - ref).abs().max().item() < 1e-4


def _f16x3_planes(ops, x_nhwc):
    """x in the f16x3 plane format, written by the product kernel itself (identity GroupNorm: mean 0, rstd 1, gamma 1)."""
    N, C = x_nhwc.shape[0], x_nhwc.shape[3]
    st = torch.tensor([0.0, 1.0]).repeat(N * 32, 1).contiguous().to(dev)
    return ops.groupnorm_apply(x_nhwc.to(dev), st, torch.ones(C, device=dev), torch.zeros(C, device=dev), act=0, split="f16x3")


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 8, 8, 32, 32), (1, 16, 16, 64, 96), (3, 4, 4, 96, 64), (2, 16, 16, 32, 6),
                                             (1, 32, 32, 256, 256), (2, 8, 8, 1024, 192), (5, 16, 16, 128, 512),
                                             (1, 24, 24, 64, 64), (8, 128, 128, 32, 256), (3, 64, 64, 64, 256)])
def test_conv3x3_f16x3(ops, N, H, W, Cin, Cout):
    """The fp32-equivalent split (fp16 hi + fp16 lo of the scaled operand, three fp16 MFMAs): plane format, packing and kernel
    against fp64, held to the error of the exact fp32 MFMA engine on the same problem (within 2x)."""
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    # rows of very different magnitude (per-row weight exponents) and activations over a wide range incl. tiny values
    w = w * torch.logspace(-3, 1, Cout).reshape(-1, 1, 1, 1)
    x = x * torch.logspace(-4, 1.5, Cin).reshape(1, -1, 1, 1)
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    scale_out = ref.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)          # per output channel magnitude
    xs = _f16x3_planes(ops, nhwc(x))
    assert xs.fmt == "f16x3" and xs.scale == ops.F16X3_ACT_SCALE
    packed = ops.pack_weight_f16x3(w.to(dev))
    y = ops.conv3x3_bf16x3(xs, packed, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=(Cout % 4 == 0))
    y32 = ops.conv3x3(nhwc(x).to(dev), ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, residual=nhwc(res).to(dev))
    err = ((nchw(y.cpu()).double() - (ref + res)).abs() / scale_out).max().item()
    err32 = ((nchw(y32.cpu()).double() - (ref + res)).abs() / scale_out).max().item()
    print(f"f16x3 rel err {err:.2e}  fp32-MFMA rel err {err32:.2e}")
    assert err < max(2 * err32, 2e-6), (err, err32)
    if N * H * W <= 70000:
        buf = xs.buf.view(torch.uint8).reshape(2, -1).cpu()
        hi = buf[0].view(torch.float16).reshape(N, H, W, Cin).double()
        lo = buf[1].view(torch.float16).reshape(N, H, W, Cin).double()
        s = nhwc(x).double() * xs.scale
        np.testing.assert_array_equal(hi.numpy(), (nhwc(x) * xs.scale).half().double().numpy())
        big = s.abs() >= 2.0 ** -3                                   # lo is a normal fp16 there: 22 significant bits
        assert ((hi + lo - s).abs()[big] / s.abs()[big]).max().item() < 2.0 ** -21
        assert (hi + lo - s).abs()[~big].max().item() <= 2.0 ** -25 if (~big).any() else True
        y = ops.conv3x3_bf16x3(xs, packed, b.to(dev), Cout, nchw_out=True)
        assert ((y.cpu().double() - ref).abs() / scale_out).max().item() < max(2 * err32, 2e-6)
    if Cout % 4:
        return
    w1 = rnd(24, Cout, Cin) / math.sqrt(Cin)
    c = ops.gemm_bf16x3_presplit(xs, ops.pack_weight_f16x3(w1.to(dev)), Cout, bias=b.to(dev))
    ref1 = nhwc(x).reshape(-1, Cin).double() @ w1.double().t() + b.double()
    c32 = ops.gemm(nhwc(x).reshape(-1, Cin).to(dev), w1.to(dev), bias=b.to(dev))
    e1, e32 = (c.cpu().double() - ref1).abs().max().item(), (c32.cpu().double() - ref1).abs().max().item()
    assert e1 < max(2 * e32, 2e-6 * ref1.abs().max().item()), (e1, e32)


@pytest.mark.parametrize("B,Tq,Tk,heads,ch", [(2, 256, 77, 8, 40), (1, 64, 7, 4, 8), (3, 1024, 77, 8, 80), (2, 100, 33, 2, 128),
                                              (2, 100, 33, 2, 192), (1, 200, 77, 1, 256)])
def test_cross_attention_f16x3(ops, B, Tq, Tk, heads, ch):
    """The split flash kernel with keys / values from their own tensors (cross-attention: Tk != Tq, e.g. 77 text tokens): against fp64,
    held to the error of the fp32 GEMM path; planes output decodes to the same values."""
    C = heads * ch
    q, k, v = rnd(80, B, Tq, C), rnd(81, B, Tk, C), rnd(82, B, Tk, C)
    sp = lambda z: z.double().reshape(B, z.shape[1], heads, ch).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, Tq, C)
    o32 = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads).cpu()
    o = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, precision="f16x3").cpu()
    e32, e = (o32.double() - ref).abs().max().item(), (o.double() - ref).abs().max().item()
    print(f"cross-attention f16x3 err {e:.2e}  fp32 GEMM path {e32:.2e}")
    assert e < max(3 * e32, 3e-6)
    if C % 32 == 0:
        pl = ops.attention(q.to(dev), k.to(dev), v.to(dev), heads, precision="f16x3", planes_out=True)
        buf = pl.buf.view(torch.uint8).reshape(2, -1).cpu()
        dec = (buf[0].view(torch.float16).double() + buf[1].view(torch.float16).double()).reshape(B, Tq, C) / pl.scale
        assert (dec - o.double()).abs().max().item() < 1e-6


@pytest.mark.parametrize("M,N,K", [(16512, 1152, 1152), (16512, 1152, 4608), (20000, 640, 320)])
def test_gemm_f16x3_tail_split(ops, M, N, K):
    """Head | tail split of the rows of a presplit GEMM whose last round of tiles would be mostly idle (U-ViT-H/2 at batch 64:
    585 tiles on 256 CUs): the tail runs split along K.  Same error against fp64 as the single launch; statistics consistent;
    deterministic."""
    a = rnd(90, M, K) * torch.logspace(-2, 1, K)
    w, b, res = rnd(91, N, K) / math.sqrt(K), rnd(92, N) * 0.1, rnd(93, M, N)
    ref = a.double() @ w.double().t() + b.double() + res.double()
    r = 1 if K <= 2048 else 4                                     # (the plane writer takes up to 3072 channels: fold wide rows)
    pl = ops.SplitPlanes(_f16x3_planes(ops, a.view(1, M * r, 1, K // r)).buf, (M, K), "f16x3", ops.F16X3_ACT_SCALE)
    packed = ops.pack_weight_f16x3(w.to(dev))
    single = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
    prev = ops.set_split_k(True, dev)
    try:
        c = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
        c2 = ops.gemm_bf16x3_presplit(pl, packed, N, bias=b.to(dev), residual=res.to(dev), colsum=True)
    finally:
        ops.set_split_k(prev, dev)
    e, e1 = (c.cpu().double() - ref).abs().max().item(), (single.cpu().double() - ref).abs().max().item()
    print(f"tail split err {e:.2e}  single launch {e1:.2e}  differ: {not torch.equal(c, single)}")
    assert e < max(2 * e1, 2e-6 * ref.abs().max().item())
    np.testing.assert_array_equal(c.cpu().numpy(), c2.cpu().numpy())
    ws = c._dua_colsums_rows.cpu().double()                       # [ceil(M/64)][2][N] column sums / sums of squares per 64-row block
    blocks = c.cpu().double()[: (M // 64) * 64].reshape(M // 64, 64, N)
    np.testing.assert_allclose(ws[: M // 64, 0].numpy(), blocks.sum(1).numpy(), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(ws[: M // 64, 1].numpy(), (blocks ** 2).sum(1).numpy(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 320), (3, 8, 8, 64), (2, 32, 32, 1280)])
def test_groupnorm_prebias_planes(ops, N, H, W, C):
    """silu(GroupNorm(x + t)) for a per-(sample, channel) bias t with neither the sum nor its statistics materialised (column sums of
    x from the conv epilogue, t folded in): against torch in fp64 through the plane decode."""
    x = rnd(95, N, 32, H, W)
    w = rnd(96, C, 32, 3, 3) / math.sqrt(9 * 32)
    t, ga, be = rnd(97, N, C) * 2, rnd(98, C), rnd(99, C)
    h = ops.conv3x3_bf16x3(_f16x3_planes(ops, nhwc(x)), ops.pack_weight_f16x3(w.to(dev)), None, C, colsum=True)      # carries its column sums
    pl = ops.groupnorm_prebias_planes(h, t.to(dev), ga.to(dev), be.to(dev), act=1, groups=32, eps=1e-5)
    assert pl is not None and pl.fmt == "f16x3"
    hh = nchw(h.cpu()).double() + t.double()[:, :, None, None]
    ref = F.silu(F.group_norm(hh, 32, ga.double(), be.double(), 1e-5))
    buf = pl.buf.view(torch.uint8).reshape(2, -1).cpu()
    dec = (buf[0].view(torch.float16).double() + buf[1].view(torch.float16).double()).reshape(N, H, W, C) / pl.scale
    err = (nchw(dec) - ref).abs().max().item()
    print(f"prebias GroupNorm planes: max err {err:.2e}")
    assert err < 2e-5
    assert ops.groupnorm_prebias_planes(h.clone(), t.to(dev), ga.to(dev), be.to(dev)) is None        # no column sums: the caller's slow path


def test_geglu_and_split_planes(ops):
    """GEGLU (diffusers activations.GEGLU, exact erf GELU) in fp32 and as f16x3 operand planes (hi + lo = 22 significant bits of y * scale)."""
    u = rnd(70, 37, 256) * 3
    a, g = u.chunk(2, dim=-1)
    ref = (a.double() * F.gelu(g.double()))
    y = ops.geglu(u.to(dev)).cpu()
    assert (y.double() - ref).abs().max().item() < 1e-5
    pl = ops.geglu(u.to(dev), split="f16x3")
    assert pl.fmt == "f16x3" and pl.shape == (37, 128) and pl.scale == ops.F16X3_RAW_SCALE
    buf = pl.buf.view(torch.uint8).reshape(2, -1).cpu()
    hi, lo = buf[0].view(torch.float16).reshape(37, 128).double(), buf[1].view(torch.float16).reshape(37, 128).double()
    np.testing.assert_array_equal(hi.float().numpy(), (y * pl.scale).half().float().numpy())
    assert ((hi + lo) / pl.scale - y.double()).abs().max().item() <= 2.0 ** -25 / pl.scale + 2.0 ** -21 * y.abs().max().item()


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 16, 16, 1280, 1280), (2, 32, 32, 640, 320), (1, 8, 8, 1280, 1280), (3, 20, 20, 96, 200),
                                             (2, 64, 64, 320, 320)])
def test_conv3x3_f16x3_split_k(ops, N, H, W, Cin, Cout):
    """Split-K of launches that cannot fill the chip (dua_set_splitk_workspace): the K range of every tile on several workgroups, the
    parts folded in index order by a second kernel together with bias / residual / GroupNorm statistics.  Held to the single-pass
    kernel's error against fp64; statistics equal those recomputed from the output; two runs bit-identical."""
    x, w, b = rnd(20, N, Cin, H, W), rnd(21, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(22, Cout) * 0.1
    res = rnd(23, N, Cout, H, W)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double()
    xs = _f16x3_planes(ops, nhwc(x))
    packed = ops.pack_weight_f16x3(w.to(dev))
    single = ops.conv3x3_bf16x3(xs, packed, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
    prev = ops.set_split_k(True, dev)
    try:
        y = ops.conv3x3_bf16x3(xs, packed, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
        y2 = ops.conv3x3_bf16x3(xs, packed, b.to(dev), Cout, residual=nhwc(res).to(dev), colsum=True)
        st = ops.groupnorm_stats(y, groups=Cout // 8 if Cout % 8 == 0 else 4)          # from the reducer's column sums
        w1 = rnd(24, Cout, Cin) / math.sqrt(Cin)
        c = ops.gemm_bf16x3_presplit(xs, ops.pack_weight_f16x3(w1.to(dev)), Cout, bias=b.to(dev))
    finally:
        ops.set_split_k(prev, dev)
    e_split = (nchw(y.cpu()).double() - ref).abs().max().item()
    e_single = (nchw(single.cpu()).double() - ref).abs().max().item()
    print(f"split-K err {e_split:.2e}  single-pass err {e_single:.2e}  differ: {not torch.equal(y, single)}")
    assert e_split < max(2 * e_single, 2e-6 * ref.abs().max().item())
    np.testing.assert_array_equal(y.cpu().numpy(), y2.cpu().numpy())
    y_plain = y.clone()                                                        # (no attached statistics: the stand-alone kernels)
    st_ref = ops.groupnorm_stats(y_plain, groups=Cout // 8 if Cout % 8 == 0 else 4)
    if (H * W) % 64 == 0:
        assert getattr(y, "_dua_colsums", None) is not None
    np.testing.assert_allclose(st.cpu().numpy(), st_ref.cpu().numpy(), rtol=2e-4, atol=2e-5)
    ref1 = nhwc(x).reshape(-1, Cin).double() @ w1.double().t() + b.double()
    c32 = ops.gemm(nhwc(x).reshape(-1, Cin).to(dev), w1.to(dev), bias=b.to(dev))
    e1, e32 = (c.cpu().double() - ref1).abs().max().item(), (c32.cpu().double() - ref1).abs().max().item()
    assert e1 < max(2 * e32, 2e-6 * ref1.abs().max().item()), (e1, e32)


@pytest.mark.parametrize("M,N,K1,K2", [(128, 128, 32, 0), (200, 72, 64, 0), (5, 2048, 256, 0), (300, 96, 64, 32), (64, 16, 1152, 0),
                                        (4096, 256, 512, 256), (33000, 256, 256, 256), (36000, 136, 64, 0), (70000, 128, 96, 32),
                                        (16384, 1024, 1024, 768)])
def test_gemm_f16x3(ops, M, N, K1, K2):
    """1x1 skip / proj_out GEMM with the raw fp32 operand split into fp16 hi + lo on the fly: held to the exact fp32 engine's error.
    The column sums left for the next GroupNorm are checked against the output itself (ragged last 64-row block included)."""
    a1 = rnd(40, M, K1) * torch.logspace(-3, 2, K1)             # residual-stream-like dynamic range
    a2 = rnd(41, M, K2) * 3 if K2 else None
    w, b, res = rnd(42, N, K1 + K2) / math.sqrt(K1 + K2) * torch.logspace(-2, 1, N).reshape(-1, 1), rnd(43, N) * 0.1, rnd(44, M, N)
    a = a1 if a2 is None else torch.cat([a1, a2], 1)
    ref = a.double() @ w.double().t() + b.double() + res.double()
    sc = ref.abs().amax(dim=0, keepdim=True)
    dv = lambda t: None if t is None else t.to(dev)
    c = ops.gemm_f16x3(dv(a1), ops.pack_weight_f16x3(w.to(dev)), N, bias=dv(b), a2=dv(a2), residual=dv(res), colsum=(N % 4 == 0))
    c32 = ops.gemm(dv(a1), w.to(dev), bias=dv(b), a2=dv(a2), residual=dv(res))
    e, e32 = ((c.cpu().double() - ref).abs() / sc).max().item(), ((c32.cpu().double() - ref).abs() / sc).max().item()
    print(f"gemm f16x3 rel err {e:.2e}  fp32-MFMA rel err {e32:.2e}")
    assert e < max(2 * e32, 2e-6), (e, e32)
    ws = getattr(c, "_dua_colsums_rows", None)
    if ws is not None:
        nb = (M + 63) // 64
        pad = torch.zeros(nb * 64, N, dtype=torch.float64)
        pad[:M] = c.cpu().double()
        blk = pad.reshape(nb, 64, N)
        w2 = ws.cpu().double().reshape(nb, 2, N)
        assert ((w2[:, 0] - blk.sum(1)).abs().max() / blk.abs().sum(1).max()).item() < 1e-5
        assert ((w2[:, 1] - (blk * blk).sum(1)).abs().max() / (blk * blk).sum(1).max()).item() < 1e-5
    # out-of-range operands: hi saturates at 65504 / scale and lo carries the remainder - finite and close, never inf
    big = a1.clone()
    big[0, 0] = 6000.0                                           # * 16 = 96000 > 65504
    cb = ops.gemm_f16x3(dv(big), ops.pack_weight_f16x3(w.to(dev)), N, a2=dv(a2))
    refb = (big if a2 is None else torch.cat([big, a2], 1)).double() @ w.double().t()
    assert torch.isfinite(cb).all() and ((cb.cpu().double() - refb).abs() / refb.abs().amax(dim=0, keepdim=True)).max().item() < 1e-3


@pytest.mark.parametrize("M,N,K1,K2", [(4096, 256, 256, 256), (33000, 256, 512, 256), (700, 512, 768, 512), (300, 1024, 1024, 1024), (257, 768, 1280, 0),
                                        (5, 2048, 256, 0), (1000, 72, 64, 0), (9000, 136, 96, 32), (64, 16, 1152, 0), (65536, 256, 32, 32),
                                        (500, 128, 96, 0), (700, 256, 96, 64), (2000, 320, 224, 0)])           # (odd numbers of 32-deep K steps: 3, 5, 7)
def test_gemm_f16x3_skip_kernel(ops, M, N, K1, K2):
    """The plain form of dua_gemm_f16x3 (bias only: the ADM 1x1 skip convolutions over the virtual concat, unet_openai.py:219-223,257,662)
    on the persistent raw-operand kernel k_sgemm (csrc/gemm_skip_f16x3.hip): held to the exact fp32 engine's error against fp64, within
    rounding of the previous engine (k_xtile), bit-identical between its two tile widths and for any subset of the rows (batch-invariant)."""
    from diffusion_uncertainty_amd import _lib
    L = _lib.lib()
    a1 = rnd(50, M, K1) * torch.logspace(-3, 2, K1)             # residual-stream-like dynamic range
    a2 = rnd(51, M, K2) * 3 if K2 else None
    w, b = rnd(52, N, K1 + K2) / math.sqrt(K1 + K2) * torch.logspace(-2, 1, N).reshape(-1, 1), rnd(53, N) * 0.1
    a = a1 if a2 is None else torch.cat([a1, a2], 1)
    ref = a.double() @ w.double().t() + b.double()
    sc = ref.abs().amax(dim=0, keepdim=True)
    dv = lambda t: None if t is None else t.to(dev)
    pk = ops.pack_weight_f16x3(w.to(dev))
    try:
        assert L.dua_set_sgemm_mode(0) == 0
        c = ops.gemm_f16x3(dv(a1), pk, N, bias=dv(b), a2=dv(a2))
        assert L.dua_set_sgemm_mode(2) == 0
        c2 = ops.gemm_f16x3(dv(a1), pk, N, bias=dv(b), a2=dv(a2))
        assert L.dua_set_sgemm_mode(4) == 0
        c4 = ops.gemm_f16x3(dv(a1), pk, N, bias=dv(b), a2=dv(a2))
        assert L.dua_set_sgemm_mode(-1) == 0
        cx = ops.gemm_f16x3(dv(a1), pk, N, bias=dv(b), a2=dv(a2))              # previous engine (k_xtile)
        assert L.dua_set_sgemm_mode(0) == 0
        # a subset of the rows (another "batch"): same bits per row
        lo, hi = M // 3, M // 3 + max(1, M // 2)
        cs = ops.gemm_f16x3(dv(a1[lo:hi].contiguous()), pk, N, bias=dv(b), a2=dv(None if a2 is None else a2[lo:hi].contiguous()))
        nb = ops.gemm_f16x3(dv(a1), pk, N, a2=dv(a2))                          # no bias
    finally:
        L.dua_set_sgemm_mode(0)
    c32 = ops.gemm(dv(a1), w.to(dev), bias=dv(b), a2=dv(a2))
    e, ex, e32 = [((t.cpu().double() - ref).abs() / sc).max().item() for t in (c, cx, c32)]
    print(f"skip gemm rel err {e:.2e}  previous engine {ex:.2e}  fp32-MFMA {e32:.2e}")
    assert e < max(2 * e32, 2e-6), (e, e32)
    assert torch.equal(c, c2) and torch.equal(c, c4), "tile widths must agree bit for bit"
    assert torch.equal(cs, c[lo:hi]), "a row's result must not depend on the rows around it"
    assert ((nb.cpu().double() + b.double() - c.cpu().double()).abs() / sc).max().item() < 1e-6


def test_gemm_f16x3_skip_kernel_saturation_and_range_check(ops):
    """Out-of-range operands on the raw-operand kernel: hi saturates and lo carries the remainder (finite, close); the range check counts
    each operand element once (two waves share a row block, several column tiles share a row tile)."""
    M, K, N = 600, 128, 384
    a = rnd(60, M, K)
    a[3, 5] = 6000.0                                            # * 16 = 96000 > 65504
    a[400, 77] = 1e-7                                           # * 16 < 2^-14: subnormal hi
    w = rnd(61, N, K) / math.sqrt(K)
    pk = ops.pack_weight_f16x3(w.to(dev))
    c = ops.gemm_f16x3(a.to(dev), pk, N)
    ref = a.double() @ w.double().t()
    assert torch.isfinite(c).all() and ((c.cpu().double() - ref).abs() / ref.abs().amax(dim=0, keepdim=True)).max().item() < 1e-3
    ops.range_check(True)
    try:
        ops.gemm_f16x3(a.to(dev), pk, N)
        sat, sub = ops.range_check_read(reset=True)
    finally:
        ops.range_check(False)
    assert (sat, sub) == (1, 1), (sat, sub)


def test_row_shift_zero_padding_forms_agree(ops):
    """the row-shift conv gets its left / right zero padding from DS reads beyond the LDS allocation.  The same kernel
    built with -DDUA_RS_ZERO_SELECT=1 (scripts/diag/build/libdua_hip_select.so, made by __graft_entry__.build()) zeroes those
    fragments by a register select instead; both must agree bit for bit."""
    import ctypes
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    alt = os.path.join(root, "scripts", "diag", "build", "libdua_hip_select.so")
    if not os.path.exists(alt):
        import __graft_entry__
        __graft_entry__.build()
    A = ctypes.CDLL(alt)
    from diffusion_uncertainty_amd import _lib
    fn = A.dua_conv3x3_f16x3_presplit
    fn.restype, fn.argtypes = _lib.SIGNATURES["dua_conv3x3_f16x3_presplit"]
    A.dua_set_dma_conv.restype, A.dua_set_dma_conv.argtypes = ctypes.c_int, [ctypes.c_int]
    L = _lib.lib()
    for (N, H, W, Cin, Cout) in [(8, 32, 32, 64, 256), (4, 64, 64, 32, 256), (16, 16, 16, 96, 256), (130, 8, 8, 64, 256)]:
        x, w, b = rnd(50, N, Cin, H, W), rnd(51, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(52, Cout) * 0.1
        xs = _f16x3_planes(ops, nhwc(x))
        packed = ops.pack_weight_f16x3(w.to(dev))
        bd = b.to(dev)
        L.dua_set_dma_conv(7)                                # the 16x16x32 row-shift kernel for every tile class
        A.dua_set_dma_conv(7)
        try:
            y0 = ops.conv3x3_bf16x3(xs, packed, bd, Cout)
            y1 = torch.empty_like(y0)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = fn(ctypes.c_void_p(xs.buf.data_ptr()), ctypes.c_void_p(packed.data_ptr()), ctypes.c_void_p(bd.data_ptr()), None,
                    ctypes.c_void_p(y1.data_ptr()), N, H, W, Cin, Cout, 0, None, xs.scale, st)
            assert rc == 0
        finally:
            L.dua_set_dma_conv(1)
            A.dua_set_dma_conv(1)
        assert torch.equal(y0, y1), (N, H, W, Cin, Cout)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        assert (nchw(y0.cpu()).double() - ref).abs().max().item() < 2e-5


def test_f16x3_fp16_subnormal_operands(ops):
    """The lo plane of small operands is a subnormal fp16: v_mfma_f32_32x32x16_f16 must not flush it (the format's error
    analysis in include/dua_hip.h assumes at most 2^-25 absolute on the scaled operand)."""
    N, H, W, Cin, Cout = 1, 8, 8, 32, 32
    x = rnd(30, N, Cin, H, W) * 1e-3                     # scaled by 2^6: |s| ~ 0.06 -> |lo| <= 2^-15 (subnormal: < 6.1e-5)
    w = rnd(31, Cout, Cin, 3, 3) / math.sqrt(9 * Cin)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y = ops.conv3x3_bf16x3(_f16x3_planes(ops, nhwc(x)), ops.pack_weight_f16x3(w.to(dev)), None, Cout)
    rel = ((nchw(y.cpu()).double() - ref).abs().max() / ref.abs().max()).item()
    print(f"subnormal-lo operands: rel err {rel:.2e}")
    assert rel < 5e-6, rel                               # flushed lo planes would leave ~2^-12 = 2.4e-4


@pytest.mark.parametrize("M,N,K1,K2", [(128, 128, 32, 0), (200, 72, 64, 0), (5, 2048, 256, 0), (300, 96, 64, 32), (64, 16, 1152, 0)])
def test_gemm_bf16x3(ops, M, N, K1, K2):
    a1, a2 = rnd(30, M, K1), (rnd(31, M, K2) if K2 else None)
    w, b, r = rnd(32, N, K1 + K2) / math.sqrt(K1 + K2), rnd(33, N), rnd(34, M, N)
    a = torch.cat([a1, a2], 1) if K2 else a1
    ref = a.double() @ w.double().T
    planes = ops.pack_weight_bf16x3(w.to(dev))
    c = ops.gemm_bf16x3(a1.to(dev), planes, N, bias=b.to(dev), a2=(a2.to(dev) if K2 else None), residual=r.to(dev), alpha=0.5)
    assert (c.cpu().double() - (0.5 * ref + b + r)).abs().max().item() < 1e-4
    c = ops.gemm_bf16x3(a1.to(dev), planes, N, bias=b.to(dev), a2=(a2.to(dev) if K2 else None), act_in=1)
    assert (c.cpu().double() - (F.silu(a.double()) @ w.double().T + b)).abs().max().item() < 1e-4
    c = ops.gemm_bf16x3(a1.to(dev), planes, N, bias=b.to(dev), a2=(a2.to(dev) if K2 else None), act_out=2)
    assert (c.cpu().double() - F.gelu(ref + b)).abs().max().item() < 1e-4


@pytest.mark.parametrize("N,S,H,perc", [(7, 2, 5, 0.15), (100, 3, 16, 0.15), (257, 2, 9, 0.5), (1000, 1, 32, 0.9), (3, 1, 4, 0.0)])
def test_pixelwise_threshold_bit_exact(N, S, H, perc):
    from diffusion_uncertainty_amd.thresholds import compute_threshold_pixel_wise
    from oracle import thresholds as oth
    u = rnd(60, N, S, 3, H, H) ** 2
    u[: N // 2, :, :, 0, 0] = 0.25                       # ties
    u[0, :, 1, 1, 1] = -1.0                              # a negative value orders below everything
    got = compute_threshold_pixel_wise(u.to(dev), perc)
    ref = oth.compute_threshold_pixel_wise(u, perc)
    assert got.dtype == torch.float16 and got.shape == (S, 3, H, H)
    assert torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("N,H,Cin,Cout,x3", [(2, 8, 32, 64, False), (3, 16, 64, 96, True), (2, 16, 96, 256, True), (1, 32, 256, 256, True), (2, 8, 4, 32, False)])
def test_epilogue_colsums_feed_groupnorm(ops, N, H, Cin, Cout, x3):
    """Column sums left by the conv epilogue give the same GroupNorm statistics as the stand-alone kernel."""
    x, w, b = rnd(70, N, Cin, H, H), rnd(71, Cout, Cin, 3, 3) / math.sqrt(9 * Cin), rnd(72, Cout) * 0.5
    res = nhwc(rnd(73, N, Cout, H, H)).to(dev)
    if Cin <= 4:
        y = ops.conv3x3(ops.nchw_to_nhwc(x.to(dev), 4), ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, residual=res, colsum=True)
    elif x3:
        y = ops.conv3x3_bf16x3(nhwc(x).to(dev), ops.pack_weight_bf16x3(w.to(dev)), b.to(dev), Cout, residual=res, colsum=True)
    else:
        y = ops.conv3x3(nhwc(x).to(dev), ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, residual=res, colsum=True)
    assert hasattr(y, "_dua_colsums")
    fused = ops.groupnorm_stats(y)
    plain = ops.groupnorm_stats(y.clone())                       # clone drops the attribute -> stand-alone kernel
    np.testing.assert_allclose(fused.cpu().numpy(), plain.cpu().numpy(), rtol=2e-5, atol=2e-6)
    # two-source (virtual concat) statistics from two producers
    y2 = ops.conv3x3(nhwc(x).to(dev) if Cin > 4 else ops.nchw_to_nhwc(x.to(dev), 4), ops.pack_conv3x3_weight(w.to(dev)), b.to(dev), Cout, colsum=True)
    fused = ops.groupnorm_stats(y, y2)
    plain = ops.groupnorm_stats(y.clone(), y2.clone())
    np.testing.assert_allclose(fused.cpu().numpy(), plain.cpu().numpy(), rtol=2e-5, atol=2e-6)


def test_quantile_workspace_release_and_reuse(ops):
    """dua_quantile_workspace_release: the two-workgroup kernel's per-(device, stream) counter workspace can be freed - for one
    stream (before it is destroyed) or for all - and the next launch allocates a fresh, zeroed one: masks stay bit-exact against
    torch.quantile before and after, on the default stream and on a side stream."""
    from diffusion_uncertainty_amd import _lib
    L = _lib.lib()
    u = (rnd(11, 6, 3, 128, 128) ** 2).to(dev)                           # n = 49152: the two-workgroup kernel's range

    def check():
        mask, q = ops.quantile_mask(u, 0.9)
        want_q = torch.quantile(u.flatten(1), 0.9, dim=1)
        assert torch.equal(q, want_q) and torch.equal(mask.bool(), u > want_q.view(-1, 1, 1, 1))

    check()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        check()
    side.synchronize()
    assert L.dua_quantile_workspace_release(side.cuda_stream, 0) == 1    # that stream's workspace only
    assert L.dua_quantile_workspace_release(side.cuda_stream, 0) == 0
    check()
    assert L.dua_quantile_workspace_release(None, 1) >= 1                # everything of this device
    check()
    with torch.cuda.stream(side):
        check()
    side.synchronize()


def test_kernel_profile_facility(ops):
    """dua_kernel_profile / dua_kernel_profile_collect (bench.py's hbm_kernels / leg rooflines): off by default (nothing recorded), on:
    one record per instrumented launch with the launch's ALGORITHMIC bytes, summed per kind, as JSON; a too-small buffer is an error."""
    import ctypes
    import json

    from diffusion_uncertainty_amd import _lib
    L = _lib.lib()
    buf = ctypes.create_string_buffer(1 << 14)
    L.dua_kernel_profile_collect(buf, len(buf))
    n = 3 * 64 * 64 * 8
    a, b = rnd(1, n).to(dev), rnd(2, n).to(dev)
    ops.axpby(a, b, 1.0, 2.0)
    assert L.dua_kernel_profile_collect(buf, len(buf)) == 2 and buf.value == b"{}"            # off: nothing
    assert L.dua_kernel_profile(1) == 0
    try:
        for _ in range(3):
            ops.axpby(a, b, 1.0, 2.0)
        ops.to_uint8(a.reshape(8, 3, 64, 64))
        torch.cuda.synchronize()
    finally:
        assert L.dua_kernel_profile(0) == 1
    small = ctypes.create_string_buffer(8)
    assert L.dua_kernel_profile_collect(small, len(small)) == -1 and b"buffer" in L.dua_last_error()        # (nothing is lost: checked first)
    ops.axpby(a, b, 1.0, 2.0)                                              # off again: not recorded
    ln = L.dua_kernel_profile_collect(buf, len(buf))
    rec = json.loads(buf.value.decode())
    assert ln == len(buf.value) and set(rec) == {"k_axpby", "k_to_uint8"}
    assert rec["k_axpby"]["launches"] == 3 and rec["k_axpby"]["bytes"] == 3 * 12.0 * n and rec["k_axpby"]["flops"] == 0
    assert rec["k_to_uint8"]["launches"] == 1 and rec["k_to_uint8"]["bytes"] == 5.0 * n
    assert 0 < rec["k_axpby"]["min_ms"] <= rec["k_axpby"]["ms"] < 50
