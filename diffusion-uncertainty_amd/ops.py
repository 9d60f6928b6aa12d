"""Tensor-level wrappers over the C ABI (include/dua_hip.h).

torch is used for device memory and the current HIP stream only; every arithmetic step runs in
libdua_hip.so.  All tensors must be contiguous fp32 (or the stated integer type) on a HIP device.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._lib import AttnBlockArgs, GemmArgs, ResBlockArgs, UViTBlockArgs, check


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _req(t, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a GPU tensor (got {getattr(t, 'device', type(t))}); "
                           "there is no CPU fallback in this package")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _f32c(t):
    """reference semantics: fp16/bf16 model outputs are promoted by the fp32 sample."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------ sampler kernels
def ddim_step(eps, x, sqrt_beta_prod_t, sqrt_alpha_prod_t, sqrt_alpha_prod_prev, dir_coef, sigma=0.0, noise=None,
              clip=True, clip_range=1.0):
    eps, x = _req(_f32c(eps), name="eps"), _req(_f32c(x), name="x")
    assert eps.shape == x.shape
    prev, x0 = torch.empty_like(x), torch.empty_like(x)
    if noise is not None:
        _req(noise, name="noise")
    check(_lib.lib().dua_ddim_step(_p(eps), _p(x), _p(noise), _p(prev), _p(x0), x.numel(), float(sqrt_beta_prod_t),
                                   float(sqrt_alpha_prod_t), float(sqrt_alpha_prod_prev), float(dir_coef), float(sigma),
                                   int(bool(clip)), float(clip_range), _stream()), "dua_ddim_step")
    return prev, x0


def axpby(src, noise, a, b, out=None):
    _req(src, name="src"), _req(noise, name="noise")
    assert src.shape == noise.shape
    out = torch.empty_like(src) if out is None else out
    check(_lib.lib().dua_axpby(_p(src), _p(noise), _p(out), src.numel(), float(a), float(b), _stream()), "dua_axpby")
    return out


def x0_from_eps(eps, x, sigma_t, alpha_t):
    """(x - sigma_t*eps) / alpha_t  (DPM-Solver++ data prediction)."""
    _req(eps, name="eps"), _req(x, name="x")
    assert eps.shape == x.shape
    out = torch.empty_like(x)
    check(_lib.lib().dua_x0_from_eps(_p(eps), _p(x), _p(out), x.numel(), float(sigma_t), float(alpha_t), _stream()), "dua_x0_from_eps")
    return out


def dpmpp_2m_update(sample, m0, m1, c1, c2, c3, rinv):
    """(c1*sample - c2*m0) - c3*(rinv*(m0 - m1))  (second-order multistep DPM-Solver++, midpoint)."""
    _req(sample, name="sample"), _req(m0, name="m0"), _req(m1, name="m1")
    assert sample.shape == m0.shape == m1.shape
    out = torch.empty_like(sample)
    check(_lib.lib().dua_dpmpp_2m_update(_p(sample), _p(m0), _p(m1), _p(out), sample.numel(), float(c1), float(c2), float(c3),
                                         float(rinv), _stream()), "dua_dpmpp_2m_update")
    return out


def mc_centered_accum(eps_m, eps, acc, m, M):
    _req(eps_m, name="eps_m"), _req(eps, name="eps"), _req(acc, name="acc")
    check(_lib.lib().dua_mc_centered_accum(_p(eps_m), _p(eps), _p(acc), eps.numel(), int(m), int(M), _stream()),
          "dua_mc_centered_accum")
    return acc


def mc_reduce_centered(stack, eps):
    _req(stack, name="stack"), _req(eps, name="eps")
    u = torch.empty_like(eps)
    check(_lib.lib().dua_mc_reduce_centered(_p(stack), _p(eps), _p(u), eps.numel(), stack.shape[0], _stream()),
          "dua_mc_reduce_centered")
    return u


def mc_reduce_var(stack):
    _req(stack, name="stack")
    u = torch.empty(stack.shape[1:], device=stack.device, dtype=torch.float32)
    check(_lib.lib().dua_mc_reduce_var(_p(stack), _p(u), u.numel(), stack.shape[0], _stream()), "dua_mc_reduce_var")
    return u


def quantile_rank_split(p, n):
    """fp32 rank arithmetic of torch.quantile."""
    rank = np.float32(p) * np.float32(n - 1)
    lo, hi = int(np.floor(rank)), int(np.ceil(rank))
    return lo, hi, float(np.float32(rank - np.float32(lo)))


def quantile_mask(u, p):
    """mask = u > torch.quantile(u.flatten(1), p, dim=1) per sample -> (mask uint8 like u, q [B])."""
    _req(u, name="u")
    B = u.shape[0]
    n = u.numel() // max(B, 1)
    lo, hi, w = quantile_rank_split(p, n)
    mask = torch.empty(u.shape, device=u.device, dtype=torch.uint8)
    q = torch.empty(B, device=u.device, dtype=torch.float32)
    check(_lib.lib().dua_quantile_mask(_p(u), B, n, lo, hi, w, _p(mask), _p(q), _stream()), "dua_quantile_mask")
    return mask, q


def flip_h(x):
    """torch.flip(x, dims=[2]) of an NCHW tensor."""
    _req(x, name="x")
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    check(_lib.lib().dua_flip_h(_p(x), _p(y), N * C, H, W, _stream()), "dua_flip_h")
    return y


def posterior_blend(eps, u, mask, M, alpha_hat_t):
    _req(eps, name="eps"), _req(u, name="u"), _req(mask, torch.uint8, "mask")
    B = eps.shape[0]
    out = torch.empty_like(eps)
    check(_lib.lib().dua_posterior_blend(_p(eps), _p(u), _p(mask), _p(out), B, eps.numel() // max(B, 1), int(M),
                                         float(alpha_hat_t), _stream()), "dua_posterior_blend")
    return out


def threshold_map(u, thr, per_sample, lower=False):
    """0/1 float map of u > thr (or u < thr): thr is [B] per-sample scalars (per_sample) or one [n] map broadcast over B."""
    _req(u, name="u"), _req(thr, name="thr")
    B = u.shape[0]
    n = u.numel() // max(B, 1)
    assert thr.numel() == (B if per_sample else n), f"threshold has {thr.numel()} elements, expected {B if per_sample else n}"
    out = torch.empty_like(u)
    check(_lib.lib().dua_threshold_map(_p(u), _p(thr), _p(out), B, n, int(bool(per_sample)), int(bool(lower)), _stream()),
          "dua_threshold_map")
    return out


def batch_sum(x):
    _req(x, name="x")
    B = x.shape[0]
    out = torch.empty(x.shape[1:], device=x.device, dtype=torch.float32)
    check(_lib.lib().dua_batch_sum(_p(x), _p(out), B, x.numel() // max(B, 1), _stream()), "dua_batch_sum")
    return out


def posterior_score(u, bsum, M, inv_alpha):
    _req(u, name="u"), _req(bsum, name="bsum")
    B = u.shape[0]
    n = u.numel() // max(B, 1)
    assert bsum.numel() == n
    out = torch.empty_like(u)
    check(_lib.lib().dua_posterior_score(_p(u), _p(bsum), _p(out), B, n, int(M), float(inv_alpha), _stream()), "dua_posterior_score")
    return out


def map_blend(a, b, m):
    """a*m + b*(1-m)."""
    _req(a, name="a"), _req(b, name="b"), _req(m, name="map")
    assert a.shape == b.shape == m.shape
    out = torch.empty_like(a)
    check(_lib.lib().dua_map_blend(_p(a), _p(b), _p(m), _p(out), a.numel(), _stream()), "dua_map_blend")
    return out


def pixelwise_order_statistic(u, k):
    """u: [N, ...] fp32 -> [...] fp16: k-th smallest over dim 0 of u.half() (exact)."""
    _req(u, name="u")
    N = u.shape[0]
    out = torch.empty(u.shape[1:], device=u.device, dtype=torch.float16)
    check(_lib.lib().dua_pixelwise_order_statistic(_p(u), N, u.numel() // max(N, 1), int(k), _p(out), _stream()),
          "dua_pixelwise_order_statistic")
    return out


def to_uint8(x):
    _req(x, name="x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    check(_lib.lib().dua_to_uint8(_p(x), _p(out), x.numel(), _stream()), "dua_to_uint8")
    return out


def randn(shape, device, seed, offset):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    check(_lib.lib().dua_randn(_p(out), out.numel(), int(seed), int(offset), _stream()), "dua_randn")
    return out


def dropout_mask(shape, device, p, seed, offset):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    check(_lib.lib().dua_dropout_mask(_p(out), out.numel(), float(p), int(seed), int(offset), _stream()), "dua_dropout_mask")
    return out


# ------------------------------------------------------------------ layout
def nchw_to_nhwc(x, cpad=None):
    _req(x, name="x")
    N, C, H, W = x.shape
    cpad = C if cpad is None else cpad
    y = torch.empty((N, H, W, cpad), device=x.device, dtype=torch.float32)
    check(_lib.lib().dua_nchw_to_nhwc(_p(x), _p(y), N, C, H, W, cpad, _stream()), "dua_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x, ckeep=None):
    _req(x, name="x")
    N, H, W, C = x.shape
    ckeep = C if ckeep is None else ckeep
    y = torch.empty((N, ckeep, H, W), device=x.device, dtype=torch.float32)
    check(_lib.lib().dua_nhwc_to_nchw(_p(x), _p(y), N, C, H, W, ckeep, _stream()), "dua_nhwc_to_nchw")
    return y


# ------------------------------------------------------------------ denoiser kernels
def _colsum_ws(M, cout, device, want):
    """Workspace for the epilogue column sums ([ceil(M/64)][2][cout]) or None."""
    if not want or cout % 4:
        return None
    return torch.empty(((M + 63) // 64, 2, cout), device=device, dtype=torch.float32)


def _attach_colsums(y, ws, hw):
    """Remember the producer's column sums on the tensor (consumed by groupnorm_stats when H*W % 64 == 0)."""
    if ws is not None and hw % 64 == 0:
        y._dua_colsums = ws
    return y


def groupnorm_prebias_planes(x, tbias, gamma, beta, act=1, groups=32, eps=1e-5):
    """act(GroupNorm(x + tbias[:, None, None, :])) as f16x3 planes WITHOUT materialising x + tbias (diffusers ResnetBlock2D: the
    time-embedding projection is added in front of norm2): statistics from the column sums x's producer left behind, the bias
    folded into them and into the affine coefficients.  None when x carries no column sums (the caller then adds the bias itself)."""
    _req(x, name="x")
    N, H, W, C = x.shape
    cs = getattr(x, "_dua_colsums", None)
    if cs is None or (H * W) % 64 or C % 32:
        return None
    assert tbias.is_cuda and tbias.dtype == torch.float32 and tbias.shape == (N, C) and tbias.stride(1) == 1
    stats = torch.empty((N, groups, 2), device=x.device, dtype=torch.float32)
    L = _lib.lib()
    check(L.dua_groupnorm_stats_from_colsums_bias(_p(cs), C, N, H * W, groups, float(eps), _p(tbias), tbias.stride(0), _p(stats), _stream()),
          "dua_groupnorm_stats_from_colsums_bias")
    buf = torch.empty((2, N, H, W, C), device=x.device, dtype=torch.int16)
    check(L.dua_groupnorm_apply_prebias_split_f16x3(_p(x), C, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(tbias), tbias.stride(0),
                                                    int(act), F16X3_ACT_SCALE, _p(buf), _stream()), "dua_groupnorm_apply_prebias_split_f16x3")
    pl = SplitPlanes(buf, (N, H, W, C), "f16x3", F16X3_ACT_SCALE)
    pl.stats = stats                                  # (mean, rstd) of x + tbias per (sample, group): what a backward pass needs
    return pl


def groupnorm_stats(x1, x2=None, groups=32, eps=1e-5):
    _req(x1, name="x1")
    N, H, W, C1 = x1.shape
    cs1 = getattr(x1, "_dua_colsums", None)
    cs2 = getattr(x2, "_dua_colsums", None) if x2 is not None else None
    if cs1 is not None and (x2 is None or cs2 is not None) and (H * W) % 64 == 0:
        # statistics from the column sums the producing conv / GEMM epilogue left behind: x is not read again
        stats = torch.empty((N, groups, 2), device=x1.device, dtype=torch.float32)
        check(_lib.lib().dua_groupnorm_stats_from_colsums(_p(cs1), C1, _p(cs2), 0 if x2 is None else x2.shape[3], N, H * W, groups,
                                                          float(eps), _p(stats), _stream()), "dua_groupnorm_stats_from_colsums")
        return stats
    C2 = 0
    if x2 is not None:
        _req(x2, name="x2")
        assert x2.shape[:3] == x1.shape[:3]
        C2 = x2.shape[3]
    L = _lib.lib()
    ws = torch.empty(L.dua_groupnorm_ws_floats(N, H * W, C1 + C2, groups), device=x1.device, dtype=torch.float32)
    stats = torch.empty((N, groups, 2), device=x1.device, dtype=torch.float32)
    check(L.dua_groupnorm_stats(_p(x1), C1, _p(x2), C2, N, H * W, groups, float(eps), _p(ws), _p(stats), _stream()),
          "dua_groupnorm_stats")
    return stats


class SplitPlanes:
    """Activation tensor stored as two bf16 NHWC planes (hi | lo) - the operand format of the bf16x3 conv."""

    def __init__(self, buf, shape, fmt="bf16x3", scale=1.0):
        # fmt "bf16x3": bf16 hi | bf16 lo;
        # "f16x3": fp16 hi | fp16 lo of x * scale (the fp32-equivalent split; scale a power of two)
        self.buf, self.shape, self.device, self.fmt, self.scale = buf, tuple(shape), buf.device, fmt, float(scale)


def _film_ld(film, N, C):
    """Row stride of the FiLM operand: a dense [N, 2C] tensor or a column block (view) of a wider row-major table."""
    if film is None:
        return 0
    assert film.is_cuda and film.dtype == torch.float32 and film.shape == (N, 2 * C) and film.stride(1) == 1, "film: [N, 2C] fp32 rows"
    return film.stride(0)


def groupnorm_apply(x1, stats, gamma, beta, x2=None, film=None, act=1, resample=0, dropmask=None, groups=32, split=False,
                    split_scale=None):
    N, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    Ho, Wo = (H // 2, W // 2) if resample < 0 else ((2 * H, 2 * W) if resample > 0 else (H, W))
    if split:
        fmt = split if split == "f16x3" else "bf16x3"
        buf = torch.empty((2, N, Ho, Wo, C1 + C2), device=x1.device, dtype=torch.int16)
        for t, nm in ((gamma, "gamma"), (beta, "beta"), (stats, "stats")):
            _req(t, name=nm)
        if split == "f32w":
            # exact-fp32 Winograd operand (csrc/conv_wino_f32.hip): the four V_f images as floats, same stage-block layout
            ntm = (N * (Ho // 2) * Wo + 127) // 128
            vbuf = torch.empty((4, ntm, (C1 + C2) // 32, 128, 32), device=x1.device, dtype=torch.float32)
            check(_lib.lib().dua_groupnorm_apply_wino_f32(_p(x1), C1, _p(x2), C2, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(film),
                                                          _film_ld(film, N, C1 + C2), int(act), int(resample), _p(dropmask),
                                                          _p(vbuf), _stream()), "dua_groupnorm_apply_wino_f32")
            return SplitPlanes(vbuf, (N, Ho, Wo, C1 + C2), "f32w", 1.0)
        if split == "f32w4":
            # exact-fp32 F(4,3) operand (csrc/conv_wino4_f32.hip): the six V_f images as floats, stage-block layout of the f16x3 form
            ntm = (N * (Ho // 4) * Wo + 127) // 128
            vbuf = torch.empty((6, ntm, (C1 + C2) // 32, 128, 32), device=x1.device, dtype=torch.float32)
            check(_lib.lib().dua_groupnorm_apply_wino4_f32(_p(x1), C1, _p(x2), C2, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(film),
                                                           _film_ld(film, N, C1 + C2), int(act), int(resample), _p(dropmask),
                                                           _p(vbuf), _stream()), "dua_groupnorm_apply_wino4_f32")
            return SplitPlanes(vbuf, (N, Ho, Wo, C1 + C2), "f32w4", 1.0)
        if split in ("f16x3w", "f16w"):
            # Winograd F(2,3)-along-y operand of conv3x3 (csrc/conv_wino_f16x3.hip): the four V_f images as f16x3 planes of V * scale
            # ("f16w": only the hi images are written - the operand of the single-product conv, mode "f16w")
            scale = 0.5 * F16X3_ACT_SCALE if split_scale is None else float(split_scale)
            ntm = (N * (Ho // 2) * Wo + 127) // 128
            vbuf = torch.empty((4, ntm, (C1 + C2) // 32, 2, 128, 32), device=x1.device, dtype=torch.int16)   # [f][tile][cb][hi | lo][128][32]
            fn = _lib.lib().dua_groupnorm_apply_wino_f16w if split == "f16w" else _lib.lib().dua_groupnorm_apply_wino_f16x3
            if split == "f16w":
                vbuf.zero_()                       # (the lo halves stay unwritten: defined contents for tests that decode the buffer)
            check(fn(_p(x1), C1, _p(x2), C2, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(film), _film_ld(film, N, C1 + C2), int(act),
                     int(resample), _p(dropmask), scale, _p(vbuf), _stream()), "dua_groupnorm_apply_wino_f16x3")
            return SplitPlanes(vbuf, (N, Ho, Wo, C1 + C2), split, scale)
        if split == "f16x3w4":
            # Winograd F(4,3)-along-y operand of conv3x3 (csrc/conv_wino4_f16x3.hip): the six V_f images as f16x3 planes of V * scale
            scale = 0.125 * F16X3_ACT_SCALE if split_scale is None else float(split_scale)
            ntm = (N * (Ho // 4) * Wo + 127) // 128
            vbuf = torch.empty((6, ntm, (C1 + C2) // 32, 2, 128, 32), device=x1.device, dtype=torch.int16)   # [f][tile][cb][hi | lo][128][32]
            check(_lib.lib().dua_groupnorm_apply_wino4_f16x3(_p(x1), C1, _p(x2), C2, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(film),
                                                             _film_ld(film, N, C1 + C2), int(act), int(resample), _p(dropmask), scale, _p(vbuf),
                                                             _stream()), "dua_groupnorm_apply_wino4_f16x3")
            return SplitPlanes(vbuf, (N, Ho, Wo, C1 + C2), split, scale)
        if fmt == "f16x3":
            scale = F16X3_ACT_SCALE if split_scale is None else float(split_scale)
            check(_lib.lib().dua_groupnorm_apply_split_f16x3(_p(x1), C1, _p(x2), C2, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(film),
                                                             _film_ld(film, N, C1 + C2), int(act), int(resample), _p(dropmask),
                                                             scale, _p(buf), _stream()), "dua_groupnorm_apply_split_f16x3")
            return SplitPlanes(buf, (N, Ho, Wo, C1 + C2), fmt, scale)
        fn = _lib.lib().dua_groupnorm_apply_split
        check(fn(_p(x1), C1, _p(x2), C2, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(film), _film_ld(film, N, C1 + C2),
                 int(act), int(resample), _p(dropmask), _p(buf), _stream()), "dua_groupnorm_apply_split")
        return SplitPlanes(buf, (N, Ho, Wo, C1 + C2), fmt)
    y = torch.empty((N, Ho, Wo, C1 + C2), device=x1.device, dtype=torch.float32)
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (stats, "stats")):
        _req(t, name=nm)
    check(_lib.lib().dua_groupnorm_apply(_p(x1), C1, _p(x2), C2, N, H, W, groups, _p(stats), _p(gamma), _p(beta), _p(film),
                                         _film_ld(film, N, C1 + C2), int(act), int(resample), _p(dropmask), _p(y), _stream()),
          "dua_groupnorm_apply")
    return y


def resample2x(x, mode):
    _req(x, name="x")
    N, H, W, C = x.shape
    Ho, Wo = (H // 2, W // 2) if mode < 0 else (2 * H, 2 * W)
    y = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.float32)
    check(_lib.lib().dua_resample2x(_p(x), _p(y), N, H, W, C, int(mode), _stream()), "dua_resample2x")
    return y


def add_channel_bias_(x, bias):
    """x [N, H, W, C] += bias [N, C] (a dense tensor or a column block of a wider table), in place."""
    _req(x, name="x")
    N, H, W, C = x.shape
    assert bias.is_cuda and bias.dtype == torch.float32 and bias.shape == (N, C) and bias.stride(1) == 1
    check(_lib.lib().dua_add_channel_bias(_p(x), _p(bias), bias.stride(0), N, H * W, C, _stream()), "dua_add_channel_bias")
    return x


def geglu(x, split=False):
    """[M, 2F] -> [M, F]: x[:, :F] * gelu(x[:, F:]); split="f16x3": as operand planes at the raw-stream scale."""
    _req(x, name="x")
    M, F2 = x.shape
    if split == "f16x3":
        buf = torch.empty((2, M, F2 // 2), device=x.device, dtype=torch.int16)
        check(_lib.lib().dua_geglu_split_f16x3(_p(x), _p(buf), M, F2 // 2, F16X3_RAW_SCALE, _stream()), "dua_geglu_split_f16x3")
        return SplitPlanes(buf, (M, F2 // 2), "f16x3", F16X3_RAW_SCALE)
    y = torch.empty((M, F2 // 2), device=x.device, dtype=torch.float32)
    check(_lib.lib().dua_geglu(_p(x), _p(y), M, F2 // 2, _stream()), "dua_geglu")
    return y


def pick_even(x):
    """NHWC [N, H, W, C] -> [N, H/2, W/2, C]: the even pixels."""
    _req(x, name="x")
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=torch.float32)
    check(_lib.lib().dua_pick_even(_p(x), _p(y), N, H, W, C, _stream()), "dua_pick_even")
    return y


def attention(q, k, v, heads, precision="fp32", planes_out=False):
    """Multi-head attention with separate operands (cross-attention): q [B, Tq, C], k / v [B, Tk, C], heads of C/heads
    channels, softmax(q k^T / sqrt(ch)) v -> [B, Tq, C].  fp32: MFMA products with the logits in HBM (the SD-1.5 shapes
    are small: Tk = 77 for the text context).  precision="f16x3" (head dims <= 256): the split flash kernel in ONE launch,
    optionally with the result as f16x3 planes [B*Tq, C] (operand of the presplit to_out GEMM)."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v")):
        _req(t, name=nm)
    B, Tq, C = q.shape
    Tk = k.shape[1]
    ch = C // heads
    assert k.shape == (B, Tk, C) and v.shape == (B, Tk, C) and C % heads == 0 and ch % 4 == 0 and B * heads <= 65535
    if precision == "f16x3" and 8 <= ch <= 256 and ATTENTION_MODE in ("auto", "fused_x"):
        if planes_out and C % 32 == 0:
            buf = torch.empty((2, B * Tq, C), device=q.device, dtype=torch.int16)
            check(_lib.lib().dua_cross_attention_f16x3(_p(q), C, _p(k), _p(v), C, None, _p(buf), F16X3_RAW_SCALE, B, Tq, Tk, heads, ch,
                                                       1.0 / math.sqrt(ch), _stream()), "dua_cross_attention_f16x3")
            return SplitPlanes(buf, (B * Tq, C), "f16x3", F16X3_RAW_SCALE)
        out = torch.empty((B, Tq, C), device=q.device, dtype=torch.float32)
        check(_lib.lib().dua_cross_attention_f16x3(_p(q), C, _p(k), _p(v), C, _p(out), None, 0.0, B, Tq, Tk, heads, ch,
                                                   1.0 / math.sqrt(ch), _stream()), "dua_cross_attention_f16x3")
        return out
    Tp = (Tk + 3) // 4 * 4
    S = torch.empty((B * heads, Tq, Tp), device=q.device, dtype=torch.float32)
    out = torch.empty((B, Tq, C), device=q.device, dtype=torch.float32)
    gemm_raw(A1=q.data_ptr(), B=k.data_ptr(), C=S.data_ptr(), M=Tq, N=Tk, K1=ch, K2=0, lda1=C, lda2=4, ldb=C, ldc=Tp, ldr=Tp,
             sA1=Tq * C, sB=Tk * C, sC=heads * Tq * Tp, batch=B * heads, batch_inner=heads, sA1i=ch, sBi=ch, sCi=Tq * Tp,
             transB=1, alpha=1.0 / math.sqrt(ch))
    softmax_rows_(S, B * heads * Tq, Tk, Tp)
    gemm_raw(A1=S.data_ptr(), B=v.data_ptr(), C=out.data_ptr(), M=Tq, N=ch, K1=Tk, K2=0, lda1=Tp, lda2=4, ldb=C, ldc=C, ldr=C,
             sA1=heads * Tq * Tp, sB=Tk * C, sC=Tq * C, batch=B * heads, batch_inner=heads, sA1i=Tq * Tp, sBi=ch, sCi=ch,
             transB=0, alpha=1.0)
    return out


def pack_conv3x3_weight(w):
    """OIHW -> packed device layout consumed by conv3x3."""
    _req(w, name="w")
    Cout, Cin = w.shape[0], w.shape[1]
    L = _lib.lib()
    wp = torch.empty(L.dua_conv3x3_packed_floats(Cout, Cin), device=w.device, dtype=torch.float32)
    check(L.dua_pack_conv3x3_weight(_p(w), _p(wp), Cout, Cin, _stream()), "dua_pack_conv3x3_weight")
    return wp


# Power-of-two scale of the f16x3 activation planes (GroupNorm / SiLU outputs): |x| up to 1023 stays finite in fp16 and
# every |x| >= 2^-9 keeps its 22 significant bits.  The Winograd images derive their scales from it: F(2,3) 0.5 * 64 = 32 (|V| <= 2 |x|:
# the same |x| <= 1023), F(4,3) 0.125 * 64 = 8 (|V| <= 10 |x|: the rows of B^T sum to 4 + 5 + 1 in absolute value, so the hi half
# saturates for |x| > 818 - a fifth less headroom - and values below 2^-6 lose low bits); dua_range_check counts both ends.
F16X3_ACT_SCALE = 64.0
# ... of raw residual-stream operands split on the fly (1x1 skip convolutions, proj_out): |x| up to 4094 stays in fp16's range
# (beyond it the lo half takes over at reduced precision), 22 significant bits down to 2^-7.
F16X3_RAW_SCALE = 16.0


# Split-K workspace of the f16x3 LDS-DMA launches (dua_set_splitk_workspace): S * M * N floats with S * tiles <= 256, i.e. at most
# 256 tiles of 256 x 128 floats = 32 MiB; the stream-K tail of the one-tap GEMM kernel (csrc/gemm_f16x3.hip) keeps up to two partial tiles
# of 256 x 256 floats per workgroup = 128 MiB.  One buffer per device, kept for the life of the process (captured HIP graphs hold its address).
_SPLITK_WS = {}
_SPLITK_ON = False


def set_split_k(on, device=None):
    """Switch split-K of under-filled f16x3 conv / presplit-GEMM launches on or off (process-global); returns the previous setting.
    On: results differ from the single-pass kernels by fp32 summation order and are no longer bit-identical across batch sizes."""
    global _SPLITK_ON
    prev = _SPLITK_ON
    if on:
        dev = torch.device(device if device is not None else "cuda")
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        if key not in _SPLITK_WS:
            _SPLITK_WS[key] = torch.empty(130 << 20, device=dev, dtype=torch.uint8)
        ws = _SPLITK_WS[key]
        check(_lib.lib().dua_set_splitk_workspace(_p(ws), ws.numel()), "dua_set_splitk_workspace")
    else:
        check(_lib.lib().dua_set_splitk_workspace(None, 0), "dua_set_splitk_workspace")
    _SPLITK_ON = bool(on)
    return prev


_RANGE_ON = False


def range_check(on):
    """Switch the f16x3 range counters on / off (dua_range_check); returns the previous setting."""
    global _RANGE_ON
    was = _lib.lib().dua_range_check(1 if on else 0)
    if was < 0:                       # DUA_EHIP: the counters could not be allocated / cleared - not "was on"
        check(was, "dua_range_check")
    _RANGE_ON = bool(on)
    return bool(was)


def range_check_is_on():
    """Whether the counters are on (as last set through range_check; reading this has no side effect on the counters)."""
    return _RANGE_ON


def range_check_read(reset=True):
    """(saturated, subnormal) operand counts since the counters were switched on / last reset (synchronises the device)."""
    a, b = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    check(_lib.lib().dua_range_check_read(ctypes.byref(a), ctypes.byref(b), 1 if reset else 0), "dua_range_check_read")
    return int(a.value), int(b.value)


# Optional per-launch timing hook used by bench.py: an object with .record(kind, flops, start_evt, end_evt).
PROFILE = None


def conv3x3(x, wp, bias, cout, residual=None, nchw_out=False, colsum=False):
    _req(x, name="x"), _req(wp, name="wp")
    N, H, W, Cin = x.shape
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = _conv3x3(x, wp, bias, cout, residual, nchw_out, colsum)
        e1.record()
        PROFILE.record("conv3x3", 2.0 * N * H * W * 9 * Cin * cout, e0, e1)
        return y
    return _conv3x3(x, wp, bias, cout, residual, nchw_out, colsum)


def _conv3x3(x, wp, bias, cout, residual=None, nchw_out=False, colsum=False):
    N, H, W, Cin = x.shape
    ws = _colsum_ws(N * H * W, cout, x.device, colsum and not nchw_out)
    y = torch.empty((N, cout, H, W) if nchw_out else (N, H, W, cout), device=x.device, dtype=torch.float32)
    if residual is not None:
        _req(residual, name="residual")
        assert residual.shape == (N, H, W, cout)
    check(_lib.lib().dua_conv3x3(_p(x), _p(wp), _p(bias), _p(residual), _p(y), N, H, W, Cin, cout, int(nchw_out), _p(ws), _stream()),
          "dua_conv3x3")
    return _attach_colsums(y, ws, H * W)


def gemm(a1, w, bias=None, a2=None, residual=None, act_in=0, act_out=0, alpha=1.0, out=None, colsum=False):
    """C[M,N] = alpha * [a1 | a2] @ w.T (+bias) (+residual); a*: [M,K*] row-major, w: [N, K1+K2]."""
    _req(a1, name="a1"), _req(w, name="w")
    M, K1 = a1.shape
    K2 = 0 if a2 is None else a2.shape[1]
    N = w.shape[0]
    assert w.shape[1] == K1 + K2
    c = torch.empty((M, N), device=a1.device, dtype=torch.float32) if out is None else out
    ws = _colsum_ws(M, N, a1.device, colsum)
    if ws is not None:
        c._dua_colsums_rows = ws          # attached to the 2-D result; the caller re-attaches after reshaping to NHWC
    g = GemmArgs(A1=a1.data_ptr(), A2=(a2.data_ptr() if a2 is not None else None), B=w.data_ptr(),
                 bias=(bias.data_ptr() if bias is not None else None),
                 residual=(residual.data_ptr() if residual is not None else None), C=c.data_ptr(),
                 M=M, N=N, K1=K1, K2=K2, lda1=K1, lda2=max(K2, 4), ldb=K1 + K2, ldc=N, ldr=N,
                 batch=1, batch_inner=1, transB=1, act_in=act_in, act_out=act_out, alpha=alpha,
                 colsum=(ws.data_ptr() if ws is not None else None))
    check(_lib.lib().dua_gemm(ctypes.byref(g), _stream()), "dua_gemm")
    return c


# ------------------------------------------------------------------ split-bf16 (bf16x3) variants
def pack_weight_bf16x3(w):
    """fp32 [N][K] linear / 1x1 weight, or OIHW 3x3 conv weight -> two bf16 planes (hi | lo) in one int16 buffer."""
    _req(w, name="w")
    L = _lib.lib()
    if w.ndim == 4 and w.shape[2] == 3:
        N, cin = w.shape[0], w.shape[1]
        K = 9 * cin
    else:
        N, cin = w.shape[0], 0
        K = w.numel() // N
    planes = torch.empty(2 * L.dua_bf16x3_plane_elems(N, K), device=w.device, dtype=torch.int16)
    check(L.dua_pack_weight_bf16x3(_p(w), _p(planes), N, K, cin, _stream()), "dua_pack_weight_bf16x3")
    return planes


def pack_weight_f16x3(w):
    """fp32 [N][K] linear / 1x1 weight, or OIHW 3x3 conv weight -> the packed f16x3 weight (fp16 hi plane, fp16 lo plane of the
    row-scaled weight, per-row 2^-s floats) in one uint8 buffer."""
    _req(w, name="w")
    L = _lib.lib()
    if w.ndim == 4 and w.shape[2] == 3:
        N, cin = w.shape[0], w.shape[1]
        K = 9 * cin
    else:
        N, cin = w.shape[0], 0
        K = w.numel() // N
    packed = torch.empty(L.dua_f16x3_weight_bytes(N, K), device=w.device, dtype=torch.uint8)
    check(L.dua_pack_weight_f16x3(_p(w), _p(packed), N, K, cin, _stream()), "dua_pack_weight_f16x3")
    return packed


def pack_weight_f16x3_wino(w):
    """OIHW 3x3 conv weight -> the packed Winograd F(2,3)-along-y weight of dua_conv3x3_f16x3_wino (U_f * 2^s as fp16 hi | lo planes in
    the kernel's tile order, per-output-channel 2^-s floats) in one uint8 buffer."""
    _req(w, name="w")
    assert w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3
    L = _lib.lib()
    cout, cin = w.shape[0], w.shape[1]
    packed = torch.empty(L.dua_f16x3_wino_weight_bytes(cout, cin), device=w.device, dtype=torch.uint8)
    check(L.dua_pack_weight_f16x3_wino(_p(w), _p(packed), cout, cin, _stream()), "dua_pack_weight_f16x3_wino")
    return packed


def pack_weight_f16x3_wino4(w):
    """OIHW 3x3 conv weight -> the packed Winograd F(4,3)-along-y weight of dua_conv3x3_f16x3_wino4."""
    _req(w, name="w")
    assert w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3
    L = _lib.lib()
    cout, cin = w.shape[0], w.shape[1]
    packed = torch.empty(L.dua_f16x3_wino4_weight_bytes(cout, cin), device=w.device, dtype=torch.uint8)
    check(L.dua_pack_weight_f16x3_wino4(_p(w), _p(packed), cout, cin, _stream()), "dua_pack_weight_f16x3_wino4")
    return packed


def tap_column_weight(w, keep):
    """OIHW 3x3 conv weight -> the [N][Cin] matrix of its tap columns for conv3x3_small_cout: row (tap * keep + co) = w[co, :, tap]
    for the first `keep` output channels, zero rows up to a multiple of 32."""
    cout, cin = w.shape[0], w.shape[1]
    assert w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3 and 0 < keep <= cout
    rows = w[:keep].reshape(keep, cin, 9).permute(2, 0, 1).reshape(9 * keep, cin)
    n = (9 * keep + 31) // 32 * 32
    wt = torch.zeros((n, cin), device=w.device, dtype=torch.float32)
    wt[:9 * keep] = rows
    return wt.contiguous()


def conv3x3_tap_gather(T, bias, N, H, W, cout):
    """Second half of the small-Cout 3x3 conv: T [N*H*W, >= 9*cout] tap columns -> NCHW [N, cout, H, W] (+ bias)."""
    _req(T, name="T")
    out = torch.empty((N, cout, H, W), device=T.device, dtype=torch.float32)
    check(_lib.lib().dua_conv3x3_tap_gather(_p(T), T.shape[1], _p(bias), _p(out), N, H, W, cout, _stream()), "dua_conv3x3_tap_gather")
    return out


def pack_weight_f32_wino(w):
    """OIHW 3x3 conv weight -> the packed exact-fp32 Winograd weight of dua_conv3x3_f32_wino (U_f as floats in the kernel's stage order)."""
    _req(w, name="w")
    assert w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3
    L = _lib.lib()
    cout, cin = w.shape[0], w.shape[1]
    packed = torch.empty(L.dua_f32_wino_weight_bytes(cout, cin), device=w.device, dtype=torch.uint8)
    check(L.dua_pack_weight_f32_wino(_p(w), _p(packed), cout, cin, _stream()), "dua_pack_weight_f32_wino")
    return packed


def pack_weight_f32_wino4(w):
    """OIHW 3x3 conv weight -> the packed exact-fp32 F(4,3) weight of dua_conv3x3_f32_wino4."""
    _req(w, name="w")
    assert w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3
    L = _lib.lib()
    cout, cin = w.shape[0], w.shape[1]
    packed = torch.empty(L.dua_f32_wino4_weight_bytes(cout, cin), device=w.device, dtype=torch.uint8)
    check(L.dua_pack_weight_f32_wino4(_p(w), _p(packed), cout, cin, _stream()), "dua_pack_weight_f32_wino4")
    return packed


def conv3x3_wino_supported(N, H, W, cin, cout):
    """Shapes dua_conv3x3_f16x3_wino takes (H, W = the conv's input / output size)."""
    return bool(_lib.lib().dua_conv3x3_f16x3_wino_supported(int(N), int(H), int(W), int(cin), int(cout)))


def conv3x3_wino_pick(N, H, W, cin, cout):
    """0 / 2 / 4: the Winograd form (none, F(2,3), F(4,3)) dua_conv3x3_f16x3_wino_pick chooses for this shape."""
    return int(_lib.lib().dua_conv3x3_f16x3_wino_pick(int(N), int(H), int(W), int(cin), int(cout)))


def conv3x3_wino4_supported(N, H, W, cin, cout):
    """Shapes dua_conv3x3_f16x3_wino4 (F(4,3) along y) takes."""
    return bool(_lib.lib().dua_conv3x3_f16x3_wino4_supported(int(N), int(H), int(W), int(cin), int(cout)))


def conv3x3_bf16x3(x, planes, bias, cout, residual=None, nchw_out=False, colsum=False, res_resample=0):
    """Split-operand 3x3 convolution; a SplitPlanes input selects the kernel by its format (the weight must be packed to match).
    res_resample (Winograd formats only): +1 / -1 = `residual` is the half / double resolution map, added through nearest 2x upsampling /
    2x2 average pooling."""
    pre = isinstance(x, SplitPlanes)
    if not pre:
        _req(x, name="x")
    N, H, W, Cin = x.shape
    y = torch.empty((N, cout, H, W) if nchw_out else (N, H, W, cout), device=x.device, dtype=torch.float32)
    ws = _colsum_ws(N * H * W, cout, x.device, colsum and not nchw_out)
    assert res_resample == 0 or (pre and x.fmt in ("f16x3w", "f16x3w4", "f16w", "f32w", "f32w4") and residual is not None), "res_resample is a Winograd-conv option"
    if res_resample:
        rs = (N, H // 2, W // 2, cout) if res_resample > 0 else (N, 2 * H, 2 * W, cout)
        assert tuple(residual.shape) == rs and residual.is_contiguous() and residual.dtype == torch.float32, f"residual must be fp32 {rs}"
    e0 = e1 = None
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if pre and x.fmt == "f32w":
        assert not nchw_out, "the Winograd conv writes NHWC"
        if (H * W) % 128:
            ws = None
        check(_lib.lib().dua_conv3x3_f32_wino(_p(x.buf), _p(planes), _p(bias), _p(residual), int(res_resample), _p(y), N, H, W, Cin, cout, _p(ws), _stream()),
              "dua_conv3x3_f32_wino")
    elif pre and x.fmt == "f32w4":
        assert not nchw_out, "the Winograd conv writes NHWC"
        if (H * W) % 256:
            ws = None
        check(_lib.lib().dua_conv3x3_f32_wino4(_p(x.buf), _p(planes), _p(bias), _p(residual), int(res_resample), _p(y), N, H, W, Cin, cout, _p(ws), _stream()),
              "dua_conv3x3_f32_wino4")
    elif pre and x.fmt == "f16x3w4":
        assert not nchw_out, "the Winograd conv writes NHWC"
        if (H * W) % 256:
            ws = None                                  # small maps: no column sums from the F(4,3) kernel
        check(_lib.lib().dua_conv3x3_f16x3_wino4(_p(x.buf), _p(planes), _p(bias), _p(residual), int(res_resample), _p(y), N, H, W, Cin, cout, _p(ws),
                                                 x.scale, _stream()), "dua_conv3x3_f16x3_wino4")
    elif pre and x.fmt in ("f16x3w", "f16w"):
        assert not nchw_out, "the Winograd conv writes NHWC"
        if (H * W) % 128:
            ws = None                                  # small maps: no column sums from the Winograd kernel
        fn = _lib.lib().dua_conv3x3_f16w_wino if x.fmt == "f16w" else _lib.lib().dua_conv3x3_f16x3_wino
        check(fn(_p(x.buf), _p(planes), _p(bias), _p(residual), int(res_resample), _p(y), N, H, W, Cin, cout, _p(ws), x.scale, _stream()),
              "dua_conv3x3_f16x3_wino")
    elif pre and x.fmt == "f16x3":
        check(_lib.lib().dua_conv3x3_f16x3_presplit(_p(x.buf), _p(planes), _p(bias), _p(residual), _p(y), N, H, W, Cin, cout, int(nchw_out),
                                                    _p(ws), x.scale, _stream()), "dua_conv3x3_f16x3_presplit")
    elif pre:
        fn = _lib.lib().dua_conv3x3_bf16x3_presplit
        check(fn(_p(x.buf), _p(planes), _p(bias), _p(residual), _p(y), N, H, W, Cin, cout, int(nchw_out), _p(ws), _stream()),
              "dua_conv3x3 presplit")
    else:
        check(_lib.lib().dua_conv3x3_bf16x3(_p(x), _p(planes), _p(bias), _p(residual), _p(y), N, H, W, Cin, cout, int(nchw_out),
                                            _p(ws), _stream()), "dua_conv3x3_bf16x3")
    if PROFILE is not None:
        e1.record()
        PROFILE.record("conv3x3_wino" if (pre and x.fmt in ("f16x3w", "f16x3w4", "f16w", "f32w", "f32w4")) else "conv3x3_bf16x3", 2.0 * N * H * W * 9 * Cin * cout, e0, e1)
    return _attach_colsums(y, ws, H * W)


def gemm_bf16x3(a1, planes, n_out, bias=None, a2=None, residual=None, act_in=0, act_out=0, alpha=1.0, colsum=False):
    _req(a1, name="a1")
    M, K1 = a1.shape
    K2 = 0 if a2 is None else a2.shape[1]
    c = torch.empty((M, n_out), device=a1.device, dtype=torch.float32)
    ws = _colsum_ws(M, n_out, a1.device, colsum)
    if ws is not None:
        c._dua_colsums_rows = ws
    check(_lib.lib().dua_gemm_bf16x3(_p(a1), K1, K1, _p(a2), K2, max(K2, 4), _p(planes), _p(bias), _p(residual), n_out, _p(c), n_out,
                                     M, n_out, int(act_in), int(act_out), float(alpha), _p(ws), _stream()), "dua_gemm_bf16x3")
    return c


def gemm_f16x3(a1, packed, n_out, bias=None, a2=None, residual=None, act_in=0, act_out=0, alpha=1.0, colsum=False, act_scale=None):
    """gemm_bf16x3 on the fp32-equivalent split (fp32 A scaled and split into fp16 hi + lo on the fly; pack_weight_f16x3 weights)."""
    _req(a1, name="a1")
    M, K1 = a1.shape
    K2 = 0 if a2 is None else a2.shape[1]
    c = torch.empty((M, n_out), device=a1.device, dtype=torch.float32)
    ws = _colsum_ws(M, n_out, a1.device, colsum)
    if ws is not None:
        c._dua_colsums_rows = ws
    check(_lib.lib().dua_gemm_f16x3(_p(a1), K1, K1, _p(a2), K2, max(K2, 4), _p(packed), _p(bias), _p(residual), n_out, _p(c), n_out,
                                    M, n_out, int(act_in), int(act_out), float(alpha), _p(ws),
                                    F16X3_RAW_SCALE if act_scale is None else float(act_scale), _stream()), "dua_gemm_f16x3")
    return c


def gemm_bf16x3_presplit(a, planes, n_out, bias=None, residual=None, act_out=0, split_out=False, colsum=False):
    """a: SplitPlanes whose trailing dim is K (rows = all leading dims) -> [rows, n_out] fp32, or SplitPlanes if split_out."""
    K = a.shape[-1]
    M = 1
    for d in a.shape[:-1]:
        M *= d
    if a.fmt == "f16x3" and split_out:
        assert residual is None and n_out % 32 == 0
        buf = torch.empty((2, M, n_out), device=a.device, dtype=torch.int16)
        check(_lib.lib().dua_gemm_f16x3_presplit_planes(_p(a.buf), _p(planes), _p(bias), _p(buf), M, n_out, K, int(act_out), a.scale,
                                                        F16X3_RAW_SCALE, _stream()), "dua_gemm_f16x3_presplit_planes")
        return SplitPlanes(buf, (M, n_out), "f16x3", F16X3_RAW_SCALE)
    if a.fmt == "f16x3":
        c = torch.empty((M, n_out), device=a.device, dtype=torch.float32)
        ws = _colsum_ws(M, n_out, a.device, colsum)
        if ws is not None:
            c._dua_colsums_rows = ws
        check(_lib.lib().dua_gemm_f16x3_presplit(_p(a.buf), _p(planes), _p(bias), _p(residual), _p(c), M, n_out, K, int(act_out), _p(ws),
                                                 a.scale, _stream()), "dua_gemm_f16x3_presplit")
        return c
    if split_out:
        buf = torch.empty((2, M, n_out), device=a.device, dtype=torch.int16)
        check(_lib.lib().dua_gemm_bf16x3_presplit(_p(a.buf), _p(planes), _p(bias), _p(residual), None, _p(buf), M, n_out, K, int(act_out),
                                                  None, _stream()), "dua_gemm_bf16x3_presplit")
        return SplitPlanes(buf, (M, n_out))
    c = torch.empty((M, n_out), device=a.device, dtype=torch.float32)
    check(_lib.lib().dua_gemm_bf16x3_presplit(_p(a.buf), _p(planes), _p(bias), _p(residual), _p(c), None, M, n_out, K, int(act_out), None,
                                              _stream()), "dua_gemm_bf16x3_presplit")
    return c


def gemm_f16x3_presplit2(a1, a2, packed, n_out, bias=None, residual=None, act_out=0, colsum=False, planes_out=False, extra_from=0,
                         extra_only=False):
    """[a1 | a2] @ W^T with both operands given as f16x3 SplitPlanes (a2 may be None): the 1x1 skip convolution / U-ViT skip_linear on
    a virtual concatenation.  planes_out: additionally return the f16x3 planes of the fp32 result (scale F16X3_RAW_SCALE)."""
    assert a1.fmt == "f16x3" and (a2 is None or (a2.fmt == "f16x3" and a2.scale == a1.scale))
    K1 = a1.shape[-1]
    K2 = 0 if a2 is None else a2.shape[-1]
    M = 1
    for d in a1.shape[:-1]:
        M *= d
    c = torch.empty((M, n_out), device=a1.device, dtype=torch.float32)
    ws = _colsum_ws(M, n_out, a1.device, colsum)
    if ws is not None:
        c._dua_colsums_rows = ws
    xp = torch.empty((2, M, n_out), device=a1.device, dtype=torch.int16) if planes_out else None
    check(_lib.lib().dua_gemm_f16x3_presplit2(_p(a1.buf), K1, _p(a2.buf) if a2 is not None else None, K2, _p(packed), _p(bias), _p(residual),
                                              _p(c), _p(xp), int(extra_from), int(bool(extra_only)), M, n_out, int(act_out), _p(ws), a1.scale,
                                              F16X3_RAW_SCALE, _stream()),
          "dua_gemm_f16x3_presplit2")
    return (c, SplitPlanes(xp, (M, n_out), "f16x3", F16X3_RAW_SCALE)) if planes_out else c


def gemm_raw(**kw):
    g = GemmArgs(**kw)
    check(_lib.lib().dua_gemm(ctypes.byref(g), _stream()), "dua_gemm")


def softmax_rows_(S, rows, cols, ld=None):
    check(_lib.lib().dua_softmax_rows(_p(S), int(rows), int(cols), int(cols if ld is None else ld), _stream()), "dua_softmax_rows")
    return S


# ---------------------------------------------------------------------------------------------------------------------
# Input gradients (gradient-mode guidance, uncertainty_guidance.py:83-108,121-129): activations only, exact fp32.

def groupnorm_bwd(x1, stats, gamma, beta, dy, x2=None, act=1, groups=32, add1=None, add2=None):
    """Gradient of act(GroupNorm(cat(x1, x2))) w.r.t. x1 (and x2), given dy [N, H, W, C1+C2] and the forward statistics."""
    for t, nm in ((x1, "x1"), (stats, "stats"), (dy, "dy")):
        _req(t, name=nm)
    N, H, W, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    assert dy.shape == (N, H, W, C1 + C2) and stats.shape == (N, groups, 2)
    dx1 = torch.empty_like(x1)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    check(_lib.lib().dua_groupnorm_bwd(_p(x1), C1, _p(x2), C2, N, H * W, groups, _p(stats), _p(gamma), _p(beta), int(act), _p(dy),
                                       _p(add1), _p(add2), _p(dx1), _p(dx2), _stream()), "dua_groupnorm_bwd")
    return dx1, dx2


def layernorm_bwd(x, gamma, dy, eps=1e-5, add=None):
    _req(x, name="x"), _req(dy, name="dy")
    D = x.shape[-1]
    assert dy.shape == x.shape
    dx = torch.empty_like(x)
    check(_lib.lib().dua_layernorm_bwd(_p(x), _p(gamma), _p(dy), _p(add), _p(dx), x.numel() // D, D, float(eps), _stream()), "dua_layernorm_bwd")
    return dx


def geglu_bwd(x, dy):
    _req(x, name="x"), _req(dy, name="dy")
    M, F2 = x.shape
    assert dy.shape == (M, F2 // 2)
    dx = torch.empty_like(x)
    check(_lib.lib().dua_geglu_bwd(_p(x), _p(dy), _p(dx), M, F2 // 2, _stream()), "dua_geglu_bwd")
    return dx


def softmax_bwd_rows_(P, dP, rows, cols, ld=None):
    check(_lib.lib().dua_softmax_bwd_rows(_p(P), _p(dP), int(rows), int(cols), int(cols if ld is None else ld), _stream()), "dua_softmax_bwd_rows")
    return dP


def place_even(g, H, W):
    """Adjoint of pick_even: [N, H/2, W/2, C] -> [N, H, W, C] with zeros off the even pixels."""
    _req(g, name="g")
    N, Ho, Wo, C = g.shape
    assert (Ho, Wo) == (H // 2, W // 2)
    out = torch.empty((N, H, W, C), device=g.device, dtype=torch.float32)
    check(_lib.lib().dua_place_even(_p(g), _p(out), N, H, W, C, _stream()), "dua_place_even")
    return out


def mc_var_grad(stack, scale):
    """d/d e_m of (scale/2) * sum_m (e_m - mean_m e)^2 for the stacked samples [M, ...]."""
    _req(stack, name="stack")
    G = torch.empty_like(stack)
    check(_lib.lib().dua_mc_var_grad(_p(stack), _p(G), stack.shape[0], stack[0].numel(), float(scale), _stream()), "dua_mc_var_grad")
    return G


# logits / probabilities of a backward attention chunk are materialised in HBM; chunks of samples bound them to this many floats
ATTN_BWD_CHUNK_FLOATS = 1 << 28


def _attn_bwd_chunk(qp, kp, vp, dop, dqp, dkp, dvp, ldq, ldk, ldo, ldg, nb, heads, Tq, Tk, ch, dev):
    """One chunk of nb samples.  q / k / v / dO / dq / dk / dv are base addresses of [nb*T, ld] row-major operands whose head h
    sits at column offset h*ch; dkp / dvp = 0: keys and values are constants (cross-attention).  The two [Tq, Tk] matrices
    (P and dP -> dS) are written once by their GEMMs, rewritten once by the fused softmax pass and read once by each gradient
    GEMM; dk / dv are formed transposed ([ch, Tk] = q^T dS, dO^T P: K-major reads of the big matrices as they lie) so that
    neither matrix is ever transposed - only the [T, ch] operands are."""
    BH = nb * heads
    Tp = (Tk + 3) // 4 * 4
    alpha = 1.0 / math.sqrt(ch)
    P = torch.empty((BH, Tq, Tp), device=dev, dtype=torch.float32)
    dP = torch.empty((BH, Tq, Tp), device=dev, dtype=torch.float32)
    common = dict(batch=BH, batch_inner=heads)
    L = _lib.lib()
    # logits alpha q k^T and dP = dO v^T, then P = softmax(logits), dS in place of dP
    gemm_raw(A1=qp, B=kp, C=P.data_ptr(), M=Tq, N=Tk, K1=ch, K2=0, lda1=ldq, lda2=4, ldb=ldk, ldc=Tp, ldr=Tp, sA1=Tq * ldq, sB=Tk * ldk,
             sC=heads * Tq * Tp, sA1i=ch, sBi=ch, sCi=Tq * Tp, transB=1, alpha=alpha, **common)
    gemm_raw(A1=dop, B=vp, C=dP.data_ptr(), M=Tq, N=Tk, K1=ch, K2=0, lda1=ldo, lda2=4, ldb=ldk, ldc=Tp, ldr=Tp, sA1=Tq * ldo, sB=Tk * ldk,
             sC=heads * Tq * Tp, sA1i=ch, sBi=ch, sCi=Tq * Tp, transB=1, alpha=1.0, **common)
    check(L.dua_softmax_fwd_bwd_rows(_p(P), _p(dP), BH * Tq, Tk, Tp, _stream()), "dua_softmax_fwd_bwd_rows")
    # dq = alpha dS k
    gemm_raw(A1=dP.data_ptr(), B=kp, C=dqp, M=Tq, N=ch, K1=Tk, K2=0, lda1=Tp, lda2=4, ldb=ldk, ldc=ldg, ldr=ldg, sA1=heads * Tq * Tp,
             sB=Tk * ldk, sC=Tq * ldg, sA1i=Tq * Tp, sBi=ch, sCi=ch, transB=0, alpha=alpha, **common)
    if not dkp:
        return
    Tqp = (Tq + 3) // 4 * 4
    small = torch.empty((2, BH, ch, Tqp), device=dev, dtype=torch.float32)          # q^T | dO^T
    outT = torch.empty((2, BH, ch, Tp), device=dev, dtype=torch.float32)            # dk^T | dv^T
    for src, ld, dst in ((qp, ldq, small[0]), (dop, ldo, small[1])):
        check(L.dua_transpose_batched(src, _p(dst), Tq, ch, ld, Tqp, Tq * ld, ch, heads * ch * Tqp, ch * Tqp, BH, heads, _stream()),
              "dua_transpose_batched")
    for a_t, big, o_t, al in ((small[0], dP, outT[0], alpha), (small[1], P, outT[1], 1.0)):
        gemm_raw(A1=a_t.data_ptr(), B=big.data_ptr(), C=o_t.data_ptr(), M=ch, N=Tp, K1=Tq, K2=0, lda1=Tqp, lda2=4, ldb=Tp, ldc=Tp, ldr=Tp,
                 sA1=heads * ch * Tqp, sB=heads * Tq * Tp, sC=heads * ch * Tp, sA1i=ch * Tqp, sBi=Tq * Tp, sCi=ch * Tp, transB=0,
                 alpha=al, **common)
    for o_t, dst in ((outT[0], dkp), (outT[1], dvp)):
        check(L.dua_transpose_batched(_p(o_t), dst, ch, Tk, Tp, ldg, heads * ch * Tp, ch * Tp, Tk * ldg, ch, BH, heads, _stream()),
              "dua_transpose_batched")


def qkv_attention_bwd(qkv, dout, B, T, heads, ch):
    """Gradient of qkv_attention w.r.t. the packed qkv rows [B*T, 3*heads*ch], given dout [B*T, heads*ch]: the probabilities are
    recomputed from qkv (fp32 MFMA GEMMs, logits in HBM, a chunk of samples at a time)."""
    _req(qkv, name="qkv"), _req(dout, name="dout")
    C = heads * ch
    assert qkv.shape == (B * T, 3 * C) and dout.shape == (B * T, C) and T <= 4096
    dqkv = torch.empty_like(qkv)
    nb = max(1, min(B, ATTN_BWD_CHUNK_FLOATS // (heads * T * ((T + 3) // 4 * 4))))
    for b0 in range(0, B, nb):
        n = min(nb, B - b0)
        qb, gb, ob = qkv.data_ptr() + 4 * b0 * T * 3 * C, dqkv.data_ptr() + 4 * b0 * T * 3 * C, dout.data_ptr() + 4 * b0 * T * C
        _attn_bwd_chunk(qb, qb + 4 * C, qb + 8 * C, ob, gb, gb + 4 * C, gb + 8 * C, 3 * C, 3 * C, C, 3 * C, n, heads, T, T, ch, qkv.device)
    return dqkv


def attention_bwd_q(q, k, v, dout, heads):
    """Gradient of attention(q, k, v, heads) w.r.t. q only (cross-attention: k / v come from the constant text context)."""
    for t, nm in ((q, "q"), (k, "k"), (v, "v"), (dout, "dout")):
        _req(t, name=nm)
    B, Tq, C = q.shape
    Tk = k.shape[1]
    ch = C // heads
    assert dout.shape == q.shape and Tk <= 4096
    dq = torch.empty_like(q)
    nb = max(1, min(B, ATTN_BWD_CHUNK_FLOATS // (heads * Tq * ((Tk + 3) // 4 * 4))))
    for b0 in range(0, B, nb):
        n = min(nb, B - b0)
        _attn_bwd_chunk(q.data_ptr() + 4 * b0 * Tq * C, k.data_ptr() + 4 * b0 * Tk * C, v.data_ptr() + 4 * b0 * Tk * C,
                        dout.data_ptr() + 4 * b0 * Tq * C, dq.data_ptr() + 4 * b0 * Tq * C, 0, 0, C, C, C, C, n, heads, Tq, Tk, ch, q.device)
    return dq


# "auto": fused kernel wherever it applies except where the split-bf16 GEMM pair is faster (see qkv_attention);
# "fused" / "gemm" force one path (A/B measurements, tests).
ATTENTION_MODE = "auto"


def fused_attention(qkv, B, T, heads, ch):
    """Flash-style fp32 attention in one kernel (dua_attention): same contract as qkv_attention."""
    _req(qkv, name="qkv")
    C = heads * ch
    assert qkv.shape == (B * T, 3 * C)
    out = torch.empty((B * T, C), device=qkv.device, dtype=torch.float32)
    check(_lib.lib().dua_attention(_p(qkv), _p(out), B, T, heads, ch, 1.0 / math.sqrt(ch), _stream()), "dua_attention")
    return out


def fused_attention_bf16x3(qkv, B, T, heads, ch):
    """Flash-style attention with both products on the bf16 matrix cores, split operands (dua_attention_bf16x3)."""
    _req(qkv, name="qkv")
    C = heads * ch
    assert qkv.shape == (B * T, 3 * C)
    out = torch.empty((B * T, C), device=qkv.device, dtype=torch.float32)
    check(_lib.lib().dua_attention_bf16x3(_p(qkv), _p(out), B, T, heads, ch, 1.0 / math.sqrt(ch), _stream()), "dua_attention_bf16x3")
    return out


def qkv_attention(qkv, B, T, heads, ch, precision="fp32", planes_out=False):
    """qkv: [B*T, 3*heads*ch] laid out (q | k | v), head-major inside each third -> [B*T, heads*ch].

    Reference: QKVAttentionLegacy / QKVAttention (unet_openai.py:338-355 / :371-390); the q/k scale
    ch^-1/4 on both operands is applied as alpha = ch^-1/2 on the logits.  Default path: the fused flash-style fp32
    kernel (dua_attention), logits never leave the CU.  GEMM path (head dims > 256, or wide heads in bf16x3 mode): the
    logits are materialised in HBM, S = q.k^T (MFMA tile engine), softmax in dua_softmax_rows, V is transposed to
    [head channel][token] so that O = P.(V^T)^T also runs through the K-major GEMM path.  precision="bf16x3" evaluates both products on
    the bf16 matrix cores with both operands split on the fly (needs ch % 32 == 0 and T % 32 == 0); softmax is fp32.
    """
    _req(qkv, name="qkv")
    C = heads * ch
    assert qkv.shape == (B * T, 3 * C)
    assert B * heads <= 65535
    if precision == "f16x3":
        if ch % 4 == 0 and 8 <= ch <= 256 and ATTENTION_MODE in ("auto", "fused_x"):      # the split flash kernel on fp16 hi / lo halves
            if planes_out and C % 32 == 0:
                buf = torch.empty((2, B * T, C), device=qkv.device, dtype=torch.int16)
                check(_lib.lib().dua_attention_f16x3(_p(qkv), None, _p(buf), F16X3_RAW_SCALE, B, T, heads, ch, 1.0 / math.sqrt(ch), _stream()),
                      "dua_attention_f16x3")
                return SplitPlanes(buf, (B * T, C), "f16x3", F16X3_RAW_SCALE)
            out = torch.empty((B * T, C), device=qkv.device, dtype=torch.float32)
            check(_lib.lib().dua_attention_f16x3(_p(qkv), _p(out), None, 0.0, B, T, heads, ch, 1.0 / math.sqrt(ch), _stream()),
                  "dua_attention_f16x3")
            return out
        precision = "fp32"                       # heads wider than 256 / forced GEMM mode: the exact fp32 paths
    if precision == "bf16x3" and ch % 4 == 0 and 8 <= ch <= 128 and ATTENTION_MODE in ("auto", "fused_x"):
        return fused_attention_bf16x3(qkv, B, T, heads, ch)       # split-bf16 flash kernel: 2-3x the GEMM pair / the fp32 flash kernel
    fast = precision == "bf16x3" and ch % 32 == 0 and T % 32 == 0
    fusable = ch % 8 == 0 and ch <= 256
    # measured (MI355X, B=128): fused fp32 117 TF vs 64 TF for the split-bf16 GEMM pair at ch=64, T=1024; at ch=128 the
    # GEMM pair is ~5 % ahead (105 vs 100 TF), and heads wider than 128 leave the fused kernel no registers for its K/V
    # prefetch (ch=192, T=256: 39 TF fused vs 65 / 89 TF for the fp32 / split-bf16 GEMM pair), so "auto" keeps those.
    if fusable and (ATTENTION_MODE == "fused" or (ATTENTION_MODE == "auto" and ch <= 128 and not (fast and ch > 96))):
        return fused_attention(qkv, B, T, heads, ch)
    Tp = (T + 3) // 4 * 4            # row stride of the logits / of V^T (16-byte aligned rows)
    S = torch.empty((B * heads, T, Tp), device=qkv.device, dtype=torch.float32)
    Vt = torch.empty((B * heads, ch, Tp), device=qkv.device, dtype=torch.float32)
    out = torch.empty((B * T, C), device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    L = _lib.lib()
    s = 1.0 / math.sqrt(math.sqrt(ch))
    check(L.dua_transpose_batched(base + 8 * C, _p(Vt), T, ch, 3 * C, Tp, T * 3 * C, ch, heads * ch * Tp, ch * Tp, B * heads, heads,
                                  _stream()), "dua_transpose_batched")
    if fast:
        check(L.dua_gemm_nt_bf16x3_batched(base, 3 * C, T * 3 * C, ch, base + 4 * C, 3 * C, T * 3 * C, ch, _p(S), Tp, heads * T * Tp, T * Tp,
                                           T, T, ch, B * heads, heads, s * s, _stream()), "dua_gemm_nt_bf16x3_batched")
    else:
        gemm_raw(A1=base, B=base + 4 * C, C=S.data_ptr(), M=T, N=T, K1=ch, K2=0, lda1=3 * C, lda2=4, ldb=3 * C, ldc=Tp, ldr=Tp,
                 sA1=T * 3 * C, sB=T * 3 * C, sC=heads * T * Tp, batch=B * heads, batch_inner=heads, sA1i=ch, sBi=ch,
                 sCi=T * Tp, transB=1, alpha=s * s)
    softmax_rows_(S, B * heads * T, T, Tp)
    if fast:
        check(L.dua_gemm_nt_bf16x3_batched(_p(S), Tp, heads * T * Tp, T * Tp, _p(Vt), Tp, heads * ch * Tp, ch * Tp, _p(out), C, T * C, ch,
                                           T, ch, T, B * heads, heads, 1.0, _stream()), "dua_gemm_nt_bf16x3_batched")
    else:
        gemm_raw(A1=S.data_ptr(), B=Vt.data_ptr(), C=out.data_ptr(), M=T, N=ch, K1=T, K2=0, lda1=Tp, lda2=4, ldb=Tp, ldc=C, ldr=C,
                 sA1=heads * T * Tp, sB=heads * ch * Tp, sC=T * C, batch=B * heads, batch_inner=heads, sA1i=T * Tp, sBi=ch * Tp,
                 sCi=ch, transB=1, alpha=1.0)
    return out


def layernorm(x, gamma, beta, eps=1e-5, split=False):
    _req(x, name="x")
    D = x.shape[-1]
    if split == "f16x3":
        buf = torch.empty((2,) + tuple(x.shape), device=x.device, dtype=torch.int16)
        check(_lib.lib().dua_layernorm_split_f16x3(_p(x), _p(gamma), _p(beta), _p(buf), x.numel() // D, D, float(eps), F16X3_ACT_SCALE, _stream()),
              "dua_layernorm_split_f16x3")
        return SplitPlanes(buf, tuple(x.shape), "f16x3", F16X3_ACT_SCALE)
    if split:
        buf = torch.empty((2,) + tuple(x.shape), device=x.device, dtype=torch.int16)
        check(_lib.lib().dua_layernorm_split(_p(x), _p(gamma), _p(beta), _p(buf), x.numel() // D, D, float(eps), _stream()), "dua_layernorm_split")
        return SplitPlanes(buf, tuple(x.shape), "bf16x3")
    y = torch.empty_like(x)
    check(_lib.lib().dua_layernorm(_p(x), _p(gamma), _p(beta), _p(y), x.numel() // D, D, float(eps), _stream()), "dua_layernorm")
    return y


def timestep_embedding(t, dim):
    _req(t, torch.int64, "timesteps")
    emb = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    check(_lib.lib().dua_timestep_embedding(_p(t), _p(emb), t.shape[0], dim, _stream()), "dua_timestep_embedding")
    return emb


def embedding_add(a, table, idx):
    _req(table, name="table"), _req(idx, torch.int64, "idx")
    N, D = idx.shape[0], table.shape[1]
    out = torch.empty((N, D), device=table.device, dtype=torch.float32)
    check(_lib.lib().dua_embedding_add(_p(a), _p(table), _p(idx), _p(out), N, D, _stream()), "dua_embedding_add")
    return out


# ------------------------------------------------------------------ U-ViT token plumbing
def patchify(x, P):
    _req(x, name="x")
    B, C, H, W = x.shape
    rows = torch.empty((B * (H // P) * (W // P), C * P * P), device=x.device, dtype=torch.float32)
    check(_lib.lib().dua_patchify(_p(x), _p(rows), B, C, H, W, P, _stream()), "dua_patchify")
    return rows


def uvit_assemble(patches, time_tok, label_table, y, pos_embed, B, L, D):
    extras = 2 if y is not None else 1
    out = torch.empty((B, L + extras, D), device=patches.device, dtype=torch.float32)
    check(_lib.lib().dua_uvit_assemble(_p(patches), _p(time_tok), _p(label_table), _p(y), _p(pos_embed), _p(out), B, L, D,
                                       _stream()), "dua_uvit_assemble")
    return out


def unpatchify(tok, B, L, extras, C, P):
    g = int(round(math.sqrt(L)))
    img = torch.empty((B, C, g * P, g * P), device=tok.device, dtype=torch.float32)
    check(_lib.lib().dua_unpatchify(_p(tok), _p(img), B, L, extras, C, P, _stream()), "dua_unpatchify")
    return img


# ------------------------------------------------------------------ block-level entry points
PRECISION_CODE = {"fp32": 0, "f16x3": 3, "f16w": 4}          # "f16w": dua_adm_resblock_fwd only (DUA_PREC_F16W)


def _cs(t):
    """Producer column sums attached to an NHWC tensor (or None)."""
    return getattr(t, "_dua_colsums", None) if t is not None else None


def adm_resblock(x, skip, gn1, conv1_packed, conv1_b, film, gn2, conv2_packed, conv2_b, skip_packed, skip_b, cout, updown=0,
                 dropmask=None, precision="fp32", want_colsum=True, conv1_wino=None, conv2_wino=None, conv1_wino_m=2, conv2_wino_m=2):
    """One ADM ResBlock forward through dua_adm_resblock_fwd (guided_diffusion/unet_openai.py:237-257).  x / skip NHWC fp32,
    gn1 / gn2 = (weight, bias) pairs, weights packed for `precision`; returns y NHWC with its column sums attached."""
    _req(x, name="x")
    N, H, W, C1 = x.shape
    C2 = 0 if skip is None else skip.shape[3]
    Ho, Wo = (H // 2, W // 2) if updown < 0 else ((2 * H, 2 * W) if updown > 0 else (H, W))
    y = torch.empty((N, Ho, Wo, cout), device=x.device, dtype=torch.float32)
    ws_y = _colsum_ws(N * Ho * Wo, cout, x.device, want_colsum and not (conv2_wino is not None and (Ho * Wo) % (256 if conv2_wino_m == 4 else 128)))
    a = ResBlockArgs(x1=x.data_ptr(), C1=C1, x2=(skip.data_ptr() if skip is not None else None), C2=C2, N=N, H=H, W=W, Cout=cout,
                     gn1_w=gn1[0].data_ptr(), gn1_b=gn1[1].data_ptr(), conv1_packed=(conv1_packed.data_ptr() if conv1_packed is not None else None), conv1_b=conv1_b.data_ptr(),
                     film=film.data_ptr(), film_ld=_film_ld(film, N, cout), gn2_w=gn2[0].data_ptr(), gn2_b=gn2[1].data_ptr(),
                     conv2_packed=(conv2_packed.data_ptr() if conv2_packed is not None else None), conv2_b=conv2_b.data_ptr(),
                     skip_packed=(skip_packed.data_ptr() if skip_packed is not None else None),
                     skip_b=(skip_b.data_ptr() if skip_b is not None else None),
                     x1_colsums=(_cs(x).data_ptr() if _cs(x) is not None else None),
                     x2_colsums=(_cs(skip).data_ptr() if _cs(skip) is not None else None), updown=int(updown),
                     dropmask=(dropmask.data_ptr() if dropmask is not None else None), precision=PRECISION_CODE[precision],
                     act_scale=F16X3_ACT_SCALE, raw_scale=F16X3_RAW_SCALE, workspace=None, workspace_bytes=0, y=y.data_ptr(),
                     y_colsums=(ws_y.data_ptr() if ws_y is not None else None),
                     conv1_wino=(conv1_wino.data_ptr() if conv1_wino is not None else None),
                     conv2_wino=(conv2_wino.data_ptr() if conv2_wino is not None else None),
                     conv1_wino_m=int(conv1_wino_m), conv2_wino_m=int(conv2_wino_m))
    L = _lib.lib()
    nbytes = L.dua_adm_resblock_workspace_bytes(ctypes.byref(a))
    if nbytes == 0:
        raise RuntimeError(f"dua_adm_resblock_workspace_bytes: {L.dua_last_error().decode()}")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(L.dua_adm_resblock_fwd(ctypes.byref(a), _stream()), "dua_adm_resblock_fwd")
    return _attach_colsums(y, ws_y, Ho * Wo)


def adm_attnblock(x, heads, gn, qkv_packed, qkv_b, proj_packed, proj_b, precision="fp32", want_colsum=True):
    """One ADM AttentionBlock forward through dua_adm_attnblock_fwd (unet_openai.py:300-306), head dims <= 128 (f16x3: <= 256)."""
    _req(x, name="x")
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    ws_y = _colsum_ws(N * H * W, C, x.device, want_colsum)
    a = AttnBlockArgs(x=x.data_ptr(), C=C, N=N, H=H, W=W, heads=int(heads), gn_w=gn[0].data_ptr(), gn_b=gn[1].data_ptr(),
                      qkv_packed=qkv_packed.data_ptr(), qkv_b=qkv_b.data_ptr(), proj_packed=proj_packed.data_ptr(), proj_b=proj_b.data_ptr(),
                      x_colsums=(_cs(x).data_ptr() if _cs(x) is not None else None), precision=PRECISION_CODE[precision],
                      act_scale=F16X3_ACT_SCALE, raw_scale=F16X3_RAW_SCALE, workspace=None, workspace_bytes=0, y=y.data_ptr(),
                      y_colsums=(ws_y.data_ptr() if ws_y is not None else None))
    L = _lib.lib()
    nbytes = L.dua_adm_attnblock_workspace_bytes(ctypes.byref(a))
    if nbytes == 0:
        raise RuntimeError(f"dua_adm_attnblock_workspace_bytes: {L.dua_last_error().decode()}")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(L.dua_adm_attnblock_fwd(ctypes.byref(a), _stream()), "dua_adm_attnblock_fwd")
    return _attach_colsums(y, ws_y, H * W)


def uvit_block(x, skip, B, L, heads, w, precision="fp32", eps=1e-5, want_planes=False):
    """One U-ViT block through dua_uvit_block_fwd (uvit/uvit.py:95-120).  x / skip: [B*L, D] fp32; w: dict of device tensors
    (skip_w, skip_b | None, norm1_w, norm1_b, norm2_w, norm2_b, qkv_w, qkv_b | None, proj_w, proj_b, fc1_w, fc1_b, fc2_w, fc2_b),
    the *_w entries packed for `precision`.  f16x3: tensors produced with want_planes carry their f16x3 planes (`_dua_planes`); when both x and
    skip carry them, skip_linear runs on the presplit two-source GEMM."""
    _req(x, name="x")
    M, D = x.shape
    hidden = int(w["hidden"])
    y = torch.empty_like(x)
    f16 = precision == "f16x3"
    xp = getattr(x, "_dua_planes", None) if (f16 and skip is not None) else None
    sp = getattr(skip, "_dua_planes", None) if (f16 and skip is not None) else None
    if xp is None or sp is None:
        xp = sp = None
    yp = torch.empty((2, M, D), device=x.device, dtype=torch.int16) if (f16 and want_planes) else None
    ptr = lambda t: (t.data_ptr() if t is not None else None)     # noqa: E731
    a = UViTBlockArgs(x=x.data_ptr(), skip=ptr(skip), B=B, L=L, D=D, heads=int(heads), hidden=hidden,
                      skip_packed=ptr(w.get("skip_w")) if skip is not None else None, skip_b=ptr(w.get("skip_b")) if skip is not None else None,
                      norm1_w=ptr(w["norm1_w"]), norm1_b=ptr(w["norm1_b"]), norm2_w=ptr(w["norm2_w"]), norm2_b=ptr(w["norm2_b"]), eps=float(eps),
                      qkv_packed=ptr(w["qkv_w"]), qkv_b=ptr(w.get("qkv_b")), proj_packed=ptr(w["proj_w"]), proj_b=ptr(w.get("proj_b")),
                      fc1_packed=ptr(w["fc1_w"]), fc1_b=ptr(w.get("fc1_b")), fc2_packed=ptr(w["fc2_w"]), fc2_b=ptr(w.get("fc2_b")),
                      precision=PRECISION_CODE[precision], act_scale=F16X3_ACT_SCALE, raw_scale=F16X3_RAW_SCALE,
                      workspace=None, workspace_bytes=0, y=y.data_ptr(), x_planes=ptr(xp), skip_planes=ptr(sp), y_planes=ptr(yp))
    Lb = _lib.lib()
    nbytes = Lb.dua_uvit_block_workspace_bytes(ctypes.byref(a))
    if nbytes == 0:
        raise RuntimeError(f"dua_uvit_block_workspace_bytes: {Lb.dua_last_error().decode()}")
    ws = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
    a.workspace, a.workspace_bytes = ws.data_ptr(), nbytes
    check(Lb.dua_uvit_block_fwd(ctypes.byref(a), _stream()), "dua_uvit_block_fwd")
    if yp is not None:
        y._dua_planes = yp
    return y
