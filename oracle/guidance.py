"""Quantile mask and percentile-thresholded posterior score update (oracle; test infrastructure).

Follows reference ``uncertainty_guidance.py``
``get_uncertainty_guided_score_with_percentile`` :61-131 (posterior branch
:101-120) and the ``torch.quantile`` 'linear' semantics the reference relies on
at :112 (fp32 rank arithmetic).  numpy only.
"""
import numpy as np


def quantile_linear(u_flat, p):
    """torch.quantile(u, p, dim=1) for fp32 rows, bit-exact (a14).

    rank = float32(p) * (n - 1) in fp32; q = s[lo] + (s[hi] - s[lo]) * w in fp32
    (torch evaluates lerp(a, b, w) as a + w*(b-a) for w < 0.5 and
    b - (b-a)*(1-w) otherwise).
    """
    u = np.ascontiguousarray(u_flat, dtype=np.float32)
    s = np.sort(u, axis=1)
    n = u.shape[1]
    rank = np.float32(p) * np.float32(n - 1)
    lo = np.floor(rank)
    hi = np.ceil(rank)
    w = np.float32(rank - lo)
    a, b = s[:, int(lo)], s[:, int(hi)]
    d = (b - a).astype(np.float32)
    if w < np.float32(0.5):
        return (a + w * d).astype(np.float32)
    return (b - d * (np.float32(1) - w)).astype(np.float32)


def quantile_mask(u, p):
    """mask = u > quantile(u.flatten(1), p) per sample, strict (:112). Returns (mask uint8, q)."""
    B = u.shape[0]
    q = quantile_linear(u.reshape(B, -1), p)
    mask = u > q.reshape((B,) + (1,) * (u.ndim - 1))
    return mask.astype(np.uint8), q


def posterior_blend(eps, u, mask, M, alpha_hat_t):
    """:115-120 with B == 1 semantics kept as written: eps.sum(dim=0) sums the batch axis."""
    eps = eps.astype(np.float32)
    inv = (np.float32(1) / u).astype(np.float32)
    tr = (np.float32(M) * inv + np.float32(1) / np.float32(alpha_hat_t)).astype(np.float32)
    prec = (np.float32(1) / tr).astype(np.float32)
    post = prec * (inv * eps.sum(axis=0))
    m = mask.astype(np.float32)
    return (eps * (np.float32(1) - m) + m * post).astype(np.float32)


def guided_score_percentile(eps, x, denoise_cfg, alpha_hat_t, percentile, M, draw, use_posterior=True):
    """:83-120, model_type='stable-diffusion', posterior mode.

    ``denoise_cfg(x_hat) -> eps_hat`` wraps the CFG UNet call + combine
    (``predict_model_stable_diffusion`` :135-147).  ``x`` may be the CFG-doubled
    latent [2B,...] with eps [B,...] (B must be 1, broadcasting as written :86).
    """
    a = np.float32(alpha_hat_t)
    x0 = (x - np.sqrt(np.float32(1) - a) * eps) / np.sqrt(a)
    samples = []
    for _ in range(M):
        x_hat = (np.sqrt(a) * x0 + np.sqrt(np.float32(1) - a) * draw(eps)).astype(np.float32)
        samples.append(denoise_cfg(x_hat))
    if use_posterior:
        samples.append(eps)
    st = np.stack(samples, 0).astype(np.float32)
    u = st.var(axis=0, ddof=1, dtype=np.float32)
    mask, q = quantile_mask(u, percentile)
    return posterior_blend(eps, u, mask, M, alpha_hat_t), u, mask, q
